"""GPU: the Benettin estimator holds a bounded part of its records on the device (qgs_amd/toolbox/lyapunov.py, round 4).

The reference keeps one trajectory's records at a time in HOST memory (qgs/toolbox/lyapunov.py:232-358, 555-632): what bounds
a run is the host.  Here the base trajectory and the records live in windows within QGS_HIP_RECORD_WINDOW_MB of device memory;
a run cut into windows is bitwise the run that fits in one."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, load_golden

pytestmark = pytest.mark.gpu


def _estimator(device=None):
    from qgs_amd.functions.tendencies import tendencies_from_tensor
    from qgs_amd.toolbox.lyapunov import LyapunovsEstimator
    g = load_golden('m36')
    f, Df = tendencies_from_tensor(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    est = LyapunovsEstimator(num_threads=1, device=device)
    est.set_func(f, Df)
    return est, f, g.ndim


def _run(est, ic, forward, ws, n_vec, seed=11, t=8.0):
    np.random.seed(seed)
    est.compute_lyapunovs(0., 2.0, t, 0.1, 0.05, ic=ic, write_steps=ws, n_vec=n_vec, forward=forward)
    tt, traj, exps, vecs = est.get_lyapunovs()
    return np.array(tt), np.array(traj), np.array(exps), np.array(vecs), list(est.last_windows)


@pytest.mark.parametrize('forward', [False, True])
@pytest.mark.parametrize('ws', [1, 3])
@pytest.mark.parametrize('n_vec', [5, 36])
def test_windowed_run_is_bitwise_the_run_in_one_piece(monkeypatch, forward, ws, n_vec):
    est, f, ndim = _estimator()
    ic = np.random.RandomState(3).rand(1024, ndim) * 0.01
    monkeypatch.delenv('QGS_HIP_RECORD_WINDOW_MB', raising=False)
    whole = _run(est, ic, forward, ws, n_vec)
    assert whole[4] == [(1, 1)]                                       # default budget: everything resident, one window of each
    monkeypatch.setenv('QGS_HIP_RECORD_WINDOW_MB', '64' if n_vec == 36 else '4')
    cut = _run(est, ic, forward, ws, n_vec)
    assert cut[4][0][0] >= 2 and cut[4][0][1] >= 3, cut[4]           # several base-trajectory windows, several record windows
    for a, b in zip(whole[:4], cut[:4]):
        assert a.shape == b.shape and np.array_equal(a, b)
    assert np.isfinite(whole[2]).all() and np.isfinite(whole[3]).all()
    est.terminate()
    f.operands.release()


def test_windowed_run_on_a_device_list(monkeypatch):
    """Two shards (both on the test box's one GPU), each with its own windows, filling their slices of the result blocks: bitwise
    the same shards with everything resident.  (Against ONE model holding all members the kernels differ -- they are chosen by
    ensemble size -- so that comparison is to rounding.)"""
    ic = np.random.RandomState(4).rand(700, 36) * 0.01
    est, f, ndim = _estimator(device=[0, 0])
    monkeypatch.delenv('QGS_HIP_RECORD_WINDOW_MB', raising=False)
    whole = _run(est, ic, False, 2, 6)
    assert whole[4] == [(1, 1), (1, 1)]
    monkeypatch.setenv('QGS_HIP_RECORD_WINDOW_MB', '2')
    cut = _run(est, ic, False, 2, 6)
    assert len(cut[4]) == 2 and all(w[0] >= 2 and w[1] >= 2 for w in cut[4]), cut[4]
    for a, b in zip(whole[:4], cut[:4]):
        assert a.shape == b.shape and np.array_equal(a, b)
    est.terminate()
    monkeypatch.delenv('QGS_HIP_RECORD_WINDOW_MB', raising=False)
    est1, f1, _ = _estimator()
    one = _run(est1, ic, False, 2, 6)
    assert np.abs(one[1] - whole[1]).max() < 1e-12 and np.abs(one[3] - whole[3]).max() < 1e-9 and np.abs(one[2] - whole[2]).max() < 1e-8
    est1.terminate()
    f.operands.release()
    f1.operands.release()


@pytest.mark.parametrize('forward', [False, True])
def test_member_groups_fill_their_slices(monkeypatch, forward):
    """Records too large for one window leave in member groups when the ensemble is large (round 5: a group's record is one
    contiguous piece of each result block).  Forced here at a small size: 5 groups of 192 members + one of 64; every group's
    slice is bitwise what a run of just those members gives, and the whole agrees with one pass to rounding."""
    est, f, ndim = _estimator()
    ic = np.random.RandomState(5).rand(1024, ndim) * 0.01
    monkeypatch.delenv('QGS_HIP_RECORD_WINDOW_MB', raising=False)
    monkeypatch.delenv('QGS_HIP_RECORD_GROUP_MEMBERS', raising=False)
    whole = _run(est, ic, forward, 1, 12)
    monkeypatch.setenv('QGS_HIP_RECORD_GROUP_MEMBERS', '192')
    grouped = _run(est, ic, forward, 1, 12)
    assert grouped[4] == [(1, 1)]
    for a, b, tol in zip(whole[1:4], grouped[1:4], (1e-12, 1e-8, 1e-9)):
        assert a.shape == b.shape and np.abs(a - b).max() <= tol * max(1.0, np.abs(a).max())
    # the third group alone (the same random draws: the start matrices are drawn for the whole ensemble, in member order)
    monkeypatch.delenv('QGS_HIP_RECORD_GROUP_MEMBERS', raising=False)
    np.random.seed(11)
    a0 = np.random.random((1024, ndim, 12))
    est._run(est._pretime, est._time, 0.05, ic[384:576], 1, 12, forward, False, False, a0=a0[384:576])
    for got, want in ((est._recorded_traj, grouped[1][384:576]), (est._recorded_exp, grouped[2][384:576]), (est._recorded_vec, grouped[3][384:576])):
        assert np.array_equal(np.squeeze(got), want)
    est.terminate()
    f.operands.release()


def test_record_larger_than_the_default_budget_reaches_the_host(monkeypatch):
    """Config-4 size (16 384 members x 36 vectors), every interval recorded: 179 MB per record, 3.8 GB of records against the
    default 8 GB budget (half of it for the two record windows) -- the record crosses in several windows.  The first members
    agree with a small run of the same members (other kernels at that size: tolerance, not bitwise)."""
    est, f, ndim = _estimator()
    monkeypatch.setenv('QGS_HIP_RECORD_WINDOW_MB', '8192')          # (set by hand: the estimator then keeps to it, see _compute_shard)
    ic = np.random.RandomState(5).rand(16384, ndim) * 0.01
    tt, traj, exps, vecs, windows = _run(est, ic, False, 1, 36, seed=21, t=4.0)
    assert vecs.shape == (16384, ndim, 36, 21) and windows[0][1] >= 2, windows
    small = _run(est, ic[:64], False, 1, 36, seed=21, t=4.0)
    assert np.abs(traj[:64] - small[1]).max() < 1e-12
    assert np.abs(vecs[:64] - small[3]).max() < 1e-8
    assert np.abs(exps[:64] - small[2]).max() < 1e-7
    for i in (0, 8191, 16383):
        q = vecs[i, :, :, -1]
        assert np.abs(q.T @ q - np.eye(36)).max() < 1e-12
    est.terminate()
    f.operands.release()


def test_a_record_beyond_host_memory_says_so(monkeypatch):
    """16 384 members x 36 vectors x 5 000 recorded intervals: 0.9 TB of records.  On a host that has that much the run is
    possible (the GPU boxes of this pool have 3 TB; 72 GB of records: `profiles/r04_lyap_big.json`); on one that has not, the
    estimator refuses up front with a message about HOST memory -- never a failed device allocation.  The host's available
    memory is what /proc/meminfo says; here it is made to say 128 GB."""
    from qgs_amd.toolbox import lyapunov
    monkeypatch.setattr(lyapunov, '_host_memory_available', lambda: 128 << 30)
    est, f, ndim = _estimator()
    ic = np.random.RandomState(6).rand(16384, ndim) * 0.01
    with pytest.raises(MemoryError, match='host memory'):
        est.compute_lyapunovs(0., 1.0, 501.0, 0.1, 0.01, ic=ic, write_steps=1, n_vec=36)
    # the same request with every 500th interval recorded fits (11 records) and runs
    monkeypatch.setattr(lyapunov, '_host_memory_available', lambda: 8 << 30)
    np.random.seed(2)
    est.compute_lyapunovs(0., 0.2, 1.2, 0.1, 0.05, ic=ic[:4096], write_steps=5, n_vec=4)
    tt, traj, exps, vecs = est.get_lyapunovs()
    assert vecs.shape == (4096, ndim, 4, 3) and np.isfinite(vecs).all()
    est.terminate()
    f.operands.release()
