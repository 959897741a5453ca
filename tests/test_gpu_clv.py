"""GPU: covariant Lyapunov vectors (qgs_amd/toolbox/lyapunov.py CovariantLyapunovsEstimator) against goldens captured from the
reference's jitted loops (qgs/toolbox/lyapunov.py:1174-1330, tests/golden/make_golden.py gen_clv) with the same np.random
seed: method 0 (Ginelli et al.) and method 1 (intersection of the backward and forward subspaces)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, rel_err

pytestmark = pytest.mark.gpu


def _columns_up_to_sign(a, b):
    """max difference of unit columns that may differ by their sign (a singular vector's sign is LAPACK's choice and flips
    with the last bit of its input)"""
    s = np.sign(np.sum(a * b, axis=1, keepdims=True))
    return np.abs(a - s * b).max()


@pytest.mark.parametrize('name,device', [('rp20', None), ('rp20', [0, 0]), ('m36', None)])
def test_clv_estimator_vs_reference(name, device):
    from model_configs import MAKERS
    from qgs_amd.functions.tendencies import create_tendencies
    from qgs_amd.toolbox.lyapunov import CovariantLyapunovsEstimator
    g = np.load(os.path.join(GOLDEN_DIR, 'clv_%s.npz' % name))
    meta = json.loads(bytes(g['meta_json']).decode())
    f, Df = create_tendencies(MAKERS[name]())
    est = CovariantLyapunovsEstimator(num_threads=1, device=device)
    est.set_func(f, Df)
    runs = [(cs, resident) for cs in meta['cases'] for resident in ((True, False) if cs['method'] == 0 else (None,))]
    for cs, resident in runs:
        tag = cs['tag']
        np.random.seed(cs['seed'])
        est.set_noise_pert(cs['noise_pert'])
        # method 0 both ways: the Q / R record resident on the GPU (R, the backward recursion and the vectors in HIP kernels),
        # and the records delivered to the host with the recursion in NumPy (what a record too large for the GPU falls back to)
        est.device_resident = resident
        est.compute_clvs(meta['t0'], meta['ta'], meta['tb'], meta['tc'], meta['dt'], meta['mdt'], ic=g['ic'], write_steps=cs['ws'],
                         method=cs['method'], backward_vectors=True, forward_vectors=True)
        if cs['method'] == 0:
            assert est.last_path == ('device' if resident else 'host')
        tt, traj, exps, vecs = est.get_clvs()
        assert traj.shape == np.squeeze(g[tag + '_traj']).shape and vecs.shape == np.squeeze(g[tag + '_vec']).shape, tag
        assert rel_err(traj, np.squeeze(g[tag + '_traj'])) < 1e-12, tag
        want_vec, want_exp = g[tag + '_vec'], g[tag + '_exp']
        got_vec = est._recorded_vec
        if cs['method'] == 0:
            assert rel_err(got_vec, want_vec) < 1e-8, tag
            assert est.get_blvs() is None and est.get_flvs() is None
        else:
            for i in range(got_vec.shape[0]):
                for r in range(got_vec.shape[3]):
                    assert _columns_up_to_sign(got_vec[i, :, :, r], want_vec[i, :, :, r]) < 1e-7, (tag, i, r)
            assert rel_err(est.get_blvs()[3], np.squeeze(g[tag + '_bvec'])) < 1e-9, tag
            assert rel_err(est.get_flvs()[3], np.squeeze(g[tag + '_fvec'])) < 1e-9, tag
        assert np.abs(est._recorded_exp - want_exp).max() < 1e-7 * max(1.0, np.abs(want_exp).max()), tag
        if cs['ws'] > 0:
            assert np.shape(tt)[0] == traj.shape[-1]
        else:
            assert tt == g['time'][-1]
    est.terminate()
    f.operands.release()


@pytest.mark.parametrize('name,n_vec,sub', [('m36', 6, 1), ('t228', 5, 2)])
def test_clvs_are_covariant(name, n_vec, sub):
    """What makes the vectors covariant, checked without the reference (method 0; MAOOAM-36, and MAOOAM 6x6 whose tangent model
    runs in the LDS-resident kernels): the tangent model carries the CLVs at one record onto the CLVs at the next, column by
    column (up to the growth factor) -- exactly so by construction (v = Q a with a(t_n) ~ R_n^-1 a(t_n+1)), whatever the
    convergence of the windows."""
    from conftest import load_golden
    from qgs_amd.functions.tendencies import tendencies_from_tensor
    from qgs_amd.integrators import integrate as fn
    from qgs_amd.toolbox.lyapunov import CovariantLyapunovsEstimator
    g = load_golden(name)
    f, Df = tendencies_from_tensor(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    ic = np.random.RandomState(3).rand(3, g.ndim) * 0.01
    est = CovariantLyapunovsEstimator(num_threads=1)
    est.set_func(f, Df)
    np.random.seed(11)
    dt = 0.1 if sub == 1 else 0.125              # (0.125: grid values exact in binary, no sliver interval from np.arange)
    mdt = dt / sub
    n_rec = 11 if sub == 1 else 4
    est.compute_clvs(0., 0.5 if sub == 1 else 0.125, 1.5 if sub == 1 else 0.5, 2.0 if sub == 1 else 0.75, dt, mdt, ic=ic, write_steps=1,
                     n_vec=n_vec, method=0)
    tt, traj, exps, vecs = est.get_clvs()
    assert est.last_path == 'device'
    assert vecs.shape == (3, g.ndim, n_vec, n_rec) and np.isfinite(exps).all()
    b, c, a = fn.resolve_tableau(None, None, None)
    for r in range(n_rec - 1):
        d = tt[r + 1] - tt[r]
        grid = np.concatenate((np.arange(tt[r], tt[r] + d, mdt), np.full((1,), tt[r] + d)))
        _, sol = fn.run_rk_tgls(f, Df, grid, np.ascontiguousarray(traj[:, :, r]), np.ascontiguousarray(vecs[:, :, :, r]), 1, 0,
                                b, c, a, False, 1., None)
        nxt = sol[..., 0] / np.sqrt(np.sum(sol[..., 0] ** 2, axis=1, keepdims=True))
        for i in range(3):
            assert _columns_up_to_sign(nxt[i], vecs[i, :, :, r + 1]) < 1e-9, (r, i)
        # the local exponent recorded at r is the growth over the interval that starts there
        growth = np.log(np.sqrt(np.sum(sol[..., 0] ** 2, axis=1))) / d
        assert np.abs(growth - exps[:, :, r]).max() < 1e-8 * max(1.0, np.abs(exps).max()), r
    est.terminate()
    f.operands.release()


@pytest.mark.parametrize('method', [0, 1])
def test_clvs_with_the_benettin_runs_in_member_groups(monkeypatch, method):
    """The Benettin runs under both methods (method 0 with its records on the host: vectors, the matrices before the QR steps and
    the junction states; method 1: backward and forward vectors) with the ensemble forced into member groups of 64 (200 members:
    three groups and a ragged one) give the vectors of the runs in one pass -- to rounding, the kernels being chosen by ensemble size."""
    from conftest import load_golden
    from qgs_amd.functions.tendencies import tendencies_from_tensor
    from qgs_amd.toolbox.lyapunov import CovariantLyapunovsEstimator
    g = load_golden('m36')
    f, Df = tendencies_from_tensor(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    ic = np.random.RandomState(8).rand(200, g.ndim) * 0.01
    res = {}
    for groups in (None, '64'):
        if groups is None:
            monkeypatch.delenv('QGS_HIP_RECORD_GROUP_MEMBERS', raising=False)
        else:
            monkeypatch.setenv('QGS_HIP_RECORD_GROUP_MEMBERS', groups)
        est = CovariantLyapunovsEstimator(num_threads=1)
        est.set_func(f, Df)
        est.device_resident = False
        np.random.seed(21)
        est.compute_clvs(0., 0.5, 1.5, 2.0, 0.1, 0.05, ic=ic, write_steps=1, n_vec=5, method=method, backward_vectors=True,
                         forward_vectors=True)
        tt, traj, exps, vecs = est.get_clvs()
        if method == 1:
            # (the intersection of the two subspaces is a singular-vector problem: inside a nearly degenerate pair its columns turn
            # with the last bits of the input, so what is compared are the two Benettin runs themselves)
            vecs = np.concatenate((est.get_blvs()[3], est.get_flvs()[3]), axis=2)
        res[groups] = (np.array(traj), np.array(exps), np.array(vecs))
        est.terminate()
    (t0, e0, v0), (t1, e1, v1) = res[None], res['64']
    assert t0.shape == t1.shape and v0.shape == v1.shape and np.isfinite(v1).all()
    assert rel_err(t1, t0) < 1e-12
    for i in range(0, 200, 7):
        for r in range(v0.shape[3]):
            assert _columns_up_to_sign(v1[i, :, :, r], v0[i, :, :, r]) < 1e-7, (method, i, r)
    if method == 0:
        assert np.abs(e1 - e0).max() < 1e-7 * max(1.0, np.abs(e0).max())
    f.operands.release()


@pytest.mark.parametrize('resident', [True, False])
def test_clv_base_trajectory_windows_shorter_than_an_interval(monkeypatch, resident):
    """Method 0 reads the fine base trajectory once per interval, dt / mdt = 10 grid steps apart; with a window budget that
    holds 3 states the reader has to cross several windows between two reads (`_BaseTrajectory.state`, which used to move one
    window per call and then refuse).  Bitwise the run with everything resident."""
    from conftest import load_golden
    from qgs_amd.functions.tendencies import tendencies_from_tensor
    from qgs_amd.toolbox.lyapunov import CovariantLyapunovsEstimator
    g = load_golden('m36')
    f, Df = tendencies_from_tensor(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    ic = np.random.RandomState(5).rand(64, g.ndim) * 0.01
    est = CovariantLyapunovsEstimator(num_threads=1)
    est.set_func(f, Df)
    est.device_resident = resident

    def run():
        np.random.seed(4)
        est.compute_clvs(0., 0.5, 1.5, 2.0, 0.1, 0.01, ic=ic, write_steps=1, n_vec=4, method=0)
        tt, traj, exps, vecs = est.get_clvs()
        return np.array(traj), np.array(exps), np.array(vecs)
    monkeypatch.delenv('QGS_HIP_RECORD_WINDOW_MB', raising=False)
    whole = run()
    # base trajectory: budget / 4 over states of 36 x 64 x 8 bytes -> windows of 3 steps
    monkeypatch.setenv('QGS_HIP_RECORD_WINDOW_MB', '%.6f' % (4 * 4 * 36 * 64 * 8 / 1048576.))
    cut = run()
    for a, b in zip(whole, cut):
        assert a.shape == b.shape and np.isfinite(a).all() and np.array_equal(a, b)
    est.terminate()
    f.operands.release()


def test_batched_matmul_and_backstep_kernels_vs_numpy():
    """The two kernels of the covariant estimator against NumPy, member by member: C = A B, A^T B, the upper triangle of a square
    product, a product with an upper-triangular right factor; one backward step a <- normalise(R^-1 a + noise) with
    solve_triangular_matrix / normalize_matrix_columns of qgs_amd/functions/util.py (the reference's helpers, util.py:56-98)."""
    import torch
    from qgs_amd.functions.tendencies import tendencies_from_tensor
    from qgs_amd.functions.util import normalize_matrix_columns, solve_triangular_matrix
    f, _ = tendencies_from_tensor(2, np.array([[1, 0, 1]], dtype=np.int32), np.array([1.0]))
    m = f.hip_model()
    rng = np.random.RandomState(8)

    def to_dev(x, ld):                                   # (n, r, c) -> [r][c][member]
        d = torch.zeros(x.shape[1:] + (ld,), dtype=torch.float64, device='cuda')
        d[..., :x.shape[0]] = torch.from_numpy(np.ascontiguousarray(np.moveaxis(x, 0, -1))).cuda()
        return d

    def to_host(d, n):
        return np.moveaxis(d[..., :n].cpu().numpy(), -1, 0)
    for n, (r, k, c) in ((70, (5, 7, 6)), (3, (36, 36, 36)), (130, (1, 3, 9)), (2, (228, 40, 5))):
        ld = (n + 63) // 64 * 64
        a, b = rng.randn(n, r, k), rng.randn(n, k, c)
        d_a, d_at, d_b = to_dev(a, ld), to_dev(np.swapaxes(a, 1, 2), ld), to_dev(b, ld)
        out = torch.full((r, c, ld), 7.0, dtype=torch.float64, device='cuda')
        m.batched_matmul_device(n, ld, r, k, c, d_a.data_ptr(), d_b.data_ptr(), out.data_ptr())
        assert np.abs(to_host(out, n) - a @ b).max() < 1e-12 * k
        assert m.last_kernel_info()['name'] == 'batched_matmul_kernel'
        out.fill_(7.0)
        m.batched_matmul_device(n, ld, r, k, c, d_at.data_ptr(), d_b.data_ptr(), out.data_ptr(), trans_a=True)
        assert np.abs(to_host(out, n) - a @ b).max() < 1e-12 * k
        assert float(out[..., n:].min()) == 7.0 == float(out[..., n:].max())          # padding lanes are not touched
    for n, nv, nd in ((70, 6, 9), (3, 36, 36), (65, 1, 4)):
        ld = (n + 63) // 64 * 64
        q, a = rng.randn(n, nd, nv), rng.randn(n, nd, nv)
        am = np.triu(rng.randn(n, nv, nv))
        rm = np.triu(rng.randn(n, nv, nv)) + 4.0 * np.eye(nv)
        noise = rng.randn(n, nv)
        d_q, d_a, d_am, d_rm = to_dev(q, ld), to_dev(a, ld), to_dev(am, ld), to_dev(rm, ld)
        d_noise = to_dev(noise[:, :, None], ld)                     # [vector][1][member]
        out = torch.full((nv, nv, ld), 7.0, dtype=torch.float64, device='cuda')
        m.batched_matmul_device(n, ld, nv, nd, nv, d_q.data_ptr(), d_a.data_ptr(), out.data_ptr(), trans_a=True, triangular=1)
        assert np.abs(to_host(out, n) - np.triu(np.swapaxes(q, 1, 2) @ a)).max() < 1e-12 * nd
        vec = torch.full((nd, nv, ld), 7.0, dtype=torch.float64, device='cuda')
        m.batched_matmul_device(n, ld, nd, nv, nv, d_q.data_ptr(), d_am.data_ptr(), vec.data_ptr(), triangular=2)
        assert np.abs(to_host(vec, n) - q @ am).max() < 1e-12 * nv
        # one backward step
        for pert, with_noise in ((0.0, False), (1e-2, True)):
            want = solve_triangular_matrix(rm, am)
            want[:, np.arange(nv), np.arange(nv)] += noise * pert
            want, want_norm = normalize_matrix_columns(want)
            a_out = torch.full((nv, nv, ld), 7.0, dtype=torch.float64, device='cuda')
            nrm = torch.zeros((nv, ld), dtype=torch.float64, device='cuda')
            m.clv_backstep_device(n, ld, nv, d_rm.data_ptr(), d_am.data_ptr(), a_out.data_ptr(), nrm.data_ptr(),
                                  d_noise.data_ptr() if with_noise else None, pert)
            assert m.last_kernel_info()['name'] == 'clv_backstep_kernel'
            assert np.abs(to_host(a_out, n) - want).max() < 1e-12
            assert np.abs(nrm[:, :n].cpu().numpy().T - want_norm).max() < 1e-12 * max(1.0, np.abs(want_norm).max())
    f.operands.release()
