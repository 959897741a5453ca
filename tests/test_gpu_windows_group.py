"""GPU: the record-window pipeline of the host-layout entry points, the multi-device group (tested on ONE GPU by listing
device 0 more than once: two models, two pipelines, two shards), BASELINE config 5 at its stated size through the shard
bookkeeping, the single-state f / Df path, and the C-ABI's argument checks.

Every comparison of two HIP runs here is BITWISE: cutting a run into record windows or an ensemble into shards must not change
one bit (members are independent; a window boundary hands the state over through memory).  Comparisons with the oracle use the
tolerances of test_gpu_parity.py.
"""
import ctypes
import os

import numpy as np
from kernel_names import LDS_STEPPER
import pytest

from conftest import GOLDEN_DIR, REPO, RK4, load_golden, rel_err

pytestmark = pytest.mark.gpu

B, C, A = RK4['b'], RK4['c'], RK4['a']


def _grid(steps, dt=0.1):
    return np.concatenate((np.arange(0., steps * dt, dt), [steps * dt]))[:steps + 1]


def _model(name, **kw):
    from qgs_amd import _lib
    g = load_golden(name)
    return g, _lib.HipModel(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'], **kw)


class _Knobs(object):
    """Set QGS_HIP_* knobs and make the model re-read them (they are read at model creation and at set_kernel)."""

    def __init__(self, monkeypatch, model, kind=0):
        self.mp, self.model, self.kind = monkeypatch, model, kind

    def set(self, **env):
        for k, v in env.items():
            if v is None:
                self.mp.delenv(k, raising=False)
            else:
                self.mp.setenv(k, str(v))
        self.model.set_kernel(self.kind)


# ---- record windows ------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize('d2h', ['kernel', 'copy'])
@pytest.mark.parametrize('kind', [0, 1])
def test_windowed_record_equals_unwindowed(monkeypatch, d2h, kind):
    """A forced 64 MB window budget cuts a 4 113-member, 101-record run (1.2 MB per record) into windows with a ragged last
    one; forward / backward, write_steps 1 and 3; page-locked destination written by the unpack kernel ('kernel') and the
    staged strided copy ('copy')."""
    g, m = _model('m36')
    knobs = _Knobs(monkeypatch, m, kind)
    n, steps = (4113, 100) if kind == 0 else (700, 40)
    ic = np.random.RandomState(11).rand(n, g.ndim) * 0.01
    t = _grid(steps)
    for direction in (1, -1):
        for ws in (1, 3):
            knobs.set(QGS_HIP_RECORD_WINDOW_MB=None, QGS_HIP_D2H=None)
            whole = np.array(m.rk_integrate(t, ic, direction, ws, B, C, A))
            assert m.last_windows == 1
            knobs.set(QGS_HIP_RECORD_WINDOW_MB=64 if kind == 0 else 2, QGS_HIP_D2H=d2h)
            cut = m.rk_integrate(t, ic, direction, ws, B, C, A)
            assert m.last_windows >= 2, m.last_windows
            assert np.array_equal(whole, cut), (direction, ws, m.last_windows)
    knobs.set(QGS_HIP_RECORD_WINDOW_MB=None, QGS_HIP_D2H=None)
    m.close()


def test_member_groups_into_pageable_memory(monkeypatch):
    """Records of a large ensemble into pageable memory leave in member groups (each a contiguous piece of the caller's array,
    qgs_hip_api.hip rk_member_groups).  Forced at a small size: 1 500 members in groups of 448 (three + a ragged one of 156); every
    group's slice is bitwise a run of just those members, the whole agrees with one pass to rounding (kernels are chosen by
    ensemble size), forward and backward, write_steps 1 and 3; a page-locked destination keeps the windows of records."""
    g, m = _model('m36')
    knobs = _Knobs(monkeypatch, m)
    n, steps = 1500, 60
    ic = np.random.RandomState(17).rand(n, g.ndim) * 0.01
    t = _grid(steps)
    for direction, ws in ((1, 1), (-1, 3)):
        knobs.set(QGS_HIP_RECORD_GROUP_MEMBERS=None, QGS_HIP_RECORD_WINDOW_MB=None)
        whole = np.array(m.rk_integrate(t, ic, direction, ws, B, C, A))
        assert m.last_groups == 1
        knobs.set(QGS_HIP_RECORD_GROUP_MEMBERS=448)
        out = np.full(whole.shape, np.nan)                                   # ordinary (pageable) memory
        m.rk_integrate(t, ic, direction, ws, B, C, A, out=out)
        assert m.last_groups == 4 and m.last_windows == 1
        assert np.abs(out - whole).max() <= 1e-13 * np.abs(whole).max()
        pinned = m.rk_integrate(t, ic, direction, ws, B, C, A)              # a block of the page-locked pool: no groups
        assert m.last_groups == 1 and np.array_equal(pinned, whole)
        knobs.set(QGS_HIP_RECORD_GROUP_MEMBERS=None)
        for lo in (448, 1344):
            part = np.array(m.rk_integrate(t, ic[lo:lo + 448], direction, ws, B, C, A))
            assert np.array_equal(part, out[lo:lo + 448]), (direction, ws, lo)
    # a window budget set by hand keeps the windows of records (no groups by the rule)
    knobs.set(QGS_HIP_RECORD_GROUP_MEMBERS=None, QGS_HIP_RECORD_WINDOW_MB=1)
    out = np.full(whole.shape, np.nan)
    m.rk_integrate(t, ic, -1, 3, B, C, A, out=out)
    assert m.last_groups == 1 and m.last_windows >= 3 and np.array_equal(out, whole)
    knobs.set(QGS_HIP_RECORD_GROUP_MEMBERS=None, QGS_HIP_RECORD_WINDOW_MB=None)
    m.close()


@pytest.mark.parametrize('case', ['rksplit4', 'dense_tableau', 'rank5', 'heun2'])
def test_windowed_record_in_the_other_stepper_families(monkeypatch, case):
    """The remaining steppers behind `rk_launch` carry their state across a window boundary as well: the 4-way row-split kernel
    (mid-size ensembles; also with two stages), the general-tableau kernel (3/8 rule: partial stage sums in LDS), a rank-5 model
    (derived monomials)."""
    name = 'd38' if case == 'rank5' else 'm36'
    g, m = _model(name)
    knobs = _Knobs(monkeypatch, m)
    n, steps = (6000, 30) if case == 'rksplit4' else (700, 30)
    ic = np.random.RandomState(31).rand(n, g.ndim) * 0.01
    t = _grid(steps)
    b, c, a = B, C, A
    if case == 'dense_tableau':
        b = np.array([1. / 8, 3. / 8, 3. / 8, 1. / 8])
        c = np.array([0., 1. / 3, 2. / 3, 1.])
        a = np.array([[0., 0, 0, 0], [1. / 3, 0, 0, 0], [-1. / 3, 1., 0, 0], [1., -1., 1., 0]])
    elif case == 'heun2':
        b, c, a = np.array([0.5, 0.5]), np.array([0., 1.]), np.array([[0., 0.], [1., 0.]])
    expect = {'rksplit4': 'qgs_spec_rksplit4_s4', 'dense_tableau': 'qgs_spec_rkd_s4', 'rank5': 'qgs_spec_rk_s4', 'heun2': 'qgs_spec_rksplit4_s2'}[case]
    m.set_kernel(2 if case in ('rank5', 'heun2', 'dense_tableau') else 0)
    knobs.kind = 2 if case in ('rank5', 'heun2', 'dense_tableau') else 0
    for direction, ws in ((1, 2), (-1, 1)):
        knobs.set(QGS_HIP_RECORD_WINDOW_MB=None)
        whole = np.array(m.rk_integrate(t, ic, direction, ws, b, c, a))
        assert m.last_kernel_info()['name'] in (expect, expect.replace('_rk_s', '_rkr_s')), m.last_kernel_info()['name']
        knobs.set(QGS_HIP_RECORD_WINDOW_MB=8 if case == 'rksplit4' else 1)
        cut = m.rk_integrate(t, ic, direction, ws, b, c, a)
        assert m.last_windows >= 3, m.last_windows
        assert np.array_equal(whole, cut), (case, direction, ws)
    knobs.set(QGS_HIP_RECORD_WINDOW_MB=None)
    m.close()


def test_windowed_ensemble_moments(monkeypatch):
    """`rk_integrate_moments` never holds the record: window after window is integrated and reduced.  Mean / variance against
    NumPy on the record of `rk_integrate`, final states bitwise, forward and backward, for one window and for many."""
    g, m = _model('m36')
    knobs = _Knobs(monkeypatch, m)
    n, steps = 3000, 40
    ic = np.random.RandomState(41).rand(n, g.ndim) * 0.01
    t = _grid(steps)
    for direction, ws in ((1, 1), (-1, 3), (1, 0)):
        knobs.set(QGS_HIP_RECORD_WINDOW_MB=None)
        rec = np.array(m.rk_integrate(t, ic, direction, ws, B, C, A))
        last = rec[:, :, 0 if direction == -1 else -1]
        for mb in (None, 4):
            knobs.set(QGS_HIP_RECORD_WINDOW_MB=mb)
            mean, var, fin = m.rk_integrate_moments(t, ic, direction, ws, B, C, A, variance=True, final_states=True)
            assert (m.last_windows == 1) == (mb is None or ws == 0), (mb, m.last_windows)
            assert rel_err(mean, rec.mean(axis=0)) < 1e-13
            assert np.abs(var - rec.var(axis=0)).max() < 1e-10 * rec.var(axis=0).max()
            assert np.array_equal(fin, last)
    knobs.set(QGS_HIP_RECORD_WINDOW_MB=None)
    m.close()


def test_windowed_record_into_pageable_memory(monkeypatch):
    """The result block of a plain C caller is pageable memory: staged copy per window, same bits."""
    from qgs_amd import _lib
    g, m = _model('m36')
    knobs = _Knobs(monkeypatch, m)
    n, steps, ws = 1500, 60, 1
    ic = np.random.RandomState(12).rand(n, g.ndim) * 0.01
    t = _grid(steps)
    nrec = _lib.n_records(t, ws)
    whole = np.array(m.rk_integrate(t, ic, 1, ws, B, C, A))
    knobs.set(QGS_HIP_RECORD_WINDOW_MB=16)
    out = np.full((n, g.ndim, nrec), np.nan)                     # a fresh NumPy array: not page-locked
    rc = _lib.lib().qgs_rk_integrate(m._h, n, ic, t, len(t), 1, ws, 4, B, C, A, out)
    assert rc == 0, _lib.last_error()
    assert m.last_windows >= 3 and np.array_equal(out, whole)
    knobs.set(QGS_HIP_RECORD_WINDOW_MB=None)
    m.close()


def test_windowed_small_ensemble_and_zero_step_tail(monkeypatch):
    """Wavefront-per-trajectory kernel (10 members, 18 KB per record) and a window that holds nothing but the final record: 9
    steps, write_steps 3 -> records at steps 0, 3, 6 and the final one; 0.16 MB pays for W = 3 on the staged route, which puts
    the final record alone into a zero-step window."""
    g, m = _model('m36')
    knobs = _Knobs(monkeypatch, m)
    ic = np.random.RandomState(13).rand(10, g.ndim) * 0.01
    for steps, ws in ((9, 3), (50, 1), (7, 0)):
        t = _grid(steps)
        for direction in (1, -1):
            knobs.set(QGS_HIP_RECORD_WINDOW_MB=None)
            whole = np.array(m.rk_integrate(t, ic, direction, ws, B, C, A))
            assert m.last_kernel_info()['name'] == 'gen_rk_wave_kernel'
            knobs.set(QGS_HIP_RECORD_WINDOW_MB=0.16, QGS_HIP_D2H='copy')
            cut = m.rk_integrate(t, ic, direction, ws, B, C, A)
            assert m.last_windows == (2 if (steps, ws) == (9, 3) else (17 if ws == 1 else 1)), m.last_windows
            assert np.array_equal(whole, cut), (steps, ws, direction)
    knobs.set(QGS_HIP_RECORD_WINDOW_MB=None, QGS_HIP_D2H=None)
    m.close()


def test_window_plan_with_final_record_alone(monkeypatch):
    """ndim 228 (LDS-resident stepper), 200 members: 0.45 MB per record, 2 MB budget -> windows of 1-2 records; 9 steps with
    write_steps 3 ends on a window that only holds the final record."""
    g, m = _model('t228')
    knobs = _Knobs(monkeypatch, m, kind=2)
    ic = np.random.RandomState(14).rand(200, g.ndim) * 0.01
    t = _grid(9)
    for direction in (1, -1):
        knobs.set(QGS_HIP_RECORD_WINDOW_MB=None)
        whole = np.array(m.rk_integrate(t, ic, direction, 3, B, C, A))
        assert m.last_kernel_info()['name'].startswith('qgs_spec_rklds')
        knobs.set(QGS_HIP_RECORD_WINDOW_MB=2)
        cut = m.rk_integrate(t, ic, direction, 3, B, C, A)
        assert m.last_windows >= 2 and np.array_equal(whole, cut), direction
    knobs.set(QGS_HIP_RECORD_WINDOW_MB=None)
    m.close()


@pytest.mark.parametrize('kind', [0, 1])
def test_windowed_tangent_model_equals_unwindowed(monkeypatch, kind):
    """Trajectory + propagator records through windows: tangent forward, adjoint backward with `inverse`."""
    g, m = _model('m36')
    knobs = _Knobs(monkeypatch, m, kind)
    n, n_tg, steps = 300, 5, 24
    rng = np.random.RandomState(15)
    ic, tg = rng.rand(n, g.ndim) * 0.01, rng.randn(n, g.ndim, n_tg)
    t = _grid(steps)
    for direction, ws, adj, inv in ((1, 1, False, 1.), (-1, 5, True, -1.), (1, 0, False, 1.)):
        knobs.set(QGS_HIP_RECORD_WINDOW_MB=None, QGS_HIP_D2H=None)
        tr0, fm0 = (np.array(q) for q in m.rk_tgls_integrate(t, ic, tg, direction, ws, B, C, A, adj, inv))
        for d2h in ('kernel', 'copy'):
            knobs.set(QGS_HIP_RECORD_WINDOW_MB=2, QGS_HIP_D2H=d2h)           # (1 + 5) * 320 * 36 * 8 B = 0.53 MB per record
            tr1, fm1 = m.rk_tgls_integrate(t, ic, tg, direction, ws, B, C, A, adj, inv)
            assert ws == 0 or m.last_windows >= 3
            assert np.array_equal(tr0, tr1) and np.array_equal(fm0, fm1), (direction, ws, d2h)
    knobs.set(QGS_HIP_RECORD_WINDOW_MB=None, QGS_HIP_D2H=None)
    m.close()


def test_tangent_member_groups_into_pageable_memory(monkeypatch):
    """The tangent model's two record blocks (trajectory and propagators, both member-major) in member groups: 700 members in groups
    of 256; every group's slices are bitwise a run of just those members; tangent forward, adjoint backward with `inverse`."""
    g, m = _model('m36')
    knobs = _Knobs(monkeypatch, m)
    n, n_tg, steps = 700, 5, 24
    rng = np.random.RandomState(19)
    ic, tg = rng.rand(n, g.ndim) * 0.01, rng.randn(n, g.ndim, n_tg)
    t = _grid(steps)
    for direction, ws, adj, inv in ((1, 1, False, 1.), (-1, 5, True, -1.)):
        knobs.set(QGS_HIP_RECORD_GROUP_MEMBERS=None, QGS_HIP_RECORD_WINDOW_MB=None)
        tr0, fm0 = (np.array(q) for q in m.rk_tgls_integrate(t, ic, tg, direction, ws, B, C, A, adj, inv))
        assert m.last_groups == 1
        knobs.set(QGS_HIP_RECORD_GROUP_MEMBERS=256)
        out = (np.full(tr0.shape, np.nan), np.full(fm0.shape, np.nan))      # ordinary (pageable) memory
        m.rk_tgls_integrate(t, ic, tg, direction, ws, B, C, A, adj, inv, out=out)
        assert m.last_groups == 3 and m.last_windows == 1
        assert np.abs(out[0] - tr0).max() <= 1e-13 * np.abs(tr0).max() and np.abs(out[1] - fm0).max() <= 1e-12 * np.abs(fm0).max()
        knobs.set(QGS_HIP_RECORD_GROUP_MEMBERS=None)
        for lo, cnt in ((256, 256), (512, 188)):
            tr, fm = m.rk_tgls_integrate(t, ic[lo:lo + cnt], tg[lo:lo + cnt], direction, ws, B, C, A, adj, inv)
            assert np.array_equal(tr, out[0][lo:lo + cnt]) and np.array_equal(fm, out[1][lo:lo + cnt]), (direction, ws, lo)
    m.close()


def test_record_larger_than_its_device_budget_matches_the_oracle(monkeypatch):
    """A 65 536-member run whose record (65 536 x 36 x 41 doubles = 774 MB) is forced through 128 MB of device windows:
    sample members against the oracle, every record."""
    from oracle.oracle import OracleModel
    g, m = _model('m36')
    knobs = _Knobs(monkeypatch, m)
    ora = OracleModel(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    n, steps = 65536, 40
    ic = np.random.RandomState(16).rand(n, g.ndim) * 0.01
    t = _grid(steps)
    knobs.set(QGS_HIP_RECORD_WINDOW_MB=128)
    rec = m.rk_integrate(t, ic, 1, 1, B, C, A)
    assert rec.shape == (n, g.ndim, steps + 1) and m.last_windows >= 6
    pick = np.array([0, 1, 63, 64, 30000, 65535])
    ref = ora.integrate_runge_kutta_jit(t, ic[pick], 1, 1, B, C, A)
    assert rel_err(rec[pick], ref) < 1e-12
    assert np.array_equal(rec[:, :, 0], ic)
    knobs.set(QGS_HIP_RECORD_WINDOW_MB=None)
    m.close()


def test_rows_device_entry_point_equals_host_call(monkeypatch):
    """qgs_rk_integrate_rows_device (both blocks in HBM in the reference's layouts, what a rank of parallel.py runs)."""
    import torch
    g, m = _model('m36')
    knobs = _Knobs(monkeypatch, m)
    n, steps, ws = 3000, 30, 2
    ic = np.random.RandomState(17).rand(n, g.ndim) * 0.01
    t = _grid(steps)
    whole = np.array(m.rk_integrate(t, ic, -1, ws, B, C, A))
    d_ic = torch.from_numpy(ic).cuda()
    d_out = torch.empty(whole.shape, dtype=torch.float64, device='cuda')
    torch.cuda.synchronize()
    for mb in (None, 8):
        knobs.set(QGS_HIP_RECORD_WINDOW_MB=mb)
        d_out.fill_(float('nan'))
        torch.cuda.synchronize()
        m.rk_integrate_rows_device(n, d_ic.data_ptr(), t, -1, ws, B, C, A, d_out.data_ptr())
        assert np.array_equal(d_out.cpu().numpy(), whole), mb
    knobs.set(QGS_HIP_RECORD_WINDOW_MB=None)
    m.close()


# ---- several GPUs behind one handle (here: device 0 listed more than once) ----------------------------------------------------------

def test_group_equals_its_shards_bitwise():
    """A group's result is, bit for bit, what a single model returns for each shard on its own (concatenated in member order);
    against ONE single-model call over the whole ensemble it agrees to rounding -- the library picks its kernel by the size of
    the ensemble it is handed (wavefront-per-trajectory, row-split, plain, single-state), and a shard is a smaller ensemble."""
    from qgs_amd import _lib
    g, m = _model('m36')
    rng = np.random.RandomState(21)
    for devices, n in (([0, 0], 1001), ([0, 0, 0], 64), ([0, 0, 0, 0, 0], 3), ([0, 0], 9001)):
        grp = _lib.HipModelGroup(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'], devices=devices)
        assert len(grp) == len(devices)
        shards = [grp.shard(n, i) for i in range(len(devices))]
        assert sum(c for _, c in shards) == n and shards[0][0] == 0
        assert all(shards[i][0] + shards[i][1] == shards[i + 1][0] for i in range(len(shards) - 1))
        assert max(c for _, c in shards) - min(c for _, c in shards) <= 1            # remainder to the first shards
        assert [int(_lib.lib().qgs_model_info(q._h, 3)) for q in grp.models] == devices

        def by_shard(call, *arrays):
            parts = [call(*(q[a0:a0 + c] for q in arrays)) for a0, c in shards if c > 0]
            if isinstance(parts[0], tuple):
                return tuple(np.concatenate([q[k] for q in parts], axis=0) for k in range(len(parts[0])))
            return np.concatenate(parts, axis=0)

        ic = rng.rand(n, g.ndim) * 0.01
        got = grp.tendencies(ic)
        assert np.array_equal(got, by_shard(m.tendencies, ic)) and rel_err(got, m.tendencies(ic)) < 1e-14
        got = grp.jacobian(ic)
        assert np.array_equal(got, by_shard(m.jacobian, ic)) and rel_err(got, m.jacobian(ic)) < 1e-14
        t = _grid(23)
        for direction, ws in ((1, 0), (-1, 1), (1, 4)):
            got = grp.rk_integrate(t, ic, direction, ws, B, C, A)
            assert np.array_equal(got, by_shard(lambda q: m.rk_integrate(t, q, direction, ws, B, C, A), ic))
            assert rel_err(got, m.rk_integrate(t, ic, direction, ws, B, C, A)) < 1e-12
        if n > 2000:
            grp.close()
            continue
        tg = rng.randn(n, g.ndim, 3)
        a_tr, a_fm = grp.rk_tgls_integrate(t[:8], ic, tg, 1, 2, B, C, A, False, 1.)
        b_tr, b_fm = by_shard(lambda q, w: m.rk_tgls_integrate(t[:8], q, w, 1, 2, B, C, A, False, 1.), ic, tg)
        assert np.array_equal(a_tr, b_tr) and np.array_equal(a_fm, b_fm)
        c_tr, c_fm = m.rk_tgls_integrate(t[:8], ic, tg, 1, 2, B, C, A, False, 1.)
        assert rel_err(a_tr, c_tr) < 1e-12 and rel_err(a_fm, c_fm) < 1e-11
        mean, var, fin = grp.rk_integrate_moments(t, ic, 1, 4, B, C, A, final_states=True)
        rec = grp.rk_integrate(t, ic, 1, 4, B, C, A)
        assert rel_err(mean, rec.mean(axis=0)) < 1e-13 and np.abs(var - rec.var(axis=0)).max() < 1e-12 * max(rec.var(axis=0).max(), 1e-300) + 1e-22
        assert np.array_equal(fin, rec[:, :, -1])
        grp.close()
    m.close()


def test_group_into_a_pageable_block_of_a_c_caller(monkeypatch):
    """A plain C caller hands `qgs_group_rk_integrate` an ordinary (pageable) result block: the library never page-locks it (round 5);
    every shard's windows reach its slice through the device's drain thread and the page-locked bounce ring (csrc/host_bridge.cpp),
    three shards on one device queueing there at once -- also window by window."""
    from qgs_amd import _lib
    g, m = _model('m36')
    grp = _lib.HipModelGroup(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'], devices=[0, 0, 0])
    n, steps, ws = 20001, 40, 1
    ic = np.random.RandomState(51).rand(n, g.ndim) * 0.01
    t = _grid(steps)
    nrec = _lib.n_records(t, ws)
    expect = np.concatenate([np.array(m.rk_integrate(t, ic[a0:a0 + c], -1, ws, B, C, A)) for a0, c in (grp.shard(n, i) for i in range(3))], axis=0)
    for mb in (None, 64):
        if mb is None:
            monkeypatch.delenv('QGS_HIP_RECORD_WINDOW_MB', raising=False)
        else:
            monkeypatch.setenv('QGS_HIP_RECORD_WINDOW_MB', str(mb))
        grp.set_kernel(0)
        out = np.full((n, g.ndim, nrec), np.nan)                 # 236 MB, not page-locked
        rc = _lib.lib().qgs_group_rk_integrate(grp._h, n, ic, t, len(t), -1, ws, 4, B, C, A, out)
        assert rc == 0, _lib.last_error()
        assert np.array_equal(out, expect), mb
        assert (grp.models[0].last_windows > 1) == (mb is not None)
    monkeypatch.delenv('QGS_HIP_RECORD_WINDOW_MB', raising=False)
    grp.close()
    m.close()


@pytest.mark.parametrize('name', ['rp20', 'm36', 'd38'])
def test_integrator_classes_on_a_device_list_vs_reference_classes(name):
    """The class-level goldens of test_gpu_api.py with `device=[0, 0]`: same outputs through the sharded engine."""
    from qgs_amd.functions.tendencies import tendencies_from_tensor
    from qgs_amd.integrators.integrator import RungeKuttaIntegrator, RungeKuttaTglsIntegrator
    from qgs_amd.integrators.integrate import integrate_runge_kutta
    from qgs_amd import _lib
    g = load_golden(name)
    f, Df = tendencies_from_tensor(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    ic = g['rk_ic']
    integ = RungeKuttaIntegrator(num_threads=2, device=[0, 0])
    integ.set_func(f)
    assert isinstance(integ._model, _lib.HipModelGroup)
    integ.integrate(0., 1., 0.1, ic=ic[:4], write_steps=5)
    tt, tr = integ.get_trajectories()
    assert np.array_equal(tt, g['cls_rk_w5_time']) and rel_err(tr, g['cls_rk_w5_traj']) < 1e-12
    integ.integrate(0., 1., 0.1, ic=ic[:4], write_steps=0, forward=False)
    tt, tr = integ.get_trajectories()
    assert np.ndim(tt) == 0 and tt == g['cls_rk_w0b_time'] and rel_err(tr, g['cls_rk_w0b_traj']) < 1e-12
    integ.terminate()
    tinteg = RungeKuttaTglsIntegrator(num_threads=2, device='all')
    tinteg.set_func(f, Df)
    tinteg.integrate(0., 0.3, 0.1, ic=ic[:2], write_steps=1)
    tt, tr, fm = tinteg.get_trajectories()
    assert np.array_equal(tt, g['cls_tgls_time'])
    assert rel_err(tr, g['cls_tgls_traj']) < 1e-12 and rel_err(fm, g['cls_tgls_fm']) < 1e-11
    tinteg.terminate()
    tt, tr = integrate_runge_kutta(f, t0=0., t=1., dt=0.1, ic=ic, forward=False, write_steps=3, device=[0, 0, 0])
    assert np.array_equal(np.asarray(tt), g['api_b_w3_time']) and rel_err(tr, g['api_b_w3_traj']) < 1e-12
    f.operands.release()


def test_config5_full_size_through_the_shard_bookkeeping():
    """BASELINE config 5 as stated: MAOOAM-36 (the qgs_maooam.py parameter set of bench.py), 1 048 576 members, 1000 RK4
    steps, write_steps 0, 8 shards of 131 072.  On the one GPU present: (a) one 1 048 576-member call, (b) a group of eight
    models on device 0 (the in-process route of the integrator classes), (c) parallel.py's per-rank route, shard by shard with
    `shard_bounds`, concatenated in rank order.  All three bitwise equal; a sample against the oracle."""
    import sys
    import torch
    sys.path.insert(0, REPO)
    from bench import load_model_tensors
    from qgs_amd import _lib
    from qgs_amd.functions.tendencies import tendencies_from_tensor
    from qgs_amd.parallel import shard_bounds, _integrate_shard_on_device
    from oracle.oracle import OracleModel
    ndim, coo, val, jcoo, jval, _ = load_model_tensors()
    f, _ = tendencies_from_tensor(ndim, coo, val, jcoo, jval)
    n, world, steps = 1048576, 8, 1000
    ic = np.random.RandomState(21217).rand(n, ndim) * 0.01
    t = _grid(steps)
    one = np.array(f.hip_model(0).rk_integrate(t, ic, 1, 0, B, C, A))
    assert f.hip_model(0).last_kernel_info()['name'] == 'qgs_spec_rk_s4'
    grp = f.hip_model([0] * world)
    assert [grp.shard(n, i) for i in range(world)] == [(r * 131072, 131072) for r in range(world)]
    assert np.array_equal(grp.rk_integrate(t, ic, 1, 0, B, C, A), one)
    bounds = shard_bounds(n, world)
    assert bounds == [(r * 131072, (r + 1) * 131072) for r in range(world)]
    dev = torch.device('cuda', 0)
    parts = [_integrate_shard_on_device(f, dev, np.ascontiguousarray(ic[a:b]), t, True, 0, B, C, A).cpu().numpy() for a, b in bounds]
    assert np.array_equal(np.concatenate(parts, axis=0), one)
    pick = np.array([0, 131071, 131072, 524288, 1048575])
    ref = OracleModel(ndim, coo, val, jcoo, jval).integrate_runge_kutta_jit(t, ic[pick], 1, 0, B, C, A)
    assert rel_err(one[pick], ref) < 1e-10                       # 1000 steps: tolerance of test_gpu_parity.py
    f.operands.release()


# ---- one state: the callables handed to ODE solvers -----------------------------------------------------------------------------

@pytest.mark.parametrize('name', ['rp20', 'a36', 'm36', 't228', 'g30', 'd38', 'q38'])
def test_single_state_f_and_Df(name):
    """f(t, x) / Df(t, x) for ONE (ndim,) state (user_guide.rst:502-517): one launch on page-locked staging, against the
    reference goldens, and equal to the same state inside a batch."""
    g, m = _model(name)
    xs = g['fx_x']
    for i in (0, 3):
        fx = m.tendencies(xs[i])
        assert fx.shape == (g.ndim,) and rel_err(fx, g['fx_f'][i]) < 1e-14
        assert rel_err(fx, m.tendencies(xs[:8])[i]) < 1e-14
        if i < g['fx_Df'].shape[0]:
            J = m.jacobian(xs[i])
            assert J.shape == (g.ndim, g.ndim) and rel_err(J, g['fx_Df'][i]) < 1e-14
            assert m.last_kernel_info()['name'] in ('qgs_spec_jac', 'gen_jac_one_kernel')
    # a solver calls f thousands of times with changing states: results must not depend on what the staging block held
    rng = np.random.RandomState(5)
    batch = rng.rand(20, g.ndim) * 0.01
    ref = m.tendencies(batch)
    for i in range(20):
        assert rel_err(m.tendencies(batch[i]), ref[i]) < 1e-14
    m.close()


# ---- hostile arguments ------------------------------------------------------------------------------------------------------------

def test_cabi_rejects_bad_arguments_and_stays_usable():
    """Every bad argument returns < 0 with a message in qgs_last_error(); the model still works afterwards."""
    import torch
    from qgs_amd import _lib
    L = _lib.lib()
    g, m = _model('m36')
    nd = g.ndim
    ic = np.random.RandomState(1).rand(70, nd) * 0.01
    t = _grid(5)
    out = np.empty((70, nd, 6))
    good = np.array(m.rk_integrate(t, ic, 1, 1, B, C, A))
    vp = ctypes.c_void_p

    def expect_fail(rc, what):
        assert rc < 0, what
        msg = _lib.last_error()
        assert msg, what
        return msg

    # host-layout entry points: null handle, zero / negative sizes, bad direction, negative write_steps, empty time grid
    expect_fail(L.qgs_rk_integrate(None, 70, ic, t, len(t), 1, 1, 4, B, C, A, out), 'null model')
    expect_fail(L.qgs_rk_integrate(m._h, 0, ic, t, len(t), 1, 1, 4, B, C, A, out), 'n_traj 0')
    expect_fail(L.qgs_rk_integrate(m._h, -5, ic, t, len(t), 1, 1, 4, B, C, A, out), 'n_traj < 0')
    expect_fail(L.qgs_rk_integrate(m._h, 70, ic, t, 0, 1, 1, 4, B, C, A, out), 'n_time 0')
    expect_fail(L.qgs_rk_integrate(m._h, 70, ic, t, len(t), 0, 1, 4, B, C, A, out), 'direction 0')
    expect_fail(L.qgs_rk_integrate(m._h, 70, ic, t, len(t), 1, -1, 4, B, C, A, out), 'write_steps < 0')
    expect_fail(L.qgs_rk_integrate(m._h, 70, ic, t, len(t), 1, 1, 0, B, C, A, out), 'zero stages')
    tg = np.zeros((70, nd, 2))
    fm = np.empty((70, nd, 2, 6))
    expect_fail(L.qgs_rk_tgls_integrate(m._h, 70, 0, ic, tg, t, len(t), 1, 1, 4, B, C, A, 0, 1., out, fm), 'n_tg 0')
    expect_fail(L.qgs_rk_tgls_integrate(m._h, 70, 2, ic, tg, t, len(t), 2, 1, 4, B, C, A, 0, 1., out, fm), 'direction 2')
    expect_fail(L.qgs_tendencies(m._h, 0, ic, out), 'n_traj 0')
    expect_fail(L.qgs_tendencies(None, 1, ic, out), 'null model')
    expect_fail(L.qgs_jacobian(m._h, -1, ic, out), 'n_traj < 0')
    # device-layout entry points: ld not a multiple of 64, ld < n_traj, null pointers
    d_x = torch.zeros((nd, 128), dtype=torch.float64, device='cuda')
    d_r = torch.zeros((6, nd, 128), dtype=torch.float64, device='cuda')
    msg = expect_fail(L.qgs_rk_integrate_device(m._h, 70, 100, d_x.data_ptr(), t, len(t), 1, 1, 4, B, C, A, d_r.data_ptr(), None), 'ld 100')
    assert 'multiple of 64' in msg
    expect_fail(L.qgs_rk_integrate_device(m._h, 70, 64, d_x.data_ptr(), t, len(t), 1, 1, 4, B, C, A, d_r.data_ptr(), None), 'ld < n_traj')
    expect_fail(L.qgs_rk_integrate_device(m._h, 70, 128, None, t, len(t), 1, 1, 4, B, C, A, d_r.data_ptr(), None), 'null d_ic')
    expect_fail(L.qgs_rk_integrate_device(m._h, 70, 128, d_x.data_ptr(), t, len(t), 1, 1, 4, B, C, A, None, None), 'null d_rec')
    expect_fail(L.qgs_tendencies_device(m._h, 70, 96, d_x.data_ptr(), d_x.data_ptr(), None), 'ld 96')
    expect_fail(L.qgs_pack_states(m._h, 0, 64, d_x.data_ptr(), d_x.data_ptr(), None), 'n_traj 0')
    expect_fail(L.qgs_unpack_records(m._h, 70, 128, 0, 6, d_r.data_ptr(), d_r.data_ptr(), None), 'n_inner 0')
    expect_fail(L.qgs_batched_qr_device(m._h, 70, 128, 4, 5, d_r.data_ptr(), d_x.data_ptr(), None), 'cols > rows')
    expect_fail(L.qgs_ensemble_moments_device(m._h, 70, 128, 0, d_r.data_ptr(), d_x.data_ptr(), None, None), 'n_rows 0')
    # the small dense algebra of the covariant Lyapunov vectors
    pa, pb, pc = d_r[0].data_ptr(), d_r[1].data_ptr(), d_r[2].data_ptr()
    expect_fail(L.qgs_batched_matmul_device(m._h, 70, 128, 0, 4, 4, 0, 0, pa, pb, pc, None), 'n_rows 0')
    expect_fail(L.qgs_batched_matmul_device(m._h, 70, 128, 4, 4, 4, 0, 3, pa, pb, pc, None), 'triangular 3')
    expect_fail(L.qgs_batched_matmul_device(m._h, 70, 128, 3, 4, 4, 1, 1, pa, pb, pc, None), 'upper triangle of a 3 x 4 result')
    expect_fail(L.qgs_batched_matmul_device(m._h, 70, 128, 4, 3, 4, 0, 2, pa, pb, pc, None), 'triangular B 3 x 4')
    expect_fail(L.qgs_batched_matmul_device(m._h, 70, 128, 4, 4, 4, 0, 0, pa, pb, pa, None), 'result aliases A')
    expect_fail(L.qgs_batched_matmul_device(m._h, 70, 128, 4, 4, 4, 0, 0, pa, None, pc, None), 'null B')
    expect_fail(L.qgs_batched_matmul_device(m._h, 70, 100, 4, 4, 4, 0, 0, pa, pb, pc, None), 'ld 100')
    expect_fail(L.qgs_batched_matmul_device(None, 70, 128, 4, 4, 4, 0, 0, pa, pb, pc, None), 'null model')
    expect_fail(L.qgs_clv_backstep_device(m._h, 70, 128, 0, pa, pb, pc, d_x.data_ptr(), None, 0., None), 'n_vec 0')
    expect_fail(L.qgs_clv_backstep_device(m._h, 70, 128, 4, pa, pb, pb, d_x.data_ptr(), None, 0., None), 'result aliases the input')
    expect_fail(L.qgs_clv_backstep_device(m._h, 70, 128, 4, pa, pb, pc, None, None, 0., None), 'null norms')
    expect_fail(L.qgs_clv_backstep_device(m._h, 0, 128, 4, pa, pb, pc, d_x.data_ptr(), None, 0., None), 'n_traj 0')
    expect_fail(L.qgs_model_set_kernel(m._h, 7), 'kind 7')
    # model creation: rank, ndim, coordinates out of range, device out of range
    h = vp()
    coo, val = np.ascontiguousarray(g['coo'], dtype=np.int32), np.ascontiguousarray(g['val'])
    pc, pv = coo.ctypes.data_as(vp), val.ctypes.data_as(vp)
    expect_fail(L.qgs_model_create_rank(0, nd, 4, len(val), pc, pv, 0, None, None, ctypes.byref(h)), 'rank 4')
    expect_fail(L.qgs_model_create_rank(0, 0, 3, len(val), pc, pv, 0, None, None, ctypes.byref(h)), 'ndim 0')
    expect_fail(L.qgs_model_create_rank(0, nd - 1, 3, len(val), pc, pv, 0, None, None, ctypes.byref(h)), 'coordinate > ndim')
    expect_fail(L.qgs_model_create_rank(99, nd, 3, len(val), pc, pv, 0, None, None, ctypes.byref(h)), 'device 99')
    expect_fail(L.qgs_model_create_rank(0, nd, 3, -1, pc, pv, 0, None, None, ctypes.byref(h)), 'nnz < 0')
    expect_fail(L.qgs_model_create_rank(0, nd, 3, len(val), None, pv, 0, None, None, ctypes.byref(h)), 'null coo')
    assert not h.value
    devs = (ctypes.c_int * 2)(0, 99)
    expect_fail(L.qgs_group_create(2, devs, nd, 3, len(val), pc, pv, 0, None, None, ctypes.byref(h)), 'group device 99')
    expect_fail(L.qgs_group_create(0, devs, nd, 3, len(val), pc, pv, 0, None, None, ctypes.byref(h)), 'empty group')
    # a model without a Jacobian tensor refuses the tangent model
    from qgs_amd import _lib as lib_mod
    bare = lib_mod.HipModel(nd, g['coo'], g['val'])
    expect_fail(L.qgs_rk_tgls_integrate(bare._h, 70, 2, ic, tg, t, len(t), 1, 1, 4, B, C, A, 0, 1., out, fm), 'no jacobian')
    expect_fail(L.qgs_jacobian(bare._h, 1, ic, out), 'no jacobian, single state')
    bare.close()
    # round-4 entry points: record-window unpack (window beyond the record, zero sizes, null pointers) ...
    host = np.empty((70, nd, 6))
    hp = host.ctypes.data_as(vp)
    expect_fail(L.qgs_unpack_window(m._h, 70, 128, nd, 3, 6, 4, d_r.data_ptr(), hp, None), 'records 4..6 of 6')
    expect_fail(L.qgs_unpack_window(m._h, 70, 128, nd, 0, 6, 0, d_r.data_ptr(), hp, None), 'empty window')
    expect_fail(L.qgs_unpack_window(m._h, 70, 128, 0, 2, 6, 0, d_r.data_ptr(), hp, None), 'n_inner 0')
    expect_fail(L.qgs_unpack_window(m._h, 70, 128, nd, 2, 6, -1, d_r.data_ptr(), hp, None), 'first record < 0')
    expect_fail(L.qgs_unpack_window(m._h, 70, 100, nd, 2, 6, 0, d_r.data_ptr(), hp, None), 'ld 100')
    expect_fail(L.qgs_unpack_window(m._h, 70, 128, nd, 2, 6, 0, None, hp, None), 'null window')
    expect_fail(L.qgs_unpack_window(m._h, 70, 128, nd, 2, 6, 0, d_r.data_ptr(), None, None), 'null destination')
    expect_fail(L.qgs_unpack_window(None, 70, 128, nd, 2, 6, 0, d_r.data_ptr(), hp, None), 'null model')
    # (a good call: records 2..3 of the pageable block from the first two records of the window buffer)
    d_r.copy_(torch.arange(d_r.numel(), dtype=torch.float64, device='cuda').reshape(d_r.shape))
    host[:] = -1.0
    assert L.qgs_unpack_window(m._h, 70, 128, nd, 2, 6, 2, d_r.data_ptr(), hp, None) == 0
    torch.cuda.synchronize()
    want = d_r[:2, :, :70].cpu().numpy().transpose(2, 1, 0)
    assert np.array_equal(host[:, :, 2:4], want) and np.all(host[:, :, :2] == -1.0) and np.all(host[:, :, 4:] == -1.0)
    # ... and the general contraction (rank, result axes, slots, coordinates, device, null pointers)
    hc = vp()
    c3 = np.array([[0, 1, 2], [3, 0, 3]], dtype=np.int32)
    v3 = np.array([1.5, -2.0])
    p3, q3 = c3.ctypes.data_as(vp), v3.ctypes.data_as(vp)
    expect_fail(L.qgs_contraction_create(0, 4, 4, 1, 2, p3, q3, ctypes.byref(hc)), 'rank 4')
    expect_fail(L.qgs_contraction_create(0, 4, 3, 3, 2, p3, q3, ctypes.byref(hc)), '3 result axes')
    expect_fail(L.qgs_contraction_create(0, 0, 3, 1, 2, p3, q3, ctypes.byref(hc)), 'n_slots 0')
    expect_fail(L.qgs_contraction_create(0, 3, 3, 1, 2, p3, q3, ctypes.byref(hc)), 'coordinate 3 with 3 slots')
    expect_fail(L.qgs_contraction_create(99, 4, 3, 1, 2, p3, q3, ctypes.byref(hc)), 'device 99')
    expect_fail(L.qgs_contraction_create(0, 4, 3, 1, 2, None, q3, ctypes.byref(hc)), 'null coo')
    expect_fail(L.qgs_contraction_create(0, 4, 3, 1, -1, p3, q3, ctypes.byref(hc)), 'nnz < 0')
    assert not hc.value
    assert L.qgs_contraction_create(0, 4, 3, 1, 2, p3, q3, ctypes.byref(hc)) == 0
    vecs, res = np.array([[2., 3., 5., 7.], [11., 13., 17., 19.]]), np.empty(4)
    expect_fail(L.qgs_contraction_apply(None, vecs, res), 'null contraction')
    assert L.qgs_contraction_apply(hc, vecs, res) == 0
    assert np.array_equal(res, [3. * 17. * 1.5, 0., 0., 2. * 19. * -2.0])            # res[0] is the caller's to overwrite
    assert L.qgs_contraction_destroy(hc) == 0 and L.qgs_contraction_destroy(None) == 0
    # ... and the model is still usable
    assert np.array_equal(m.rk_integrate(t, ic, 1, 1, B, C, A), good)
    m.close()


def test_cache_miss_compiles_the_same_lds_stepper(tmp_path):
    """ndim 228 on an empty kernel cache: the LDS-resident stepper (requested with set_kernel(2)) is compiled by the
    out-of-process helper and must come out as the pre-built one: the hand-scheduled kernel of round 6 -- 256 registers (8 wavefronts
    per workgroup), its stage body allocated by the generator, and nothing spilled inside it: what the compiler spills in the frame
    code around the assembly statement is 32 - 36 B per lane (the compiler-scheduled kernel it replaces spilled 420 B); more than
    64 B means the frame has started to spill into the hot loop."""
    import json
    import subprocess
    import sys
    code = ("import os, sys, json, numpy as np, torch\n"
            "sys.path.insert(0, %r)\n"
            "from qgs_amd import _lib\n"
            "g = np.load(%r)\n"
            "ndim = int(g['ndim'])\n"
            "m = _lib.HipModel(ndim, g['coo'], g['val'], g['jcoo'], g['jval'])\n"
            "m.set_kernel(2)\n"
            "n = 256\n"
            "ic = torch.rand((ndim, n), dtype=torch.float64, device='cuda') * 0.01\n"
            "rec = torch.empty((1, ndim, n), dtype=torch.float64, device='cuda')\n"
            "t = np.arange(0., 0.35, 0.1)\n"
            "b = np.array([1 / 6, 1 / 3, 1 / 3, 1 / 6]); c = np.array([0., .5, .5, 1.]); a = np.zeros((4, 4)); a[1, 0] = .5; a[2, 1] = .5; a[3, 2] = 1.\n"
            "m.rk_integrate_device(n, n, ic.data_ptr(), t, 1, 0, b, c, a, rec.data_ptr(), torch.cuda.current_stream().cuda_stream)\n"
            "torch.cuda.synchronize()\n"
            "print('INFO ' + json.dumps(m.last_kernel_info()))\n"
            "print('FINITE %%d' %% int(torch.isfinite(rec).all().item()))\n"
            % (REPO, os.path.join(GOLDEN_DIR, 't228.npz')))
    p = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500,
                       env=dict(os.environ, QGS_HIP_CACHE_DIR=str(tmp_path)))
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    out = p.stdout.decode()
    info = json.loads([ln for ln in out.splitlines() if ln.startswith('INFO ')][0][5:])
    assert info['name'] == LDS_STEPPER, info
    assert info['vgprs'] <= 256 and info['scratch_bytes'] <= 64, info
    assert 'FINITE 1' in out


def test_cache_miss_compiles_the_batched_qr(tmp_path):
    """The shape-specialised QR kernels of the Lyapunov estimator on an empty cache: compiled on first use (helper process),
    published, correct (against np.linalg.qr), and found in the cache by the next process."""
    import json
    import subprocess
    import sys
    code = ("import os, sys, json, numpy as np, torch\n"
            "sys.path.insert(0, %r)\n"
            "from qgs_amd import _lib\n"
            "m = _lib.HipModel(2, np.array([[1, 0, 1]], dtype=np.int32), np.array([1.0]))\n"
            "rng = np.random.RandomState(0)\n"
            "worst = 0.0\n"
            "for rows, cols in ((36, 36), (20, 5)):\n"
            "    a = rng.randn(3, rows, cols)\n"
            "    d = torch.zeros((rows, cols, 64), dtype=torch.float64, device='cuda')\n"
            "    d[:, :, :3] = torch.from_numpy(np.ascontiguousarray(a.transpose(1, 2, 0))).cuda()\n"
            "    rd = torch.zeros((cols, 64), dtype=torch.float64, device='cuda')\n"
            "    m.batched_qr_device(3, 64, rows, cols, d.data_ptr(), rd.data_ptr())\n"
            "    torch.cuda.synchronize()\n"
            "    assert m.last_kernel_info()['name'] == 'qgs_spec_qr_%%dx%%d' %% (rows, cols)\n"
            "    q = d[:, :, :3].cpu().numpy().transpose(2, 0, 1)\n"
            "    for i in range(3):\n"
            "        worst = max(worst, float(np.abs(q[i] - np.linalg.qr(a[i])[0]).max()))\n"
            "print('WORST %%.3e' %% worst)\n"
            "print('FILES %%d' %% len([f for f in os.listdir(os.environ['QGS_HIP_CACHE_DIR']) if f.endswith('.hsaco')]))\n" % REPO)
    counts = []
    for _ in range(2):
        p = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900,
                           env=dict(os.environ, QGS_HIP_CACHE_DIR=str(tmp_path)))
        assert p.returncode == 0, p.stderr.decode()[-3000:]
        out = p.stdout.decode()
        assert float([ln for ln in out.splitlines() if ln.startswith('WORST ')][0][6:]) < 1e-13
        counts.append(int([ln for ln in out.splitlines() if ln.startswith('FILES ')][0][6:]))
    assert counts == [2, 2], counts


def test_record_windows_into_pageable_blocks_that_end_with_their_mapping():
    """`qgs_unpack_window` into ordinary (pageable) NumPy blocks at record offsets > 0, for block sizes whose data ends close
    to the end of its memory mapping.  Such blocks are pageable memory whatever hipPointerGetAttributes says about their
    addresses (the runtime's cached pins of earlier pageable copies make reused addresses look page-locked -- and read-only for
    the GPU when the pin belonged to a copy source): the library stores into host blocks only when it registered them itself,
    and strided rows reach pageable blocks through a page-locked bounce block."""
    import torch
    from qgs_amd import _lib
    L = _lib.lib()
    g, m = _model('m36')
    nd = g.ndim
    vp = ctypes.c_void_p
    rng = np.random.RandomState(0)
    for n, nrec in ((256, 64), (257, 61), (300, 57), (512, 32), (63, 509), (128, 128), (1000, 17), (64, 1024)):
        ld = (n + 63) // 64 * 64
        host = np.full((n, nd, nrec), -1.0)                               # mmap'ed by the allocator at these sizes
        w = 3
        win = torch.from_numpy(rng.rand(w, nd, ld)).cuda()
        for first in (nrec - w, nrec // 2, 1):
            assert L.qgs_unpack_window(m._h, n, ld, nd, w, nrec, first, win.data_ptr(), host.ctypes.data_as(vp), None) == 0, _lib.last_error()
            torch.cuda.synchronize()
            want = win[:, :, :n].cpu().numpy().transpose(2, 1, 0)
            assert np.array_equal(host[:, :, first:first + w], want)
        assert np.all(host[:, :, 0] == -1.0)
    m.close()


def test_enqueued_windows_drain_while_the_caller_goes_on():
    """`qgs_unpack_window_enqueue` + `qgs_drain_wait` (what the Lyapunov estimator's record windows use): several windows into three
    pageable destination arrays are handed over back to back -- each destination has its own staging block, so the host is not held
    for the transfer of the large one -- then one wait; every record is where the blocking `qgs_unpack_window` puts it.  Also a
    window whose rows are longer than a bounce block (one member, many records), and the bridge's transfer counters."""
    import torch
    from qgs_amd import _lib
    L = _lib.lib()
    g, m = _model('m36')
    nd = g.ndim
    rng = np.random.RandomState(7)
    n, nrec, w = 3000, 40, 8
    ld = (n + 63) // 64 * 64
    inner = (nd * 5, nd, 5)
    hosts = [np.full((n, q, nrec), -2.0) for q in inner]
    wins = []
    for first in range(0, nrec, w):
        ws = [torch.from_numpy(rng.rand(w, q, ld)).cuda() for q in inner]
        wins.append((first, ws))
        for t, q, host in zip(ws, inner, hosts):
            m.unpack_window_enqueue(n, ld, q, w, nrec, first, t.data_ptr(), host.ctypes.data)
    m.drain_wait()
    for first, ws in wins:
        for t, q, host in zip(ws, inner, hosts):
            assert np.array_equal(host[:, :, first:first + w], t[:, :, :n].cpu().numpy().transpose(2, 1, 0))
    # rows longer than a 16 MiB bounce block: 1 member x 36 variables x 2.2 M records in one window
    n1, nrec1 = 1, 2200000
    win = torch.from_numpy(rng.rand(nrec1, nd, 64)).cuda()
    host1 = np.zeros((n1, nd, nrec1))
    m.unpack_window_enqueue(n1, 64, nd, nrec1, nrec1, 0, win.data_ptr(), host1.ctypes.data)
    m.drain_wait()
    assert np.array_equal(host1[0], win[:, :, 0].cpu().numpy().T)
    m.close()


def test_kernel_clock_probe():
    """`qgs_kernel_clock`: the generated kernels note the shader-clock and the 100 MHz counters in workgroup 0; the last launch's
    clock lies in the range this GPU can run at and its interval is about the launch's duration.  Generic kernels carry no probe."""
    import torch
    from qgs_amd import _lib
    g, m = _model('m36')
    n, steps = 65536, 200
    ic = torch.from_numpy(np.random.RandomState(2).rand(g.ndim, n) * 0.01).cuda()
    rec = torch.empty((1, g.ndim, n), dtype=torch.float64, device='cuda')
    t = _grid(steps)
    m.set_kernel(2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    m.rk_integrate_device(n, n, ic.data_ptr(), t, 1, 0, B, C, A, rec.data_ptr())
    e0.record()
    m.rk_integrate_device(n, n, ic.data_ptr(), t, 1, 0, B, C, A, rec.data_ptr())
    e1.record()
    e1.synchronize()
    ghz, ms = m.kernel_clock()
    assert m.last_kernel_info()['name'] == 'qgs_spec_rk_s4'
    assert 0.8 < ghz < 2.6, ghz
    assert 0.5 * e0.elapsed_time(e1) < ms <= 1.05 * e0.elapsed_time(e1), (ms, e0.elapsed_time(e1))
    m.set_kernel(1)
    m.rk_integrate_device(n, n, ic.data_ptr(), t[:3], 1, 0, B, C, A, rec.data_ptr())
    torch.cuda.synchronize()
    assert m.kernel_clock() is None
    m.close()


def test_fp64_fma_rate_is_below_the_nominal_peak_and_plausible():
    """`qgs_fp64_fma_rate`: independent fp64 FMAs, eight wavefronts per SIMD, over ~ 20 ms: between a third of the nominal 78.6 TFLOP/s
    and the nominal peak itself (the board lowers the clock under this load; bench.py quotes its fp64 kernels against both)."""
    from qgs_amd import _lib
    tf, ms = _lib.fp64_fma_rate(0, 20.0)
    assert 25.0 < tf <= 78.6 * 1.02, tf
    assert 5.0 < ms < 200.0, ms
