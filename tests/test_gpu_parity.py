"""GPU: parity of the HIP path (through the C-ABI) against the goldens captured from the Python
reference and against the CPU oracle on seeded inputs.

Tolerances (fp64; the HIP kernels use FMA and a different summation order than the reference):
  f, Df                      |d| <= 1e-14 * max|ref|
  trajectories <= 100 steps  1e-12 relative
  trajectories 1000 steps    1e-10 relative (weakly chaotic amplification of rounding differences)
  tangent / adjoint 10 steps 1e-11 relative
"""
import json
import os

import numpy as np
from kernel_names import (LDS_STEPPER, LDS_STEPPER_RANK5, TGL_PAIR, TGL_PAIR_ASM, LDS_TANGENT, LDS_ADJOINT, LDS_TANGENT_RANK5, LDS_ADJOINT_RANK5,
                          LDS_TANGENT_ASM, LDS_ADJOINT_ASM, LDS_TANGENT_CC, LDS_ADJOINT_CC)
import pytest

from conftest import GOLDEN_DIR, REPO, RK4, load_golden, rel_err

pytestmark = pytest.mark.gpu

KINDS = {'auto': 0, 'generic': 1, 'specialised': 2}      # auto = wave-per-trajectory kernel for small ensembles


@pytest.fixture(scope='module')
def models():
    from qgs_amd import _lib
    cache = {}

    def get(name):
        if name not in cache:
            g = load_golden(name)
            cache[name] = _lib.HipModel(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
        return cache[name]
    yield get
    for m in cache.values():
        m.close()


def _kinds(model):
    return [k for k in KINDS if k != 'specialised' or model.specialised_available]


def test_backend_is_gfx950():
    from qgs_amd import _lib
    n, arch = _lib.backend_info()
    assert n >= 1 and arch.startswith('gfx950'), arch


@pytest.mark.parametrize('name', ['rp20', 'a36', 'm36', 't228', 'g30', 'd38', 'q38'])
def test_f_and_Df_vs_golden(models, name):
    g, m = load_golden(name), models(name)
    for kind in _kinds(m):
        m.set_kernel(KINDS[kind])
        assert rel_err(m.tendencies(g['fx_x']), g['fx_f']) < 1e-14, kind
        n = g['fx_Df'].shape[0]
        assert rel_err(m.jacobian(g['fx_x'][:n]), g['fx_Df']) < 1e-14, kind
        # single-state call keeps the reference's shapes
        assert m.tendencies(g['fx_x'][0]).shape == (g.ndim,)
        assert m.jacobian(g['fx_x'][0]).shape == (g.ndim, g.ndim)


@pytest.mark.parametrize('name', ['rp20', 'a36', 'm36', 't228', 'g30', 'd38', 'q38'])
def test_rk_cases_vs_golden(models, name):
    g, m = load_golden(name), models(name)
    for kind in _kinds(m):
        m.set_kernel(KINDS[kind])
        for cs in g.meta['rk_cases']:
            t = cs['tag']
            ic = g['rk_ic'][:cs['n_traj']]
            rec = m.rk_integrate(g['rk_%s_time' % t], ic, 1 if cs['forward'] else -1, cs['ws'], g['rk_%s_b' % t],
                                 g['rk_%s_c' % t], g['rk_%s_a' % t])
            tol = 1e-10 if cs['steps'] >= 1000 else 1e-12
            assert rel_err(rec, g['rk_%s_traj' % t]) < tol, (kind, t)


@pytest.mark.parametrize('name', ['rp20', 'a36', 'm36', 't228', 'g30', 'd38', 'q38'])
def test_tgls_cases_vs_golden(models, name):
    g, m = load_golden(name), models(name)
    for kind in _kinds(m):
        m.set_kernel(KINDS[kind])
        for cs in g.meta['tgls_cases']:
            t = cs['tag']
            rec, fm = m.rk_tgls_integrate(g['tgls_%s_time' % t], g['tgls_ic'], g['tgls_%s_tgic' % t],
                                          1 if cs['forward'] else -1, cs['ws'], g['tgls_%s_b' % t], g['tgls_%s_c' % t],
                                          g['tgls_%s_a' % t], cs['adjoint'], -1. if cs['inverse'] else 1.)
            assert rel_err(rec, g['tgls_%s_traj' % t]) < 1e-11, (kind, t)
            assert rel_err(fm, g['tgls_%s_fm' % t]) < 1e-11, (kind, t)


@pytest.mark.parametrize('n_traj', [1, 63, 64, 65, 1000])
def test_ragged_ensemble_sizes_vs_oracle(models, n_traj):
    """Wavefront tails: member counts around the 64-lane boundary."""
    from oracle.oracle import OracleModel
    g, m = load_golden('m36'), models('m36')
    ora = OracleModel(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    ic = np.random.RandomState(n_traj).rand(n_traj, g.ndim) * 0.01
    t = np.concatenate((np.arange(0., 2.0, 0.1), [2.0]))
    ref = ora.integrate_runge_kutta_jit(t, ic, 1, 7, RK4['b'], RK4['c'], RK4['a'], threads=4)
    for kind in _kinds(m):
        m.set_kernel(KINDS[kind])
        assert rel_err(m.tendencies(ic), ora.f(0., ic)) < 1e-14
        assert rel_err(m.rk_integrate(t, ic, 1, 7, RK4['b'], RK4['c'], RK4['a']), ref) < 1e-12, kind


@pytest.mark.parametrize('n_traj', [1, 65, 200])
def test_lds_resident_stepper_ndim228_vs_oracle(models, n_traj):
    """MAOOAM 6x6 (ndim 228): the JIT LDS-resident stepper (kind 2) and the generic tiled kernel (kind 1) against the
    oracle: forward with records, backward, a 2-stage and a 3-stage sub-diagonal tableau (the stepper takes the stage
    count at run time), and the trajectory pass of the tangent model (stage states stored by the same stepper)."""
    from oracle.oracle import OracleModel
    g, m = load_golden('t228'), models('t228')
    ora = OracleModel(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    rng = np.random.RandomState(100 + n_traj)
    ic = rng.rand(n_traj, g.ndim) * 0.01
    t = np.concatenate((np.arange(0., 1.2, 0.1), [1.2]))
    b3, c3 = np.array([1. / 6, 2. / 3, 1. / 6]), np.array([0., .5, 1.])
    a3 = np.zeros((3, 3)); a3[1, 0] = .5; a3[2, 1] = 1.          # sub-diagonal 3-stage scheme
    b2, c2 = np.array([0., 1.]), np.array([0., .5])
    a2 = np.zeros((2, 2)); a2[1, 0] = .5
    ak = np.array([[0., 0, 0], [.5, 0, 0], [-1., 2., 0]])               # Kutta's third-order scheme: not sub-diagonal
    cases = [(1, 5, RK4['b'], RK4['c'], RK4['a']), (-1, 1, RK4['b'], RK4['c'], RK4['a']), (1, 0, b2, c2, a2), (1, 4, b3, c3, a3),
             (1, 3, b3, c3, ak), (-1, 0, b3, c3, ak)]
    refs = [ora.integrate_runge_kutta_jit(t, ic, d, ws, b, c, a, threads=4) for d, ws, b, c, a in cases]
    tg = rng.randn(min(n_traj, 3), g.ndim, 2)
    rtr, rfm = ora.integrate_runge_kutta_tgls_jit(t[:5], ic[:tg.shape[0]], tg, 1, 2, RK4['b'], RK4['c'], RK4['a'], False, 1.)
    for kind, kname in ((1, 'gen_rk_tiled_kernel'), (2, LDS_STEPPER)):
        m.set_kernel(kind)
        for (d, ws, b, c, a), ref in zip(cases, refs):
            out = m.rk_integrate(t, ic, d, ws, b, c, a)
            dense = a is ak
            assert m.last_kernel_info()['name'] == (kname if not dense else ('gen_rk_kernel' if kind == 1 else 'qgs_spec_rkldsd16'))
            assert out.shape == ref.shape and rel_err(out, ref) < 1e-12, (kind, d, ws, len(b))
        tr, fm = m.rk_tgls_integrate(t[:5], ic[:tg.shape[0]], tg, 1, 2, RK4['b'], RK4['c'], RK4['a'], False, 1.)
        assert rel_err(tr, rtr) < 1e-12 and rel_err(fm, rfm) < 1e-11, kind
        if kind == 2:
            assert m.last_kernel_info()['name'] == LDS_TANGENT
    m.set_kernel(0)


@pytest.mark.parametrize('n_traj,n_tg', [(1, 5), (17, 4), (40, 7), (1, 300)])      # 228 x 300 > 65 535 (modes x columns)
def test_lds_resident_tangent_ndim228_vs_oracle(models, n_traj, n_tg):
    """MAOOAM 6x6: the JIT LDS-resident tangent and adjoint kernels (16 members x 4 columns per workgroup; ragged
    member and column tiles) against the oracle: tangent forward with records, adjoint backward with `inverse`."""
    from oracle.oracle import OracleModel
    g, m = load_golden('t228'), models('t228')
    ora = OracleModel(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    rng = np.random.RandomState(7 * n_traj + n_tg)
    ic = rng.rand(n_traj, g.ndim) * 0.01
    tg = rng.randn(n_traj, g.ndim, n_tg)
    t = np.concatenate((np.arange(0., 0.6, 0.1), [0.6]))
    b2, c2 = np.array([0., 1.]), np.array([0., .5])
    a2 = np.zeros((2, 2)); a2[1, 0] = .5
    cases = [(1, 2, RK4['b'], RK4['c'], RK4['a'], False, 1.), (-1, 1, RK4['b'], RK4['c'], RK4['a'], True, -1.),
             (1, 0, b2, c2, a2, True, 1.)]
    m.set_kernel(2)
    for d, ws, b, c, a, adj, inv in cases:
        rtr, rfm = ora.integrate_runge_kutta_tgls_jit(t, ic, tg, d, ws, b, c, a, adj, inv)
        tr, fm = m.rk_tgls_integrate(t, ic, tg, d, ws, b, c, a, adj, inv)
        assert m.last_kernel_info()['name'] == (LDS_ADJOINT if adj else LDS_TANGENT)
        assert fm.shape == rfm.shape and rel_err(tr, rtr) < 1e-12 and rel_err(fm, rfm) < 1e-11, (d, ws, adj, inv)
    m.set_kernel(0)


@pytest.mark.parametrize('name,n_traj', [('d38', 1), ('d38', 130), ('q38', 70)])
def test_rank5_models_vs_oracle(models, name, n_traj):
    """Rank-5 tensors (sparse_mul5 / sparse_mul4 path): the derived-monomial specialised kernels (dynamic T: 4 / 22 derived
    monomials, register-resident; T4: 111 / 516, kept in LDS by the LDS-resident kernels) and the 4-factor generic kernels
    against the oracle on seeded ensembles around the wavefront boundary: f, Df, RK4 with records, backward Heun,
    tangent and adjoint models."""
    from oracle.oracle import OracleModel
    g, m = load_golden(name), models(name)
    ora = OracleModel(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    assert ora.rank == 5
    rng = np.random.RandomState(n_traj)
    ic = rng.rand(n_traj, g.ndim) * 0.01
    ic[:, 10] += 1.5
    ic[:, 29] += 3.
    t = np.concatenate((np.arange(0., 1.0, 0.1), [1.0]))
    b2, c2 = np.array([.5, .5]), np.array([0., 1.])
    a2 = np.zeros((2, 2)); a2[1, 0] = 1.
    ref4 = ora.integrate_runge_kutta_jit(t, ic, 1, 3, RK4['b'], RK4['c'], RK4['a'], threads=4)
    ref2 = ora.integrate_runge_kutta_jit(t, ic, -1, 0, b2, c2, a2, threads=4)
    nt = min(n_traj, 4)
    tg = rng.randn(nt, g.ndim, 3)
    rtr, rfm = ora.integrate_runge_kutta_tgls_jit(t[:5], ic[:nt], tg, 1, 2, RK4['b'], RK4['c'], RK4['a'], False, 1.)
    atr, afm = ora.integrate_runge_kutta_tgls_jit(t[:5], ic[:nt], tg, -1, 1, RK4['b'], RK4['c'], RK4['a'], True, -1.)
    for kind in _kinds(m):
        m.set_kernel(KINDS[kind])
        assert rel_err(m.tendencies(ic), ora.f(0., ic)) < 1e-14, kind
        assert rel_err(m.jacobian(ic[:nt]), ora.Df(0., ic[:nt])) < 1e-14, kind
        assert rel_err(m.rk_integrate(t, ic, 1, 3, RK4['b'], RK4['c'], RK4['a']), ref4) < 1e-12, kind
        if kind == 'specialised':
            assert m.last_kernel_info()['name'] == ('qgs_spec_rk_s4' if name == 'd38' else LDS_STEPPER_RANK5)
        assert rel_err(m.rk_integrate(t, ic, -1, 0, b2, c2, a2), ref2) < 1e-12, kind
        tr, fm = m.rk_tgls_integrate(t[:5], ic[:nt], tg, 1, 2, RK4['b'], RK4['c'], RK4['a'], False, 1.)
        assert rel_err(tr, rtr) < 1e-12 and rel_err(fm, rfm) < 1e-11, kind
        if kind == 'specialised':
            assert m.last_kernel_info()['name'] == ('qgs_spec_tglp_s4' if name == 'd38' else LDS_TANGENT_RANK5)
        tr, fm = m.rk_tgls_integrate(t[:5], ic[:nt], tg, -1, 1, RK4['b'], RK4['c'], RK4['a'], True, -1.)
        assert rel_err(tr, atr) < 1e-12 and rel_err(fm, afm) < 1e-11, kind
        if kind == 'specialised':
            assert m.last_kernel_info()['name'] == ('qgs_spec_tglp_s4' if name == 'd38' else LDS_ADJOINT_RANK5)
    m.set_kernel(0)


@pytest.mark.parametrize('name', ['m36', 'rp20', 'd38'])
def test_general_tableaus_on_the_specialised_stepper(models, name):
    """Explicit schemes whose `a` is not sub-diagonal (the reference takes any b, c, a: integrate.py:214-219): Kutta's
    third-order scheme, the 3/8 rule, and a 4-stage tableau with a zero column (its stage tendencies are never stored).
    `qgs_spec_rkd_s<S>` keeps the tendencies in registers and the stage tendencies in a scratch array; against the oracle
    and the generic kernel, forward with records and backward."""
    from oracle.oracle import OracleModel
    g, m = load_golden(name), models(name)
    ora = OracleModel(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    kutta3 = (np.array([1. / 6, 2. / 3, 1. / 6]), np.array([0., .5, 1.]), np.array([[0., 0, 0], [.5, 0, 0], [-1., 2., 0]]))
    rule38 = (np.array([1., 3., 3., 1.]) / 8., np.array([0., 1. / 3, 2. / 3, 1.]),
              np.array([[0., 0, 0, 0], [1. / 3, 0, 0, 0], [-1. / 3, 1., 0, 0], [1., -1., 1., 0]]))
    sparse4 = (np.array([.25, 0., .5, .25]), np.array([0., .5, .5, 1.]),
               np.array([[0., 0, 0, 0], [.5, 0, 0, 0], [.5, 0, 0, 0], [.25, 0., .75, 0]]))     # k_2 never reused
    rng = np.random.RandomState(38)
    ic = rng.rand(70, g.ndim) * 0.01
    if name == 'd38':
        ic[:, 10] += 1.5
        ic[:, 29] += 3.
    t = np.concatenate((np.arange(0., 1.5, 0.1), [1.5]))
    for b, c, a in (kutta3, rule38, sparse4):
        for d, ws in ((1, 4), (-1, 0)):
            ref = ora.integrate_runge_kutta_jit(t, ic, d, ws, b, c, a, threads=4)
            m.set_kernel(2)
            out = m.rk_integrate(t, ic, d, ws, b, c, a)
            assert m.last_kernel_info()['name'] == 'qgs_spec_rkd_s%d' % len(b)
            assert rel_err(out, ref) < 1e-12, (len(b), d, ws)
            m.set_kernel(1)
            assert rel_err(m.rk_integrate(t, ic, d, ws, b, c, a), ref) < 1e-12
    # the tangent / adjoint model with such tableaus: stage-storing flavour of the same stepper + `qgs_spec_tgld_s<S>`
    tg = rng.randn(5, g.ndim, 6)
    for b, c, a in (kutta3, rule38):
        for d, ws, adj, inv in ((1, 2, False, 1.), (-1, 1, True, -1.)):
            rtr, rfm = ora.integrate_runge_kutta_tgls_jit(t[:7], ic[:5], tg, d, ws, b, c, a, adj, inv)
            m.set_kernel(2)
            tr, fm = m.rk_tgls_integrate(t[:7], ic[:5], tg, d, ws, b, c, a, adj, inv)
            assert m.last_kernel_info()['name'] == 'qgs_spec_tgld_s%d' % len(b)
            assert rel_err(tr, rtr) < 1e-12 and rel_err(fm, rfm) < 1e-11, (len(b), d, adj)
    m.set_kernel(0)


@pytest.mark.parametrize('variant', ['plain', 'split'])
@pytest.mark.parametrize('name', ['m36', 'rp20'])
def test_subdiagonal_tableaus_of_one_to_three_stages(models, monkeypatch, name, variant):
    """Euler, midpoint, Heun and Heun's third-order scheme on the fused stepper (`qgs_spec_rk_s<S>`, `qgs_spec_rkr_s<S>` when
    every step is a record): the last stage writes the new state in place (not for S = 1, where the stage input is the
    state itself), and the steps between two records are an inner loop -- record cadences that divide the run, that do not,
    that exceed it; forward and backward; against the oracle (integrate.py:182-223)."""
    from oracle.oracle import OracleModel
    g, m = load_golden(name), models(name)
    monkeypatch.setenv('QGS_HIP_RK_VARIANT', variant)               # one wavefront per 64 members / rows split over four
    ora = OracleModel(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    z = np.zeros
    euler = (np.array([1.]), z(1), z((1, 1)))
    a = z((2, 2)); a[1, 0] = .5
    midpoint = (np.array([0., 1.]), np.array([0., .5]), a)
    a = z((2, 2)); a[1, 0] = 1.
    heun = (np.array([.5, .5]), np.array([0., 1.]), a)
    a = z((3, 3)); a[1, 0] = 1. / 3; a[2, 1] = 2. / 3
    heun3 = (np.array([.25, 0., .75]), np.array([0., 1. / 3, 2. / 3]), a)
    ic = np.random.RandomState(7).rand(130, g.ndim) * 0.01
    t = np.concatenate((np.arange(0., 1.3, 0.1), [1.3]))            # 13 steps
    for b, c, a in (euler, midpoint, heun, heun3):
        for d in (1, -1):
            for ws in (0, 1, 4, 13, 50):
                ref = ora.integrate_runge_kutta_jit(t, ic, d, ws, b, c, a, threads=4)
                m.set_kernel(2)                                     # (re-reads the selection knobs)
                out = m.rk_integrate(t, ic, d, ws, b, c, a)
                kname = m.last_kernel_info()['name']
                if variant == 'plain':
                    assert kname == ('qgs_spec_rkr_s%d' if ws == 1 else 'qgs_spec_rk_s%d') % len(b), kname
                else:
                    assert kname.startswith('qgs_spec_rk'), kname
                assert out.shape == ref.shape and rel_err(out, ref) < 1e-13, (len(b), d, ws, kname)
    monkeypatch.delenv('QGS_HIP_RK_VARIANT')
    m.set_kernel(0)


def test_zero_steps_and_single_step(models):
    """n_time == 1 (no step): the only record is the initial condition (integrate.py:221)."""
    g, m = load_golden('a36'), models('a36')
    ic = g['rk_ic']
    for kind in _kinds(m):
        m.set_kernel(KINDS[kind])
        for ws in (0, 1, 4):
            rec = m.rk_integrate(np.array([0.]), ic, 1, ws, RK4['b'], RK4['c'], RK4['a'])
            assert rec.shape == (ic.shape[0], g.ndim, 1) and np.array_equal(rec[:, :, 0], ic)


def test_tgls_chunked_equals_unchunked(models, monkeypatch):
    """The stage-state buffer is processed in chunks of steps; chunking must not change the result."""
    g, m = load_golden('m36'), models('m36')
    ic = g['rk_ic'][:5]
    tg = np.random.RandomState(3).randn(5, g.ndim, 3)
    t = np.concatenate((np.arange(0., 2.3, 0.1), [2.3]))
    for kind in _kinds(m):
        m.set_kernel(KINDS[kind])
        monkeypatch.delenv('QGS_HIP_TGLS_CHUNK', raising=False)
        m.set_kernel(KINDS[kind])                            # (the selection knobs are read at set_kernel / model creation)
        a_tr, a_fm = m.rk_tgls_integrate(t, ic, tg, 1, 4, RK4['b'], RK4['c'], RK4['a'], False, 1.)
        monkeypatch.setenv('QGS_HIP_TGLS_CHUNK', '5')
        m.set_kernel(KINDS[kind])
        b_tr, b_fm = m.rk_tgls_integrate(t, ic, tg, 1, 4, RK4['b'], RK4['c'], RK4['a'], False, 1.)
        assert np.array_equal(a_tr, b_tr) and np.array_equal(a_fm, b_fm), kind
    monkeypatch.delenv('QGS_HIP_TGLS_CHUNK', raising=False)
    m.set_kernel(0)


def test_specialised_equals_generic_long_run(models):
    """Two independent HIP implementations agree on a 2000-step run of 4096 members."""
    g, m = load_golden('m36'), models('m36')
    ic = np.random.RandomState(5).rand(4096, g.ndim) * 0.01
    t = np.concatenate((np.arange(0., 200.0, 0.1), [200.0]))[:2001]
    m.set_kernel(1)
    a = m.rk_integrate(t, ic, 1, 0, RK4['b'], RK4['c'], RK4['a'])
    m.set_kernel(2)
    b = m.rk_integrate(t, ic, 1, 0, RK4['b'], RK4['c'], RK4['a'])
    assert rel_err(a, b) < 1e-9


def test_full_size_properties(models):
    """BASELINE config 2 size (65 536 members): size-independent properties.

    (1) the ensemble result is independent of the batch composition: any member integrated alone or inside
        the full batch gives bitwise the same answer (members never interact);
    (2) time reversal: integrating forward then backward over the same grid returns to the initial state
        to O(dt^4) truncation, well inside 1e-6 for 50 steps;
    (3) a sample of members agrees with the CPU oracle.
    """
    from oracle.oracle import OracleModel
    g, m = load_golden('m36'), models('m36')
    m.set_kernel(2)          # one kernel family for the bitwise batch-independence property
    n = 65536
    ic = np.random.RandomState(21217).rand(n, g.ndim) * 0.01
    t = np.concatenate((np.arange(0., 5.0, 0.1), [5.0]))
    full = m.rk_integrate(t, ic, 1, 0, RK4['b'], RK4['c'], RK4['a'])[:, :, 0]
    assert np.isfinite(full).all()
    pick = np.array([0, 1, 63, 64, 4097, 32768, 65535])
    alone = m.rk_integrate(t, ic[pick], 1, 0, RK4['b'], RK4['c'], RK4['a'])[:, :, 0]
    assert np.array_equal(alone, full[pick])
    m.set_kernel(0)          # automatic selection takes the wave-per-trajectory kernel for 7 members
    auto = m.rk_integrate(t, ic[pick], 1, 0, RK4['b'], RK4['c'], RK4['a'])[:, :, 0]
    assert rel_err(auto, alone) < 1e-12
    back = m.rk_integrate(t, full, -1, 0, RK4['b'], RK4['c'], RK4['a'])[:, :, 0]
    assert np.abs(back - ic).max() < 1e-6
    ora = OracleModel(g.ndim, g['coo'], g['val'])
    ref = ora.integrate_runge_kutta_jit(t, ic[pick], 1, 0, RK4['b'], RK4['c'], RK4['a'])[:, :, 0]
    assert rel_err(alone, ref) < 1e-12


def test_full_size_properties_config3(models):
    """BASELINE config 3 size (65 536 members of the 228-variable model, LDS-resident stepper): members integrated alone
    give bitwise the batch's answer, forward-then-backward returns to the start, a sample agrees with the oracle."""
    from oracle.oracle import OracleModel
    g, m = load_golden('t228'), models('t228')
    m.set_kernel(2)
    n = 65536
    ic = np.random.RandomState(3).rand(n, g.ndim) * 0.01
    t = np.concatenate((np.arange(0., 2.0, 0.1), [2.0]))
    full = m.rk_integrate(t, ic, 1, 0, RK4['b'], RK4['c'], RK4['a'])[:, :, 0]
    assert m.last_kernel_info()['name'] == LDS_STEPPER
    assert np.isfinite(full).all()
    pick = np.array([0, 15, 16, 63, 64, 1000, 32767, 65535])
    alone = m.rk_integrate(t, ic[pick], 1, 0, RK4['b'], RK4['c'], RK4['a'])[:, :, 0]
    assert np.array_equal(alone, full[pick])
    back = m.rk_integrate(t, full, -1, 0, RK4['b'], RK4['c'], RK4['a'])[:, :, 0]
    err = np.abs(back - ic).max()
    assert err < 5e-5                       # RK4 truncation of the fastest 6x6 modes at dt = 0.1 ...
    t2 = np.concatenate((np.arange(0., 2.0, 0.05), [2.0]))
    half = m.rk_integrate(t2, ic, 1, 0, RK4['b'], RK4['c'], RK4['a'])[:, :, 0]
    err2 = np.abs(m.rk_integrate(t2, half, -1, 0, RK4['b'], RK4['c'], RK4['a'])[:, :, 0] - ic).max()
    assert err2 < err / 12.                 # ... which falls as dt^4 (16x for dt / 2)
    ora = OracleModel(g.ndim, g['coo'], g['val'])
    ref = ora.integrate_runge_kutta_jit(t, ic[pick], 1, 0, RK4['b'], RK4['c'], RK4['a'])[:, :, 0]
    assert rel_err(alone, ref) < 1e-12


def test_full_size_properties_config4_config5(models):
    """BASELINE config 4 size (16 384 members x 36 tangent columns, MAOOAM-36) and one rank's share of config 5
    (131 072 members): batch independence (bitwise), the propagator applied to a vector equals the tangent run of that
    vector, <M u, w> = <u, M^T w> with the adjoint run, a sample against the oracle."""
    from oracle.oracle import OracleModel
    g, m = load_golden('m36'), models('m36')
    ora = OracleModel(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    m.set_kernel(2)
    n, nd = 16384, g.ndim
    rng = np.random.RandomState(4)
    ic = rng.rand(n, nd) * 0.01
    t = np.concatenate((np.arange(0., 1.0, 0.1), [1.0]))
    eye = np.ascontiguousarray(np.broadcast_to(np.eye(nd), (n, nd, nd)))
    tr, fm = m.rk_tgls_integrate(t, ic, eye, 1, 0, RK4['b'], RK4['c'], RK4['a'], False, 1.)
    tr, fm = tr[:, :, 0], fm[..., 0]
    assert np.isfinite(fm).all()
    pick = np.array([0, 63, 64, 5000, 16383])
    tr1, fm1 = m.rk_tgls_integrate(t, ic[pick], eye[pick], 1, 0, RK4['b'], RK4['c'], RK4['a'], False, 1.)
    assert np.array_equal(tr1[:, :, 0], tr[pick]) and np.array_equal(fm1[..., 0], fm[pick])
    u, w = rng.randn(len(pick), nd, 1), rng.randn(len(pick), nd, 1)
    _, mu = m.rk_tgls_integrate(t, ic[pick], u, 1, 0, RK4['b'], RK4['c'], RK4['a'], False, 1.)
    assert rel_err(mu[..., 0], fm[pick] @ u) < 1e-12
    # adjoint run: backwards in time along the same trajectory (integrate.py:555-614), started from the end state
    _, mtw = m.rk_tgls_integrate(t, tr[pick], w, -1, 0, RK4['b'], RK4['c'], RK4['a'], True, -1.)
    lhs = np.einsum('nik,nik->n', mu[..., 0], w)
    rhs = np.einsum('nik,nik->n', u, mtw[..., 0])
    assert np.abs(lhs - rhs).max() < 1e-6 * np.abs(lhs).max()          # exact up to the O(dt^4) mismatch of the two trajectories
    rtr, rfm = ora.integrate_runge_kutta_tgls_jit(t, ic[pick], eye[pick], 1, 0, RK4['b'], RK4['c'], RK4['a'], False, 1.)
    assert rel_err(tr[pick], rtr[:, :, 0]) < 1e-12 and rel_err(fm[pick], rfm[..., 0]) < 1e-11
    del fm, eye
    # config 5, one rank of eight
    n = 131072
    ic = rng.rand(n, nd) * 0.01
    t = np.concatenate((np.arange(0., 2.0, 0.1), [2.0]))
    full = m.rk_integrate(t, ic, 1, 0, RK4['b'], RK4['c'], RK4['a'])[:, :, 0]
    pick = np.array([0, 65535, 65536, 131071])
    alone = m.rk_integrate(t, ic[pick], 1, 0, RK4['b'], RK4['c'], RK4['a'])[:, :, 0]
    assert np.array_equal(alone, full[pick])
    ref = ora.integrate_runge_kutta_jit(t, ic[pick], 1, 0, RK4['b'], RK4['c'], RK4['a'])[:, :, 0]
    assert rel_err(alone, ref) < 1e-12


def test_linearity_of_tangent_model(models):
    """TL(a*u + b*v) = a*TL(u) + b*TL(v) along the same trajectory (columns are independent lanes)."""
    g, m = load_golden('m36'), models('m36')
    rng = np.random.RandomState(11)
    ic = g['rk_ic'][:4]
    u, v = rng.randn(4, g.ndim, 1), rng.randn(4, g.ndim, 1)
    tg = np.concatenate((u, v, 2.0 * u - 3.0 * v), axis=2)
    t = np.concatenate((np.arange(0., 1.0, 0.1), [1.0]))
    for kind in _kinds(m):
        m.set_kernel(KINDS[kind])
        _, fm = m.rk_tgls_integrate(t, ic, tg, 1, 0, RK4['b'], RK4['c'], RK4['a'], False, 1.)
        fm = fm[..., 0]
        assert rel_err(fm[:, :, 2], 2.0 * fm[:, :, 0] - 3.0 * fm[:, :, 1]) < 1e-13, kind


@pytest.mark.parametrize('env', [{'QGS_HIP_RK_VARIANT': 'plain'}, {'QGS_HIP_RK_VARIANT': 'split'},
                                 {'QGS_HIP_GENERIC': 'simple'}, {'QGS_HIP_WAVE_MAX_TRAJ': '0'},
                                 {'QGS_HIP_TGL_VARIANT': 'plain', 'QGS_HIP_WAVE_MAX_TRAJ': '0'},
                                 {'QGS_HIP_TGL_PAIR': '0', 'QGS_HIP_WAVE_MAX_TRAJ': '0'},
                                 {'QGS_HIP_TGL_SHARE_MIN_MB': '0', 'QGS_HIP_WAVE_MAX_TRAJ': '0'}])
def test_kernel_variants_agree_with_oracle(monkeypatch, env):
    """Every kernel-selection variant a normal build offers (plain / row-split stepper, simple generic kernel, lane kernels
    instead of the wavefront-per-trajectory ones, plain / paired / shared stage record of the tangent model) against the
    oracle on the same inputs.  (The generator's experiment knobs exist in developer builds only: `make DEV=1`.)"""
    from qgs_amd import _lib
    from oracle.oracle import OracleModel
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    g = load_golden('a36')
    m = _lib.HipModel(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])      # env is read at model creation
    ora = OracleModel(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    ic = np.random.RandomState(17).rand(130, g.ndim) * 0.01
    t = np.concatenate((np.arange(0., 3.0, 0.1), [3.0]))
    ref = ora.integrate_runge_kutta_jit(t, ic, -1, 4, RK4['b'], RK4['c'], RK4['a'], threads=4)
    tg = np.random.RandomState(18).randn(3, g.ndim, 2)
    rtr, rfm = ora.integrate_runge_kutta_tgls_jit(t[:8], ic[:3], tg, 1, 2, RK4['b'], RK4['c'], RK4['a'], True, -1.)
    for kind in (0, 1, 2):
        m.set_kernel(kind)
        assert rel_err(m.rk_integrate(t, ic, -1, 4, RK4['b'], RK4['c'], RK4['a']), ref) < 1e-12, (env, kind)
        tr, fm = m.rk_tgls_integrate(t[:8], ic[:3], tg, 1, 2, RK4['b'], RK4['c'], RK4['a'], True, -1.)
        assert rel_err(tr, rtr) < 1e-12 and rel_err(fm, rfm) < 1e-11, (env, kind)
    m.close()


@pytest.mark.parametrize('n_traj,n_tg', [(70, 5), (64, 36), (1, 2)])
def test_shared_stage_state_tangent_kernel(models, n_traj, n_tg):
    """`qgs_spec_tglx4_s<S>`: four columns of the same 64 members per workgroup, stage states prefetched through LDS.
    Ragged member blocks and column groups (wavefronts past the last column shadow it), records, backward adjoint with
    `inverse`, a 2-stage tableau; against the oracle and bitwise against the one-wavefront kernel it replaces."""
    from oracle.oracle import OracleModel
    import os
    os.environ['QGS_HIP_TGL_SHARE_MIN_MB'] = '0'          # in production only stage records beyond the Infinity Cache take it
    g, m = load_golden('m36'), models('m36')
    m.set_kernel(2)                                           # re-reads the selection knobs
    ora = OracleModel(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    rng = np.random.RandomState(1000 * n_traj + n_tg)
    ic = rng.rand(n_traj, g.ndim) * 0.01
    tg = rng.randn(n_traj, g.ndim, n_tg)
    t = np.concatenate((np.arange(0., 0.7, 0.1), [0.7]))
    b2, c2 = np.array([0., 1.]), np.array([0., .5])
    a2 = np.zeros((2, 2)); a2[1, 0] = .5
    cases = [(1, 2, RK4['b'], RK4['c'], RK4['a'], False, 1.), (-1, 1, RK4['b'], RK4['c'], RK4['a'], True, -1.), (1, 0, b2, c2, a2, False, 1.)]
    m.set_kernel(2)
    for d, ws, b, c, a, adj, inv in cases:
        rtr, rfm = ora.integrate_runge_kutta_tgls_jit(t, ic, tg, d, ws, b, c, a, adj, inv)
        tr, fm = m.rk_tgls_integrate(t, ic, tg, d, ws, b, c, a, adj, inv)
        assert m.last_kernel_info()['name'] == 'qgs_spec_tglx4_s%d' % len(b)
        assert rel_err(tr, rtr) < 1e-12 and rel_err(fm, rfm) < 1e-11, (d, ws, adj)
        os.environ['QGS_HIP_TGL_VARIANT'] = 'plain'
        try:
            m.set_kernel(2)                                   # re-reads the selection knobs
            _, fm1 = m.rk_tgls_integrate(t, ic, tg, d, ws, b, c, a, adj, inv)
            assert m.last_kernel_info()['name'] == TGL_PAIR % len(b)      # (stage record in mode pairs)
        finally:
            del os.environ['QGS_HIP_TGL_VARIANT']
            m.set_kernel(2)
        assert np.array_equal(fm, fm1)                       # same arithmetic, only the way the stage states arrive differs
    del os.environ['QGS_HIP_TGL_SHARE_MIN_MB']
    m.set_kernel(0)                                           # back to the defaults


def test_lds_resident_kernels_on_a_second_tensor_ndim72():
    """The LDS-resident generators on another tensor: atmosphere 4x4 with orography (ndim 72, between the
    register-resident limit of 64 and MAOOAM 6x6).  Stepper, tangent and adjoint against the oracle."""
    import model_configs
    from qgs_amd.functions.tendencies import create_tendencies
    from oracle.oracle import OracleModel
    f, Df = create_tendencies(model_configs.params_a72())
    assert f.ndim == 72
    m = f.hip_model()
    ora = OracleModel(f.ndim, f.coo, f.val, Df.coo, Df.val)
    rng = np.random.RandomState(72)
    ic = rng.rand(70, f.ndim) * 0.05
    t = np.concatenate((np.arange(0., 1., 0.1), [1.]))
    ref = ora.integrate_runge_kutta_jit(t, ic, 1, 3, RK4['b'], RK4['c'], RK4['a'], threads=4)
    tg = rng.randn(9, f.ndim, 6)
    for kind, names in ((1, ('gen_rk_tiled_kernel', None)), (2, (LDS_STEPPER, LDS_TANGENT))):
        m.set_kernel(kind)
        out = m.rk_integrate(t, ic, 1, 3, RK4['b'], RK4['c'], RK4['a'])
        assert m.last_kernel_info()['name'] == names[0]
        assert rel_err(out, ref) < 1e-12, kind
        for adj in (False, True):
            rtr, rfm = ora.integrate_runge_kutta_tgls_jit(t[:6], ic[:9], tg, 1, 1, RK4['b'], RK4['c'], RK4['a'], adj, 1.)
            tr, fm = m.rk_tgls_integrate(t[:6], ic[:9], tg, 1, 1, RK4['b'], RK4['c'], RK4['a'], adj, 1.)
            assert rel_err(tr, rtr) < 1e-12 and rel_err(fm, rfm) < 1e-11, (kind, adj)
        if names[1]:
            assert m.last_kernel_info()['name'] == LDS_ADJOINT
    f.operands.release()


@pytest.mark.parametrize('tile_members', [16, 8])
def test_lds_resident_kernels_rank5_ndim106(monkeypatch, tile_members):
    """Dynamic-T MAOOAM 4x4 / 4x4 (ndim 106, rank-5 tensor, 4 / 56 derived monomials): the LDS-resident stepper, tangent and
    adjoint kernels with the derived monomials as LDS nodes, and the 4-factor generic kernels, against the oracle.  The
    tangent kernels in both tile shapes: 16 members x 4 columns and 8 x 8 (chosen automatically when the stage state and
    its derived monomials of 16 members do not fit the LDS, e.g. dynamic-T MAOOAM 6x6)."""
    from model_configs import params_d106
    if tile_members == 8:
        monkeypatch.setenv('QGS_HIP_LDS_TGL_MEMBERS', '8')
    from oracle.oracle import OracleModel
    from qgs_amd.functions.tendencies import create_tendencies
    p = params_d106()
    f, Df = create_tendencies(p)
    assert f.ndim == 106 and f.coo.shape[1] == 5
    m = f.hip_model()
    ora = OracleModel(f.ndim, f.coo, f.val, Df.coo, Df.val)
    vr = p.variables_range
    rng = np.random.RandomState(106)
    ic = rng.rand(70, f.ndim) * 0.01
    ic[:, vr[0]] += 1.5                                   # T_a,0
    ic[:, vr[2]] += 3.                                    # T_o,0
    t = np.concatenate((np.arange(0., 1., 0.1), [1.]))
    ref = ora.integrate_runge_kutta_jit(t, ic, 1, 3, RK4['b'], RK4['c'], RK4['a'], threads=4)
    tg = rng.randn(9, f.ndim, 6)
    sfx = 'm8' if tile_members == 8 else ''
    for kind, names in ((1, ('gen_rk_kernel', None)), (2, (LDS_STEPPER_RANK5, LDS_TANGENT_RANK5 + sfx))):
        m.set_kernel(kind)
        assert rel_err(m.tendencies(ic), ora.f(0., ic)) < 1e-14, kind
        out = m.rk_integrate(t, ic, 1, 3, RK4['b'], RK4['c'], RK4['a'])
        assert m.last_kernel_info()['name'] == names[0]
        assert rel_err(out, ref) < 1e-12, kind
        for adj in (False, True):
            rtr, rfm = ora.integrate_runge_kutta_tgls_jit(t[:6], ic[:9], tg, 1, 1, RK4['b'], RK4['c'], RK4['a'], adj, 1.)
            tr, fm = m.rk_tgls_integrate(t[:6], ic[:9], tg, 1, 1, RK4['b'], RK4['c'], RK4['a'], adj, 1.)
            assert rel_err(tr, rtr) < 1e-12 and rel_err(fm, rfm) < 1e-11, (kind, adj)
        if names[1]:
            assert m.last_kernel_info()['name'] == LDS_ADJOINT_RANK5 + sfx
    f.operands.release()


def test_kernel_selection_at_the_baseline_configurations(models):
    """Which kernel the library picks at the sizes of BASELINE.json (a guard against silently falling onto a slower path):
    config 2 (65 536 members, MAOOAM-36) -> the plain register-resident stepper; smaller ensembles -> 4-way row split /
    wavefront-per-trajectory; config 3 (ndim 228) -> the LDS-resident stepper; config 4 (16 384 members x 36 columns) -> the
    one-wavefront tangent kernel, 65 536 members -> the shared-stage-state one; rank-5 models -> register-resident
    (dynamic T) / LDS-resident (T4) kernels; a non-sub-diagonal tableau -> the general-tableau stepper."""
    import torch
    from qgs_amd import _lib  # noqa: F401
    st = torch.cuda.current_stream().cuda_stream
    t2 = np.array([0., 0.1, 0.2])

    def stepper_name(m, ndim, n, b=RK4['b'], c=RK4['c'], a=RK4['a']):
        ld = (n + 63) // 64 * 64
        ic = torch.full((ndim, ld), 0.01, dtype=torch.float64, device='cuda')
        rec = torch.empty((1, ndim, ld), dtype=torch.float64, device='cuda')
        m.rk_integrate_device(n, ld, ic.data_ptr(), t2, 1, 0, b, c, a, rec.data_ptr(), st)
        torch.cuda.synchronize()
        return m.last_kernel_info()['name']

    def tangent_name(m, ndim, n, n_tg):
        ic = torch.full((ndim, n), 0.01, dtype=torch.float64, device='cuda')
        tg = torch.zeros((ndim, n_tg, n), dtype=torch.float64, device='cuda')
        rec = torch.empty((1, ndim, n), dtype=torch.float64, device='cuda')
        recm = torch.empty((1, ndim, n_tg, n), dtype=torch.float64, device='cuda')
        tt = np.concatenate((np.arange(0., 0.1 - 1e-12, 0.01), [0.1]))
        m.rk_tgls_integrate_device(n, n, n_tg, ic.data_ptr(), tg.data_ptr(), tt, 1, 0, RK4['b'], RK4['c'], RK4['a'], False, 1.,
                                   rec.data_ptr(), recm.data_ptr(), st)
        torch.cuda.synchronize()
        return m.last_kernel_info()['name']

    m36 = models('m36')
    m36.set_kernel(0)
    assert stepper_name(m36, 36, 65536) == 'qgs_spec_rk_s4'
    assert stepper_name(m36, 36, 16384) == 'qgs_spec_rksplit4_s4'
    assert stepper_name(m36, 36, 64) == 'gen_rk_wave_kernel'
    assert tangent_name(m36, 36, 16384, 36) == TGL_PAIR % 4          # fed by qgs_spec_rkstagesp_s4: stage record in mode pairs
    assert tangent_name(m36, 36, 1024, 36) == 'qgs_spec_tgl_s4'            # fed by the wave-per-trajectory stepper: plain record
    assert tangent_name(m36, 36, 65536, 36) == 'qgs_spec_tglx4_s4'
    r38 = (np.array([1., 3., 3., 1.]) / 8., np.array([0., 1. / 3, 2. / 3, 1.]),
           np.array([[0., 0, 0, 0], [1. / 3, 0, 0, 0], [-1. / 3, 1., 0, 0], [1., -1., 1., 0]]))
    assert stepper_name(m36, 36, 65536, *r38) == 'qgs_spec_rkd_s4'
    t228 = models('t228')
    t228.set_kernel(0)
    assert stepper_name(t228, 228, 4096) == LDS_STEPPER
    assert stepper_name(t228, 228, 1) == 'gen_rk_wave_kernel'        # lane-group variant (4 lanes per row)
    d38, q38 = models('d38'), models('q38')
    d38.set_kernel(0)
    q38.set_kernel(0)
    assert stepper_name(d38, 38, 65536) == 'qgs_spec_rk_s4'
    assert stepper_name(q38, 38, 4096) == LDS_STEPPER_RANK5
    assert stepper_name(q38, 38, 1) == 'gen_rk_wave_kernel'
    assert stepper_name(d38, 38, 1) == 'gen_rk_wave_kernel'


def _random_system(seed, ndim, rank, nnz):
    """A random polynomial system in the tensor form of the reference: coordinates (i, j, k[, l, m]) with i >= 1, index 0 =
    the constant slot, duplicates and repeated factors allowed; the Jacobian tensor is built the way the reference does it
    (jacobian_from_tensor, qgtensor.py:700-722: the tensor plus its copies with axis 1 swapped with every later axis)."""
    rng = np.random.RandomState(seed)
    coo = rng.randint(0, ndim + 1, size=(nnz, rank))
    coo[:, 0] = rng.randint(1, ndim + 1, size=nnz)
    coo[: nnz // 4, 2:] = 0                                      # a good share of linear / constant terms
    val = rng.randn(nnz) * 0.2
    jc, jv = [coo], [val]
    for ax in range(2, rank):
        sw = coo.copy()
        sw[:, [1, ax]] = sw[:, [ax, 1]]
        jc.append(sw)
        jv.append(val)
    return coo.astype(np.int32), val, np.vstack(jc).astype(np.int32), np.concatenate(jv)


@pytest.mark.parametrize('seed,ndim,rank,nnz', [(1, 5, 3, 40), (2, 11, 3, 150), (3, 4, 5, 60), (4, 9, 5, 200), (5, 70, 3, 900), (6, 66, 5, 700), (7, 66, 5, 120),
                                                 (8, 1, 3, 3), (9, 64, 3, 600), (10, 65, 3, 600), (11, 30, 3, 3000), (12, 2, 5, 12), (13, 130, 3, 4000)])
def test_random_polynomial_systems(seed, ndim, rank, nnz):
    """The code generators and the generic kernels on tensors that do not come from a qgs model: random coordinates with
    duplicate entries, squares and higher powers of one variable, rows without terms.  f, Df, RK4 (records, backward), a
    general tableau and the tangent / adjoint model against the oracle, in every kernel family the size allows
    (register-resident up to 64 variables, LDS-resident above)."""
    from qgs_amd import _lib
    from oracle.oracle import OracleModel
    coo, val, jcoo, jval = _random_system(seed, ndim, rank, nnz)
    m = _lib.HipModel(ndim, coo, val, jcoo, jval)
    ora = OracleModel(ndim, coo, val, jcoo, jval)
    rng = np.random.RandomState(100 + seed)
    x = rng.rand(70, ndim) * 0.3
    t = np.concatenate((np.arange(0., 0.05, 0.01), [0.05]))
    kutta3 = (np.array([1. / 6, 2. / 3, 1. / 6]), np.array([0., .5, 1.]), np.array([[0., 0, 0], [.5, 0, 0], [-1., 2., 0]]))
    tg = rng.randn(6, ndim, 3)
    ref_f, ref_J = ora.f(0., x), ora.Df(0., x[:6])
    ref_rk = ora.integrate_runge_kutta_jit(t, x, 1, 2, RK4['b'], RK4['c'], RK4['a'])
    ref_rkb = ora.integrate_runge_kutta_jit(t, x, -1, 0, RK4['b'], RK4['c'], RK4['a'])
    ref_k3 = ora.integrate_runge_kutta_jit(t, x, 1, 0, *kutta3)
    ref_tg = ora.integrate_runge_kutta_tgls_jit(t, x[:6], tg, 1, 1, RK4['b'], RK4['c'], RK4['a'], False, 1.)
    ref_ad = ora.integrate_runge_kutta_tgls_jit(t, x[:6], tg, -1, 0, RK4['b'], RK4['c'], RK4['a'], True, -1.)
    # seed 6: ~600 distinct quartic monomials do not fit the LDS next to 66 variables -> generic kernels only
    assert m.specialised_available == (seed != 6)
    for kind in ((0, 1, 2) if m.specialised_available else (0, 1)):
        m.set_kernel(kind)
        assert rel_err(m.tendencies(x), ref_f) < 1e-13, kind
        assert np.abs(m.jacobian(x[:6]) - ref_J).max() < 1e-13 * max(1., np.abs(ref_J).max()), kind
        assert rel_err(m.rk_integrate(t, x, 1, 2, RK4['b'], RK4['c'], RK4['a']), ref_rk) < 1e-12, kind
        assert rel_err(m.rk_integrate(t, x, -1, 0, RK4['b'], RK4['c'], RK4['a']), ref_rkb) < 1e-12, kind
        assert rel_err(m.rk_integrate(t, x, 1, 0, *kutta3), ref_k3) < 1e-12, kind
        tr, fm = m.rk_tgls_integrate(t, x[:6], tg, 1, 1, RK4['b'], RK4['c'], RK4['a'], False, 1.)
        assert rel_err(tr, ref_tg[0]) < 1e-12 and rel_err(fm, ref_tg[1]) < 1e-11, kind
        tr, fm = m.rk_tgls_integrate(t, x[:6], tg, -1, 0, RK4['b'], RK4['c'], RK4['a'], True, -1.)
        assert rel_err(tr, ref_ad[0]) < 1e-12 and rel_err(fm, ref_ad[1]) < 1e-11, kind
    m.close()


@pytest.mark.parametrize('seed,ndim', [(21, 7), (22, 12), (23, 1)])
def test_paired_stage_record_equals_plain(monkeypatch, seed, ndim):
    """`qgs_spec_rkstagesp_s<S>` -> `qgs_spec_tglp_s<S>` exchange the stage record in mode pairs (128-bit accesses, an odd last
    mode on its own); the arithmetic is the same as with the plain record, so the results must be bitwise equal -- odd and even
    ndim, ensembles that do not fill their last wavefront."""
    from qgs_amd import _lib
    coo, val, jcoo, jval = _random_system(seed, ndim, 3, 12 * ndim)
    rng = np.random.RandomState(seed)
    x = rng.rand(150, ndim) * 0.3
    tg = rng.randn(150, ndim, 3)
    t = np.concatenate((np.arange(0., 0.05, 0.01), [0.05]))
    out = {}
    for pair in ('1', '0'):
        monkeypatch.setenv('QGS_HIP_TGL_PAIR', pair)
        m = _lib.HipModel(ndim, coo, val, jcoo, jval)
        m.set_kernel(2)
        res = []
        for d, ws, adj, inv in ((1, 1, False, 1.), (-1, 2, True, -1.)):
            res.append(m.rk_tgls_integrate(t, x, tg, d, ws, RK4['b'], RK4['c'], RK4['a'], adj, inv))
            assert m.last_kernel_info()['name'] == (TGL_PAIR % 4 if pair == '1' else 'qgs_spec_tgl_s4')
        out[pair] = res
        m.close()
    for (tr1, fm1), (tr0, fm0) in zip(out['1'], out['0']):
        assert np.array_equal(tr1, tr0) and np.array_equal(fm1, fm0)


@pytest.mark.parametrize('seed,ndim,stages', [(31, 36, 4), (32, 7, 4), (33, 12, 2), (34, 5, 3)])
def test_hand_scheduled_tangent_kernel_equals_the_compiler_scheduled_one(monkeypatch, seed, ndim, stages):
    """`qgs_spec_tglpa_s<S>` (codegen_tangent_asm.cpp: registers allocated by the generator, the next stage state prefetched into the
    accumulation registers, step-start vector and running sum in LDS, coefficients through a register ring + DPP broadcast, all steps
    between two records one assembly loop) does the arithmetic of `qgs_spec_tglp_s<S>` in the same order: the results must be
    BITWISE equal -- tangent and adjoint, forward and backward, records every step / every 2 / none, odd and even ndim, 2 - 4 stages,
    ensembles that do not fill their last wavefront."""
    from qgs_amd import _lib
    coo, val, jcoo, jval = _random_system(seed, ndim, 3, 12 * ndim)
    rng = np.random.RandomState(seed)
    x = rng.rand(150, ndim) * 0.3
    tg = rng.randn(150, ndim, 3)
    t = np.concatenate((np.arange(0., 0.07, 0.01), [0.07]))
    if stages == 4:
        b, c, a = RK4['b'], RK4['c'], RK4['a']
    elif stages == 2:
        b, c = np.array([0., 1.]), np.array([0., .5]); a = np.zeros((2, 2)); a[1, 0] = .5
    else:
        b, c = np.array([1. / 6, 2. / 3, 1. / 6]), np.array([0., .5, 1.]); a = np.zeros((3, 3)); a[1, 0] = .5; a[2, 1] = 1.
    out = {}
    for asm in ('1', '0'):
        monkeypatch.setenv('QGS_HIP_TGL_ASM', asm)
        m = _lib.HipModel(ndim, coo, val, jcoo, jval)
        m.set_kernel(2)
        res = []
        for d, ws, adj, inv in ((1, 1, False, 1.), (-1, 2, True, -1.), (1, 0, False, 1.), (1, 3, True, 1.)):
            res.append(m.rk_tgls_integrate(t, x, tg, d, ws, b, c, a, adj, inv))
            assert m.last_kernel_info()['name'] == (TGL_PAIR_ASM if asm == '1' else TGL_PAIR) % stages
        out[asm] = res
        m.close()
    for (tr1, fm1), (tr0, fm0) in zip(out['1'], out['0']):
        assert np.array_equal(tr1, tr0) and np.array_equal(fm1, fm0)


@pytest.mark.parametrize('tensor,n_traj,n_tg', [('t228', 21, 6), ('t228', 1, 3), ('a72', 40, 9)])
def test_hand_scheduled_lds_tangent_kernels_vs_oracle_and_their_twins(monkeypatch, tensor, n_traj, n_tg):
    """`qgs_spec_tglldsa8` / `qgs_spec_adjldsa8` (codegen_lds_asm.cpp: the stage body of the hand-scheduled LDS stepper in the frame of
    the LDS-resident tangent kernels) and the compiler-scheduled `qgs_spec_tgllds16` / `qgs_spec_adjlds16`, each selected by
    QGS_HIP_LDS_TGL_ASM, against the oracle and against each other: tangent forward with records, adjoint backward with `inverse`,
    a 2-stage scheme without records, a 3-stage scheme with a record every third step; ragged member and column tiles."""
    from qgs_amd import _lib
    from oracle.oracle import OracleModel
    if tensor == 'a72':
        import model_configs
        from qgs_amd.functions.tendencies import create_tendencies
        f, Df = create_tendencies(model_configs.params_a72())
        ndim, coo, val, jcoo, jval = f.ndim, f.coo, f.val, Df.coo, Df.val
        f.operands.release()
    else:
        g = load_golden(tensor)
        ndim, coo, val, jcoo, jval = g.ndim, g['coo'], g['val'], g['jcoo'], g['jval']
    ora = OracleModel(ndim, coo, val, jcoo, jval)
    rng = np.random.RandomState(5 * n_traj + n_tg)
    ic = rng.rand(n_traj, ndim) * 0.01
    tg = rng.randn(n_traj, ndim, n_tg)
    t = np.concatenate((np.arange(0., 0.6, 0.1), [0.6]))
    b2, c2 = np.array([0., 1.]), np.array([0., .5])
    a2 = np.zeros((2, 2)); a2[1, 0] = .5
    b3, c3 = np.array([1. / 6, 2. / 3, 1. / 6]), np.array([0., .5, 1.])
    a3 = np.zeros((3, 3)); a3[1, 0] = .5; a3[2, 1] = 1.
    cases = [(1, 2, RK4['b'], RK4['c'], RK4['a'], False, 1.), (-1, 1, RK4['b'], RK4['c'], RK4['a'], True, -1.),
             (1, 0, b2, c2, a2, True, 1.), (1, 0, RK4['b'], RK4['c'], RK4['a'], False, 1.), (1, 3, b3, c3, a3, False, 1.)]
    refs = [ora.integrate_runge_kutta_tgls_jit(t, ic, tg, d, ws, b, c, a, adj, inv) for d, ws, b, c, a, adj, inv in cases]
    out = {}
    for asm in ('1', '0'):
        monkeypatch.setenv('QGS_HIP_LDS_TGL_ASM', asm)
        m = _lib.HipModel(ndim, coo, val, jcoo, jval)
        m.set_kernel(2)
        res = []
        for (d, ws, b, c, a, adj, inv), (rtr, rfm) in zip(cases, refs):
            tr, fm = m.rk_tgls_integrate(t, ic, tg, d, ws, b, c, a, adj, inv)
            names = (LDS_TANGENT_ASM, LDS_ADJOINT_ASM) if asm == '1' else (LDS_TANGENT_CC, LDS_ADJOINT_CC)
            assert m.last_kernel_info()['name'] == names[1 if adj else 0]
            assert fm.shape == rfm.shape and rel_err(tr, rtr) < 1e-12 and rel_err(fm, rfm) < 1e-11, (asm, d, ws, adj, inv)
            res.append(fm)
        out[asm] = res
        m.close()
    for fm1, fm0 in zip(out['1'], out['0']):
        assert rel_err(fm1, fm0) < 1e-12


def test_cache_miss_compiles_the_same_stepper(tmp_path):
    """A tensor that misses the kernel cache is compiled at model-creation time by the out-of-process helper (system hiprtc),
    not by whatever hiprtc this -- torch-importing -- process has mapped: the stepper must come out with the registers of the
    pre-built one (282; PyTorch's bundled compiler gives 324) and without scratch."""
    import subprocess
    import sys
    code = ("import os, sys, json, numpy as np, torch\n"
            "sys.path.insert(0, %r)\n"
            "from qgs_amd import _lib\n"
            "g = np.load(%r)\n"
            "ndim = int(g['ndim'])\n"
            "m = _lib.HipModel(ndim, g['coo'], g['val'], g['jcoo'], g['jval'])\n"
            "n = 65536\n"
            "ic = torch.rand((ndim, n), dtype=torch.float64, device='cuda') * 0.01\n"
            "rec = torch.empty((1, ndim, n), dtype=torch.float64, device='cuda')\n"
            "t = np.arange(0., 0.55, 0.1)\n"
            "b = np.array([1 / 6, 1 / 3, 1 / 3, 1 / 6]); c = np.array([0., .5, .5, 1.]); a = np.zeros((4, 4)); a[1, 0] = .5; a[2, 1] = .5; a[3, 2] = 1.\n"
            "m.rk_integrate_device(n, n, ic.data_ptr(), t, 1, 0, b, c, a, rec.data_ptr(), torch.cuda.current_stream().cuda_stream)\n"
            "torch.cuda.synchronize()\n"
            "print('INFO ' + json.dumps(m.last_kernel_info()))\n"
            "print('FILES %%d' %% len([f for f in os.listdir(os.environ['QGS_HIP_CACHE_DIR']) if f.endswith('.hsaco')]))\n"
            % (REPO, os.path.join(GOLDEN_DIR, 'm36.npz')))
    p = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900,
                       env=dict(os.environ, QGS_HIP_CACHE_DIR=str(tmp_path)))
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    out = p.stdout.decode()
    info = json.loads([ln for ln in out.splitlines() if ln.startswith('INFO ')][0][5:])
    assert info['name'] == 'qgs_spec_rk_s4'
    assert info['vgprs'] <= 288 and info['scratch_bytes'] == 0, info
    assert int([ln for ln in out.splitlines() if ln.startswith('FILES ')][0][6:]) >= 1
