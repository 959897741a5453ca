"""GPU: the drop-in Python API (create_tendencies objects, functional integrators, integrator classes)
against API-level goldens captured from the reference (shapes, time axes, squeezing, tg_ic conventions)."""
import os
import pickle

import numpy as np
import pytest

from conftest import GOLDEN_DIR, load_golden, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', params=['rp20', 'a36', 'm36', 'g30', 'd38'])       # d38: rank-5 tensor (dynamic T)
def setup(request):
    from qgs_amd.functions.tendencies import tendencies_from_tensor
    g = load_golden(request.param)
    f, Df = tendencies_from_tensor(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    yield g, f, Df
    f.operands.release()


def _kw(g, case):
    kw = dict(case['kw'])
    return kw


def test_f_and_Df_callables(setup):
    g, f, Df = setup
    x = g['fx_x'][3]
    assert f(0., x).shape == (g.ndim,) and rel_err(f(0., x), g['fx_f'][3]) < 1e-14
    assert Df(0., x).shape == (g.ndim, g.ndim) and rel_err(Df(0., x), g['fx_Df'][3]) < 1e-14
    assert f.coo.shape[1] == (5 if g.name == 'd38' else 3) and len(f.val) == f.coo.shape[0] and f.ndim == g.ndim
    f2 = pickle.loads(pickle.dumps(f))                       # picklable like the reference's f
    assert rel_err(f2(0., x), g['fx_f'][3]) < 1e-14
    f2.operands.release()


def test_functional_rk_api(setup):
    from qgs_amd.integrators.integrate import integrate_runge_kutta
    g, f, _ = setup
    ic = g['rk_ic']
    cases = {
        'f_w3': dict(t0=0., t=1., dt=0.1, ic=ic, forward=True, write_steps=3),
        'b_w3': dict(t0=0., t=1., dt=0.1, ic=ic, forward=False, write_steps=3),
        'f_w0': dict(t0=0., t=1., dt=0.1, ic=ic, forward=True, write_steps=0),
        'f_w1_single': dict(t0=0., t=0.5, dt=0.1, ic=ic[0], forward=True, write_steps=1),
        'f_w0_single': dict(t0=0., t=0.5, dt=0.1, ic=ic[0], forward=True, write_steps=0),
        'b_w2_t0': dict(t0=1., t=2.05, dt=0.1, ic=ic[:3], forward=False, write_steps=2),
    }
    for tag, kw in cases.items():
        tt, tr = integrate_runge_kutta(f, **kw)
        assert np.shape(tt) == g['api_%s_time' % tag].shape, tag
        assert np.array_equal(np.asarray(tt), g['api_%s_time' % tag]), tag
        assert rel_err(tr, g['api_%s_traj' % tag]) < 1e-12, tag


def test_functional_tgls_api(setup):
    from qgs_amd.integrators.integrate import integrate_runge_kutta_tgls
    g, f, Df = setup
    ic = g['rk_ic']
    common = dict(t0=0., t=0.3, dt=0.1)
    cases = {
        'tg_none': (None, dict(ic=ic[:2], write_steps=1)),
        'tg_1d': ('arr', dict(ic=ic[:2], write_steps=1)),
        'tg_2d_ntg': ('arr', dict(ic=ic[:2], write_steps=0)),
        'tg_2d_ntraj': ('arr', dict(ic=ic[:2], write_steps=2)),
        'tg_3d_swapped': ('arr', dict(ic=ic[:2], write_steps=1, adjoint=True)),
        'tg_3d': ('arr', dict(ic=ic[:2], write_steps=1, forward=False, inverse=True)),
        'tg_none_single': (None, dict(ic=ic[0], write_steps=0)),
    }
    for tag, (kind, kw) in cases.items():
        tg = None if kind is None else g['api_%s_tgic' % tag]
        tt, tr, fm = integrate_runge_kutta_tgls(f, Df, tg_ic=tg, **common, **kw)
        assert np.array_equal(np.asarray(tt), g['api_%s_time' % tag]), tag
        assert rel_err(tr, g['api_%s_traj' % tag]) < 1e-12, tag
        assert rel_err(fm, g['api_%s_fm' % tag]) < 1e-11, tag


def test_integrator_classes_vs_reference_classes(setup):
    from qgs_amd.integrators.integrator import RungeKuttaIntegrator, RungeKuttaTglsIntegrator
    g, f, Df = setup
    ic = g['rk_ic']
    integ = RungeKuttaIntegrator(num_threads=2)
    assert integ.integrate(0., 1., 0.1, ic=ic[:4]) == 0          # no function set: prints and returns 0
    integ.set_func(f)
    integ.integrate(0., 1., 0.1, ic=ic[:4], write_steps=5)
    tt, tr = integ.get_trajectories()
    assert np.array_equal(tt, g['cls_rk_w5_time']) and rel_err(tr, g['cls_rk_w5_traj']) < 1e-12
    assert (integ.n_traj, integ.n_dim, integ.n_records) == (4, g.ndim, 3)
    assert integ.get_ic() is integ.ic and integ.ic.shape == (4, g.ndim)
    integ.integrate(0., 1., 0.1, ic=ic[:4], write_steps=0, forward=False)
    tt, tr = integ.get_trajectories()
    assert np.ndim(tt) == 0 and tt == g['cls_rk_w0b_time'] and rel_err(tr, g['cls_rk_w0b_traj']) < 1e-12
    integ.terminate()

    tinteg = RungeKuttaTglsIntegrator(num_threads=2)
    tinteg.set_func(f, Df)
    tinteg.integrate(0., 0.3, 0.1, ic=ic[:2], write_steps=1)
    tt, tr, fm = tinteg.get_trajectories()
    assert np.array_equal(tt, g['cls_tgls_time'])
    assert rel_err(tr, g['cls_tgls_traj']) < 1e-12 and rel_err(fm, g['cls_tgls_fm']) < 1e-11
    assert tinteg.get_tg_ic().shape == (2, g.ndim, g.ndim)
    tinteg.terminate()


def test_initialize_draw_order_and_resume(setup):
    """`initialize` consumes np.random in num_threads-sized batches (integrator.py:257-291) and leaves
    (number_of_trajectories, n_dim) states in `ic`; feeding the last state back resumes a run."""
    from qgs_amd.integrators.integrator import RungeKuttaIntegrator
    g, f, _ = setup
    scale = 0.05
    integ = RungeKuttaIntegrator(num_threads=3)
    integ.set_func(f)
    np.random.seed(4)
    integ.initialize(2.0, 0.1, ic=np.random.rand(5, g.ndim) * scale)
    assert integ.get_ic().shape == (5, g.ndim) and np.isfinite(integ.get_ic()).all()
    # resume: 20 steps in one go == 10 + 10 steps fed back (qgs_rp.py:106-108 idiom)
    ic = g['rk_ic'][:3]
    integ.integrate(0., 2.0, 0.1, ic=ic, write_steps=0)
    _, whole = integ.get_trajectories()
    integ.integrate(0., 1.0, 0.1, ic=ic, write_steps=0)
    _, half = integ.get_trajectories()
    integ.integrate(0., 1.0, 0.1, ic=half, write_steps=0)
    _, two = integ.get_trajectories()
    assert rel_err(two, whole) < 1e-13


def test_tensor_tendencies_with_a_boundary_callable():
    """`boundary=` on tendencies from create_tendencies (integrate.py:600-603): the stage combination runs on the host, f and Df
    are still the HIP kernels (called with the whole ensemble); against the reference's output for the same call."""
    from callables_l84 import rp20_boundary
    from qgs_amd.functions.tendencies import tendencies_from_tensor
    from qgs_amd.integrators.integrate import integrate_runge_kutta_tgls
    from qgs_amd.integrators.integrator import RungeKuttaTglsIntegrator
    gold = np.load(os.path.join(GOLDEN_DIR, 'callables.npz'))
    g = load_golden('rp20')
    f, Df = tendencies_from_tensor(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    tt, tr, fm = integrate_runge_kutta_tgls(f, Df, 0., 1., 0.1, ic=gold['rp20_ic'], boundary=rp20_boundary, write_steps=2)
    assert np.array_equal(tt, gold['rp20_bnd_time'])
    assert rel_err(tr, gold['rp20_bnd_traj']) < 1e-12 and rel_err(fm, gold['rp20_bnd_fm']) < 1e-11
    integ = RungeKuttaTglsIntegrator(num_threads=1)
    integ.set_func(f, Df)
    integ.integrate(0., 1., 0.1, ic=gold['rp20_ic'], tg_ic=gold['rp20_tg1'], boundary=rp20_boundary, adjoint=True, write_steps=0)
    tt, tr, fm = integ.get_trajectories()
    assert tt == float(gold['rp20_bnd_adj_time'])
    assert rel_err(tr, gold['rp20_bnd_adj_traj']) < 1e-12 and rel_err(fm, gold['rp20_bnd_adj_fm']) < 1e-11
    # boundary=None stays on the device
    integ.integrate(0., 1., 0.1, ic=gold['rp20_ic'], write_steps=0)
    assert integ._model.last_kernel_info()['name'] != ''


def test_trajectories_statistics(setup):
    """qgs/integrators/statistics.py: batching must not change the ensemble mean."""
    from qgs_amd.integrators.integrator import RungeKuttaIntegrator
    from qgs_amd.integrators.statistics import TrajectoriesStatistics
    g, f, _ = setup
    integ = RungeKuttaIntegrator(num_threads=2)
    integ.set_func(f)
    ic = np.concatenate([g['rk_ic'], g['rk_ic'] * 1.01, g['rk_ic'] * 0.99])        # 24 members
    stats = TrajectoriesStatistics()
    stats.set_integrator(integ)
    stats.set_func_list([lambda traj: traj, lambda traj: traj ** 2])
    stats.compute_stats(0., 1., 0.1, ic=ic, write_steps=5, num=1)
    one = stats.get_stats()
    stats.compute_stats(0., 1., 0.1, ic=ic, write_steps=5, num=3)
    three = stats.get_stats()
    assert one.shape == (2, g.ndim, 3) and rel_err(three, one) < 1e-13
    integ.integrate(0., 1., 0.1, ic=ic, write_steps=5)
    _, traj = integ.get_trajectories()
    assert rel_err(one[0], traj.mean(axis=0)) < 1e-15
    integ.terminate()


def test_user_defined_quadratic_system_lorenz63():
    """The integrators accept any system of the form dx_i = sum T_ijk x_j x_k (x_0 = 1), not only qgs models:
    Lorenz-63 written as a COO tensor (the reference's docstrings integrate such user systems with jitted
    Python functions; here the tensor form runs on the same kernels)."""
    from qgs_amd.functions.tendencies import tendencies_from_tensor
    from qgs_amd.integrators.integrator import RungeKuttaIntegrator
    from oracle.oracle import OracleModel
    sigma, rho, beta = 10., 28., 8. / 3.
    terms = [(1, 0, 1, -sigma), (1, 0, 2, sigma),                 # dx = sigma (y - x)
             (2, 0, 1, rho), (2, 0, 2, -1.), (2, 1, 3, -1.),       # dy = x (rho - z) - y
             (3, 1, 2, 1.), (3, 0, 3, -beta)]                       # dz = x y - beta z
    coo = np.array([t[:3] for t in terms], dtype=np.int32)
    val = np.array([t[3] for t in terms])
    jterms = []
    for i, j, k, v in terms:                                        # Jacobian tensor: T + T.swapaxes(1, 2)
        jterms += [(i, j, k, v), (i, k, j, v)]
    jcoo = np.array([t[:3] for t in jterms], dtype=np.int32)
    jval = np.array([t[3] for t in jterms])
    f, Df = tendencies_from_tensor(3, coo, val, jcoo, jval)
    x = np.array([1., 2., 3.])
    assert np.allclose(f(0., x), [sigma * (2. - 1.), 1. * (rho - 3.) - 2., 1. * 2. - beta * 3.], rtol=1e-15)
    assert np.allclose(Df(0., x), [[-sigma, sigma, 0.], [rho - 3., -1., -1.], [2., 1., -beta]], rtol=1e-15)
    ic = np.random.RandomState(0).rand(300, 3)
    integ = RungeKuttaIntegrator()
    integ.set_func(f)
    integ.integrate(0., 1., 0.01, ic=ic, write_steps=10)
    tt, traj = integ.get_trajectories()
    ref = OracleModel(3, coo, val).integrate_runge_kutta_jit(np.concatenate((np.arange(0., 1., 0.01), [1.])), ic, 1, 10,
                                                              integ.b, integ.c, integ.a, threads=4)
    assert traj.shape == ref.shape and rel_err(traj, ref) < 1e-11        # chaotic system, 100 steps
    integ.terminate()
    f.operands.release()


@pytest.mark.parametrize('n_traj,write_steps,forward', [(1, 1, True), (63, 3, True), (1000, 2, False), (5000, 0, True)])
def test_ensemble_moments_match_numpy(n_traj, write_steps, forward):
    """integrate_moments (device reduction) == np.mean / np.var over the member axis of the full trajectories."""
    import model_configs
    from qgs_amd.functions.tendencies import create_tendencies
    from qgs_amd.integrators.integrator import RungeKuttaIntegrator
    from qgs_amd.integrators.statistics import TrajectoriesStatistics
    f, Df = create_tendencies(model_configs.params_m36())
    integ = RungeKuttaIntegrator()
    integ.set_func(f)
    ic = np.random.RandomState(n_traj).rand(n_traj, f.ndim) * 0.01 + 0.02
    integ.integrate(0., 2., 0.1, ic=ic, forward=forward, write_steps=write_steps)
    tt, traj = integ.get_trajectories()
    traj = np.reshape(traj, (n_traj, f.ndim, -1))
    t2, mean, var = integ.integrate_moments(0., 2., 0.1, ic=ic, forward=forward, write_steps=write_steps)
    assert np.array_equal(np.atleast_1d(t2), np.atleast_1d(tt))
    assert mean.shape == traj.shape[1:] and rel_err(mean, traj.mean(axis=0)) < 1e-13
    assert np.abs(var - traj.var(axis=0)).max() <= 1e-10 * traj.var(axis=0).max() + 1e-26
    assert rel_err(integ.last_final_states, traj[:, :, 0 if not forward else -1]) < 1e-15
    if n_traj >= 63:
        stats = TrajectoriesStatistics()
        stats.set_integrator(integ)
        t3, m3, v3 = stats.compute_moments(0., 2., 0.1, ic=ic, forward=forward, write_steps=write_steps, num=3)
        assert rel_err(m3, traj.mean(axis=0)) < 1e-13
        assert np.abs(v3 - traj.var(axis=0)).max() <= 1e-10 * traj.var(axis=0).max() + 1e-26
    integ.terminate()
    f.operands.release()


def test_sparse_mul_entry_points_match_oracle():
    """qgs.functions.sparse_mul.sparse_mul3 / sparse_mul2 as called from hand-written tendencies
    (user_guide.rst:437-458): same signatures, evaluated on the device."""
    from qgs_amd.functions.sparse_mul import sparse_mul2, sparse_mul3
    from oracle import oracle
    g = load_golden('a36')
    x = g['fx_x'][5]
    xx = np.concatenate(([1.], x))
    r3 = sparse_mul3(g['coo'], g['val'], xx, xx)
    assert r3.shape == (g.ndim + 1,) and r3[0] == 1. and rel_err(r3[1:], g['fx_f'][5]) < 1e-14
    assert rel_err(r3, oracle.sparse_mul3(g['coo'], g['val'], xx, xx)) < 1e-14
    r2 = sparse_mul2(g['jcoo'], g['jval'], xx)
    ref2 = oracle.sparse_mul2(g['jcoo'], g['jval'], xx)
    assert r2.shape == ref2.shape and np.abs(r2 - ref2).max() < 1e-14 * np.abs(ref2).max()
    assert np.abs(ref2[1:, 0]).max() > 0                      # column 0 (linear part) is exercised
    # any arguments the reference's functions take (sparse_mul.py:48-81: `a`, `b` are just two arrays): different vectors, a
    # constant slot that is not 1, entries in row 0 -- the general contraction kernel, bitwise the reference's loops
    rng = np.random.RandomState(3)
    a, b = rng.randn(g.ndim + 1), rng.randn(g.ndim + 1)
    coo0 = np.concatenate((g['coo'], np.array([[0, 2, 3], [0, 0, 0], [0, 5, 0]], dtype=g['coo'].dtype)))
    val0 = np.concatenate((g['val'], [0.25, -1.5, 2.0]))
    for coo_, val_ in ((g['coo'], g['val']), (coo0, val0)):
        got = sparse_mul3(coo_, val_, a, b)
        ref = oracle.sparse_mul3(coo_, val_, a, b)
        assert got[0] == 1. and np.array_equal(got, ref)
        assert np.array_equal(sparse_mul3(coo_, val_, 2. * xx, xx), oracle.sparse_mul3(coo_, val_, 2. * xx, xx))
    jcoo0 = np.concatenate((g['jcoo'], np.array([[0, 2, 3], [0, 0, 1], [4, 0, 0]], dtype=g['jcoo'].dtype)))
    jval0 = np.concatenate((g['jval'], [0.5, -0.75, 3.0]))
    for coo_, val_, v in ((g['jcoo'], g['jval'], a), (jcoo0, jval0, a), (jcoo0, jval0, xx)):
        assert np.array_equal(sparse_mul2(coo_, val_, v), oracle.sparse_mul2(coo_, val_, v))


def test_sparse_mul_rank5_entry_points_match_oracle():
    """sparse_mul5 / sparse_mul4 (sparse_mul.py:84-158) on the dynamic-T tensor, evaluated on the device."""
    from qgs_amd.functions.sparse_mul import sparse_mul4, sparse_mul5
    from oracle import oracle
    g = load_golden('d38')
    x = g['fx_x'][2]
    xx = np.concatenate(([1.], x))
    r5 = sparse_mul5(g['coo'], g['val'], xx, xx, xx, xx)
    assert r5.shape == (g.ndim + 1,) and r5[0] == 1. and rel_err(r5[1:], g['fx_f'][2]) < 1e-14
    r4 = sparse_mul4(g['jcoo'], g['jval'], xx, xx, xx)
    ref4 = oracle.sparse_mul4(g['jcoo'], g['jval'], xx, xx, xx)
    assert r4.shape == ref4.shape and np.abs(r4 - ref4).max() < 1e-14 * np.abs(ref4).max()
    assert np.abs(ref4[1:, 0]).max() > 0                      # column 0 (the j == 0 entries) is exercised
    # four (three) different vectors, entries in row 0 (sparse_mul.py:84-158 take any arrays): the general contraction kernel
    rng = np.random.RandomState(4)
    va, vb, vc, vd = (rng.randn(g.ndim + 1) for _ in range(4))
    coo0 = np.concatenate((g['coo'], np.array([[0, 2, 3, 0, 1], [0, 0, 0, 0, 0]], dtype=g['coo'].dtype)))
    val0 = np.concatenate((g['val'], [0.25, -1.5]))
    for coo_, val_ in ((g['coo'], g['val']), (coo0, val0)):
        got = sparse_mul5(coo_, val_, va, vb, vc, vd)
        assert got[0] == 1. and np.array_equal(got, oracle.sparse_mul5(coo_, val_, va, vb, vc, vd))
    jcoo0 = np.concatenate((g['jcoo'], np.array([[0, 2, 3, 0, 1], [3, 0, 0, 0, 0]], dtype=g['jcoo'].dtype)))
    jval0 = np.concatenate((g['jval'], [0.5, -0.75]))
    for coo_, val_ in ((g['jcoo'], g['jval']), (jcoo0, jval0)):
        assert np.array_equal(sparse_mul4(coo_, val_, va, vb, vc), oracle.sparse_mul4(coo_, val_, va, vb, vc))


def test_dynamic_T_model_end_to_end():
    """QgParams(dynamic_T=True) -> quadrature inner products -> QgsTensorDynamicT -> rank-5 f / Df on the GPU ->
    RungeKuttaIntegrator / RungeKuttaTglsIntegrator, against the reference's outputs for the same model
    (notebooks/maooam_dynamic_temperature.ipynb parameters).  The reference's tensor carries its quadrature error
    (2e-14 relative), hence 1e-10 instead of the 1e-12 of the models with closed-form inner products."""
    from model_configs import params_d38
    from qgs_amd.functions.tendencies import create_tendencies
    from qgs_amd.integrators.integrator import RungeKuttaIntegrator, RungeKuttaTglsIntegrator
    g = load_golden('d38')
    p = params_d38()
    f, Df = create_tendencies(p)
    assert p.ndim == 38 and f.coo.shape[1] == 5
    assert f.hip_model().specialised_available and f.hip_model().n_derived == (4, 22)
    assert rel_err(f(0., g['fx_x']), g['fx_f']) < 1e-10
    n = g['fx_Df'].shape[0]
    assert rel_err(Df(0., g['fx_x'][:n]), g['fx_Df']) < 1e-10
    ic = g['rk_ic']
    integ = RungeKuttaIntegrator()
    integ.set_func(f)
    integ.integrate(0., 1., 0.1, ic=ic[:4], write_steps=5)
    tt, tr = integ.get_trajectories()
    assert np.array_equal(tt, g['cls_rk_w5_time']) and rel_err(tr, g['cls_rk_w5_traj']) < 1e-10
    integ.terminate()
    tinteg = RungeKuttaTglsIntegrator()
    tinteg.set_func(f, Df)
    tinteg.integrate(0., 0.3, 0.1, ic=ic[:2], write_steps=1)
    tt, tr, fm = tinteg.get_trajectories()
    assert rel_err(tr, g['cls_tgls_traj']) < 1e-10 and rel_err(fm, g['cls_tgls_fm']) < 1e-10
    tinteg.terminate()
    f.operands.release()


def test_initialize_vs_the_reference_classes():
    """`initialize` against the reference's own (tests/golden/init_rp20.npz, make_golden.py gen_initialize): the states left in
    `ic` and where np.random stands afterwards.  `num_threads >= number_of_trajectories` is the fast way to spin up a large
    ensemble here -- ONE launch for all members -- and is exactly the reference's behaviour for that setting; fewer workers than
    members reproduce the reference's batch-by-batch growth (as many dependent launches as it has batches)."""
    from qgs_amd.functions.tendencies import tendencies_from_tensor
    from qgs_amd.integrators.integrator import RungeKuttaIntegrator
    g = load_golden('rp20')
    z = np.load(os.path.join(GOLDEN_DIR, 'init_rp20.npz'))
    f, _ = tendencies_from_tensor(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    integ = RungeKuttaIntegrator(num_threads=6)
    integ.set_func(f)
    np.random.seed(2023)
    integ.initialize(3.0, 0.1, number_of_trajectories=6)
    assert rel_err(integ.get_ic(), z['one_batch_ic']) < 1e-11
    assert np.array_equal(np.random.rand(3), z['one_batch_next_draw'])
    integ = RungeKuttaIntegrator(num_threads=2)
    integ.set_func(f)
    np.random.seed(2024)
    integ.initialize(3.0, 0.1, pert_size=0.01, reconvergence_time=1.0, number_of_trajectories=5)
    assert rel_err(integ.get_ic(), z['batched_ic']) < 1e-11
    assert np.array_equal(np.random.rand(3), z['batched_next_draw'])
    np.random.seed(2025)
    integ.initialize(2.0, 0.1, number_of_trajectories=2, forward=False)
    assert rel_err(integ.get_ic(), z['backward_ic']) < 1e-11
    # the one-batch spin-up of a large ensemble is one stepper launch per call
    big = RungeKuttaIntegrator(num_threads=20000)
    big.set_func(f)
    np.random.seed(1)
    big.initialize(1.0, 0.1, number_of_trajectories=20000)
    assert big.get_ic().shape == (20000, g.ndim) and np.isfinite(big.get_ic()).all()
    f.operands.release()


def test_host_transfers_go_through_the_library():
    """`_lib.to_device` / `_lib.to_host` (qgs_memcpy_h2d / qgs_memcpy_d2h): pageable NumPy memory reaches the GPU through the
    library's page-locked bounce blocks (qgs_amd/csrc/host_bridge.cpp), never as an operand of the runtime's own copies.  Round
    trips of sizes around the block boundaries (8 MiB blocks), other dtypes, non-contiguous sources, an empty array."""
    import torch
    from qgs_amd import _lib
    dev = torch.device('cuda', 0)
    rng = np.random.RandomState(12)
    for n in (0, 1, 7, (8 << 20) // 8 - 1, (8 << 20) // 8, (8 << 20) // 8 + 1, (24 << 20) // 8 + 5, 5000017):
        a = rng.rand(n)
        t = _lib.to_device(a, dev)
        assert t.device.type == 'cuda' and tuple(t.shape) == a.shape and t.dtype == torch.float64
        assert np.array_equal(t.cpu().numpy(), a)
        assert np.array_equal(_lib.to_host(t * 1.0), a)
    m = rng.rand(300, 17, 5)
    assert np.array_equal(_lib.to_host(_lib.to_device(m.transpose(2, 0, 1), dev)), m.transpose(2, 0, 1))     # non-contiguous source
    idx = np.arange(100000, dtype=np.int32)[::-1]
    assert np.array_equal(_lib.to_host(_lib.to_device(idx, dev)), idx)                                        # another dtype
    big = _lib.to_host(torch.arange(0, (520 << 20) // 8, dtype=torch.float64, device=dev))                    # >= 512 MB: result pool block
    assert big[0] == 0. and big[-1] == (520 << 20) // 8 - 1 and np.array_equal(big[::1000003], np.arange(0, len(big), 1000003, dtype=float))
    # the raw entry points refuse null pointers
    assert _lib.lib().qgs_memcpy_d2h(0, None, None, 8, None) != 0 and _lib.last_error()
