"""GPU: the reference's own integrator test (model_test/test_tlad.py) through the drop-in API:
QgParams -> create_tendencies -> RungeKuttaIntegrator / RungeKuttaTglsIntegrator.

Same model (RP 20-variable, hd=0.3, h_2=0.4, theta*_1=0.2), same Taylor test (ratio -> 1 within
delta/10) and the same adjoint identity <TL x, y> = <x, AD y> within 1e-3 (the adjoint run is the RK4
integration of J^T, not the exact discrete adjoint, hence the reference's loose bound).  The spin-up is
2*10^5 steps instead of 2*10^6 to keep the test short.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REAL_EPS = 1.e-3


@pytest.fixture(scope='module')
def tlad():
    from qgs_amd.params.params import QgParams
    from qgs_amd.integrators.integrator import RungeKuttaIntegrator, RungeKuttaTglsIntegrator
    from qgs_amd.functions.tendencies import create_tendencies

    p = QgParams({'phi0_npi': np.deg2rad(50.) / np.pi, 'hd': 0.3})
    p.set_atmospheric_channel_fourier_modes(2, 2)
    p.ground_params.set_orography(0.4, 1)
    p.atemperature_params.set_thetas(0.2, 0)
    f, Df = create_tendencies(p)
    integrator = RungeKuttaIntegrator()
    integrator.set_func(f)
    np.random.seed(20250328)
    ic = np.random.rand(p.ndim) * 0.01
    integrator.integrate(0., 20000., 0.1, ic=ic, write_steps=0)
    _, ic = integrator.get_trajectories()
    assert ic.shape == (p.ndim,) and np.isfinite(ic).all()
    tgls = RungeKuttaTglsIntegrator()
    tgls.set_func(f, Df)
    yield p, integrator, tgls, ic
    integrator.terminate()
    tgls.terminate()


def test_taylor(tlad):
    p, integrator, tgls, y0 = tlad
    for n in range(0, 7):
        dy = np.full_like(y0, 2. ** (-n) / np.sqrt(float(p.ndim)))
        integrator.integrate(0., 0.1, 0.1, ic=y0, write_steps=0)
        _, y1 = integrator.get_trajectories()
        integrator.integrate(0., 0.1, 0.1, ic=y0 + dy, write_steps=0)
        _, y1prime = integrator.get_trajectories()
        dy1 = y1prime - y1
        tgls.integrate(0., 0.1, dt=0.1, write_steps=0, ic=y0, tg_ic=dy)
        _, _, dy1_tl = tgls.get_trajectories()
        ratio = np.dot(dy1, dy1) / np.dot(dy1_tl, dy1_tl)
        assert abs(ratio - 1.) < dy[0] / 10, (n, ratio)


def test_adjoint_identity(tlad):
    p, _, tgls, y0 = tlad
    rng = np.random.RandomState(3)
    for _ in range(20):
        dy, dy_bis = rng.randn(p.ndim), rng.randn(p.ndim)
        out = {}
        for key, vec, adj in (('tl', dy, False), ('ad', dy, True), ('bis_tl', dy_bis, False), ('bis_ad', dy_bis, True)):
            tgls.integrate(0., 0.1, dt=0.1, write_steps=0, ic=y0, tg_ic=vec, adjoint=adj)
            out[key] = tgls.get_trajectories()[2]
        assert abs(np.dot(out['tl'], dy_bis) - np.dot(dy, out['bis_ad'])) < REAL_EPS
        assert abs(np.dot(out['bis_tl'], dy) - np.dot(dy_bis, out['ad'])) < REAL_EPS


def test_scripts_flow_maooam():
    """qgs_maooam.py:78-131 in miniature: parameters -> tendencies -> transient -> recorded run."""
    from model_configs import params_m36
    from qgs_amd.integrators.integrator import RungeKuttaIntegrator
    from qgs_amd.functions.tendencies import create_tendencies
    from conftest import load_golden, rel_err
    p = params_m36()
    f, Df = create_tendencies(p)
    g = load_golden('m36')
    assert rel_err(f(0., g['fx_x'][0]), g['fx_f'][0]) < 1e-14
    integrator = RungeKuttaIntegrator()
    integrator.set_func(f)
    np.random.seed(1)
    y = np.random.rand(p.ndim) * 0.01
    total = 0.
    for _ in range(3):
        integrator.integrate(0., 100 * 0.1, 0.1, ic=y, write_steps=0)
        t, y = integrator.get_trajectories()
        total += t
    assert abs(total - 30.) < 1e-9 and y.shape == (36,)
    integrator.integrate(0., 10., 0.1, ic=y, write_steps=10)
    t, traj = integrator.get_trajectories()
    assert t.shape == (11,) and traj.shape == (36, 11) and np.array_equal(traj[:, 0], y)
    integrator.terminate()
