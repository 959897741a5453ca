"""CPU: user-written Python callables and `boundary` terms through the integrators (host stepper,
qgs_amd/integrators/host_stepper.py) against outputs of the reference for the same calls (tests/golden/callables.npz, written
by tests/golden/make_golden.py `callables`).  Reference: qgs/integrators/integrator.py:1237-1291 (usage example),
integrate.py:182-223, 555-614."""
import os

import numpy as np
import pytest

from callables_l84 import DfL84, fL84, tboundary
from conftest import GOLDEN_DIR, rel_err

TOL = 1e-12      # same arithmetic up to the summation order of the stage combinations (the reference's `@` goes through BLAS)


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(GOLDEN_DIR, 'callables.npz'))


KUTTA3 = dict(b=np.array([1. / 6, 2. / 3, 1. / 6]), c=np.array([0., .5, 1.]), a=np.array([[0., 0, 0], [.5, 0, 0], [-1., 2., 0]]))


@pytest.mark.parametrize('tag,kw', [('fw_w10', dict(write_steps=10)), ('bw_w7', dict(forward=False, write_steps=7)),
                                    ('w0', dict(write_steps=0)), ('kutta3_w5', dict(write_steps=5, **KUTTA3))])
def test_functional_api_with_a_python_callable(gold, tag, kw):
    from qgs_amd.integrators.integrate import integrate_runge_kutta
    tt, tr = integrate_runge_kutta(fL84, 0., 2., 0.01, ic=gold['ic'], **kw)
    assert np.array_equal(np.asarray(tt), gold['rk_%s_time' % tag])
    assert rel_err(tr, gold['rk_%s_traj' % tag]) < TOL


def test_single_trajectory_is_squeezed_like_the_reference(gold):
    from qgs_amd.integrators.integrate import integrate_runge_kutta
    tt, tr = integrate_runge_kutta(fL84, 0., 1., 0.01, ic=gold['ic'][0], write_steps=20)
    assert tr.shape == gold['rk_single_traj'].shape == (3, 6)
    assert rel_err(tr, gold['rk_single_traj']) < TOL


@pytest.mark.parametrize('tag,kw', [('bnd_zero_tg', dict(tg_ic=np.zeros(3), boundary=tboundary, write_steps=10)),
                                    ('bnd_adj_inv_bw', dict(tg_ic='tg2', boundary=tboundary, write_steps=10, adjoint=True,
                                                            inverse=True, forward=False)),
                                    ('nobnd_identity', dict(write_steps=25)),
                                    ('bnd_identity_w0', dict(boundary=tboundary, write_steps=0))])
def test_tangent_model_with_callables_and_boundary(gold, tag, kw):
    from qgs_amd.integrators.integrate import integrate_runge_kutta_tgls
    kw = dict(kw)
    if isinstance(kw.get('tg_ic'), str):
        kw['tg_ic'] = gold[kw['tg_ic']]
    tt, tr, fm = integrate_runge_kutta_tgls(fL84, DfL84, 0., 2., 0.01, ic=gold['ic'], **kw)
    assert np.array_equal(np.asarray(tt), gold['tg_%s_time' % tag])
    assert tr.shape == gold['tg_%s_traj' % tag].shape and fm.shape == gold['tg_%s_fm' % tag].shape
    assert rel_err(tr, gold['tg_%s_traj' % tag]) < TOL
    assert rel_err(fm, gold['tg_%s_fm' % tag]) < TOL


def test_usage_example_of_the_reference_classes(gold):
    """integrator.py:1253-1291: set_func(fL84) with no initial condition (the dimension is found by probing), results fed
    back as initial conditions, backward run from the last record, then the tangent model with a boundary term."""
    from qgs_amd.integrators.integrator import RungeKuttaIntegrator, RungeKuttaTglsIntegrator
    integ = RungeKuttaIntegrator(num_threads=2)
    integ.set_func(fL84)
    integ.integrate(0., 5., 0.01, write_steps=0)
    tt, tr0 = integ.get_trajectories()
    assert tt == float(gold['cls_spinup_time']) and tr0.shape == (3,)
    assert rel_err(tr0, gold['cls_spinup_traj']) < TOL
    assert integ.n_dim == 3 and integ.n_traj == 1
    integ.integrate(0., 2., 0.01, ic=tr0, write_steps=10)
    tt, tr1 = integ.get_trajectories()
    # chained runs on the Lorenz-84 attractor: rounding differences of the stage sums grow along the trajectory
    assert np.array_equal(tt, gold['cls_fw_time']) and rel_err(tr1, gold['cls_fw_traj']) < 1e-9
    integ.integrate(0., 2., 0.01, ic=tr1[:, -1], write_steps=10, forward=False)
    tt, tr2 = integ.get_trajectories()
    assert np.array_equal(tt, gold['cls_bw_time']) and rel_err(tr2, gold['cls_bw_traj']) < 1e-8
    integ.terminate()

    tgls = RungeKuttaTglsIntegrator(num_threads=2)
    tgls.set_func(fL84, DfL84)
    tgls.initialize(1., 0.01, ic=gold['ic'])
    assert rel_err(tgls.get_ic(), gold['cls_tg_ic']) < TOL
    tgls.integrate(0., 2., 0.01, write_steps=10, tg_ic=np.zeros(3), boundary=tboundary)
    t, x, fm = tgls.get_trajectories()
    assert np.array_equal(t, gold['cls_tg_time'])
    assert x.shape == gold['cls_tg_traj'].shape and fm.shape == gold['cls_tg_fm'].shape
    assert rel_err(x, gold['cls_tg_traj']) < 1e-9 and rel_err(fm, gold['cls_tg_fm']) < 1e-9
    tgls.terminate()


def test_missing_function_prints_and_returns_zero(capsys):
    from qgs_amd.integrators.integrator import RungeKuttaIntegrator
    integ = RungeKuttaIntegrator(num_threads=1)
    assert integ.integrate(0., 1., 0.1) == 0
    assert 'No function to integrate defined!' in capsys.readouterr().out
