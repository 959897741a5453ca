"""CPU: pin the oracle (oracle/qgs_oracle.c) against the goldens captured from the Python reference.

f and Df must be BITWISE equal (same loop order, -ffp-contract=off).  Stepper outputs go through BLAS
`@` products in the reference (summation order unspecified) so they are compared to 1e-13 relative
(1e-12 for the 1000-step chaotic runs).
"""
import numpy as np
import pytest

from conftest import rel_err
from oracle.oracle import OracleModel, sparse_mul2, sparse_mul3, sparse_mul4, sparse_mul5


def _model(g):
    return OracleModel(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])


def test_f_bitwise(golden):
    m = _model(golden)
    assert np.array_equal(m.f(0., golden['fx_x']), golden['fx_f'])


def test_Df_bitwise(golden):
    m = _model(golden)
    n = golden['fx_Df'].shape[0]
    assert np.array_equal(m.Df(0., golden['fx_x'][:n]), golden['fx_Df'])


def test_sparse_mul_direct(golden_small):
    g = golden_small
    x = np.concatenate(([1.], g['fx_x'][0]))
    r3 = sparse_mul3(g['coo'], g['val'], x, x)
    assert r3[0] == 1.0                                          # sparse_mul.py:80
    assert np.array_equal(r3[1:], g['fx_f'][0])
    r2 = sparse_mul2(g['jcoo'], g['jval'], x)
    assert np.array_equal(r2[1:, 1:], g['fx_Df'][0])


def test_sparse_mul_rank5_direct():
    """sparse_mul5 / sparse_mul4 (sparse_mul.py:84-158) on the dynamic-T tensor, as the rank-5 closures call them
    (tendencies.py:98-109)."""
    from conftest import load_golden
    g = load_golden('d38')
    assert g['coo'].shape[1] == 5 and g['jcoo'].shape[1] == 5
    for n in range(3):
        x = np.concatenate(([1.], g['fx_x'][n]))
        r5 = sparse_mul5(g['coo'], g['val'], x, x, x, x)
        assert r5[0] == 1.0                                      # sparse_mul.py:157
        assert np.array_equal(r5[1:], g['fx_f'][n])
        r4 = sparse_mul4(g['jcoo'], g['jval'], x, x, x)
        assert np.array_equal(r4[1:, 1:], g['fx_Df'][n])


def test_known_answers_survey():
    """SURVEY.md section 8(c) smoke known-answers (config A, x = RandomState(0).rand(36)*0.01)."""
    from conftest import load_golden
    g = load_golden('a36')
    m = _model(g)
    x = np.random.RandomState(0).rand(36) * 0.01
    f = m.f(0., x)
    assert np.allclose(f[[0, 1, 10, 20, 28, 35]],
                       [0.001396219628498062, 0.0003946983379165183, 0.0004241706609841259,
                        -4.4547687518609185e-07, 1.0930977948513325e-05, -8.892286561930322e-06], rtol=1e-14, atol=0)
    assert abs(f.sum() - (-0.0008270056083908564)) < 1e-17
    assert abs(np.abs(m.Df(0., x)).sum() - 5.63329111859456) < 1e-12


def test_rk_cases(golden):
    m = _model(golden)
    for cs in golden.meta['rk_cases']:
        t = cs['tag']
        ic = golden['rk_ic'][:cs['n_traj']]
        rec = m.integrate_runge_kutta_jit(golden['rk_%s_time' % t], ic, 1 if cs['forward'] else -1, cs['ws'],
                                          golden['rk_%s_b' % t], golden['rk_%s_c' % t], golden['rk_%s_a' % t])
        tol = 1e-12 if cs['steps'] >= 1000 else 1e-13
        assert rel_err(rec, golden['rk_%s_traj' % t]) < tol, t


def test_tgls_cases(golden):
    m = _model(golden)
    for cs in golden.meta['tgls_cases']:
        t = cs['tag']
        rec, fm = m.integrate_runge_kutta_tgls_jit(golden['tgls_%s_time' % t], golden['tgls_ic'],
                                                   golden['tgls_%s_tgic' % t], 1 if cs['forward'] else -1, cs['ws'],
                                                   golden['tgls_%s_b' % t], golden['tgls_%s_c' % t],
                                                   golden['tgls_%s_a' % t], cs['adjoint'], -1. if cs['inverse'] else 1.)
        assert rel_err(rec, golden['tgls_%s_traj' % t]) < 1e-13, t
        assert rel_err(fm, golden['tgls_%s_fm' % t]) < 1e-13, t


def test_threads_do_not_change_results(golden_small):
    g = golden_small
    m = _model(g)
    t = g['rk_s10_w1_f_rk4_time']
    a = m.integrate_runge_kutta_jit(t, g['rk_ic'], 1, 1, g['rk_s10_w1_f_rk4_b'], g['rk_s10_w1_f_rk4_c'],
                                    g['rk_s10_w1_f_rk4_a'], threads=1)
    b = m.integrate_runge_kutta_jit(t, g['rk_ic'], 1, 1, g['rk_s10_w1_f_rk4_b'], g['rk_s10_w1_f_rk4_c'],
                                    g['rk_s10_w1_f_rk4_a'], threads=4)
    assert np.array_equal(a, b)


@pytest.mark.parametrize('n_time,ws', [(1, 0), (1, 1), (2, 1), (11, 1), (11, 3), (11, 5), (11, 10), (11, 11), (12, 4)])
def test_n_records_matches_reference_formula(n_time, ws):
    """integrate.py:190-196 evaluated literally with NumPy."""
    time = np.concatenate((np.arange(0., (n_time - 1) * 0.1 - 1e-12, 0.1), [(n_time - 1) * 0.1]))[:n_time]
    time = np.linspace(0., 1., n_time) if len(time) != n_time else time
    if ws == 0:
        expect = 1
    else:
        tot = time[::ws]
        expect = len(tot) + (1 if tot[-1] != time[-1] else 0)
    assert OracleModel.n_records(time, ws) == expect
