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
    for cs in meta['cases']:
        tag = cs['tag']
        np.random.seed(cs['seed'])
        est.set_noise_pert(cs['noise_pert'])
        est.compute_clvs(meta['t0'], meta['ta'], meta['tb'], meta['tc'], meta['dt'], meta['mdt'], ic=g['ic'], write_steps=cs['ws'],
                         method=cs['method'], backward_vectors=True, forward_vectors=True)
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
