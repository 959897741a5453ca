"""Dynamic-T MAOOAM 6x6 / 6x6 (ndim 230, rank 5, 71 825 tensor entries): host setup time, LDS-resident stepper and the 8 x 8-tile
tangent kernel against the generic kernels (parity + time)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qgs_amd.params.params import QgParams
from qgs_amd.functions.tendencies import create_tendencies
RK4 = dict(c=np.array([0., 0.5, 0.5, 1.]), b=np.array([1. / 6, 1. / 3, 1. / 3, 1. / 6]), a=np.array([[0., 0, 0, 0], [0.5, 0, 0, 0], [0, 0.5, 0, 0], [0, 0, 1., 0]]))
t0 = time.time()
p = QgParams({'n': 1.5}, dynamic_T=True)
p.set_atmospheric_channel_fourier_modes(6, 6, mode="symbolic")
p.set_oceanic_basin_fourier_modes(6, 6, mode="symbolic")
f, Df = create_tendencies(p)
print('host setup %.1f s: ndim %d, %d tensor entries, %d Jacobian-tensor entries' % (time.time() - t0, p.ndim, len(f.val), len(Df.val)), flush=True)
m = f.hip_model()
print('derived monomials', m.n_derived, flush=True)
rng = np.random.RandomState(0)
vr = p.variables_range
def ics(n):
    ic = rng.rand(n, p.ndim) * 0.01; ic[:, vr[0]] += 1.5; ic[:, vr[2]] += 3.; return ic
t = np.concatenate((np.arange(0., 0.5, 0.1), [0.5]))
ic = ics(64)
tg = rng.randn(64, p.ndim, 8)
res = {}
for kind in (2, 1):
    m.set_kernel(kind)
    for rep in range(2):
        t1 = time.time(); tr = m.rk_integrate(t, ic, 1, 0, RK4['b'], RK4['c'], RK4['a']); dt_rk = time.time() - t1
    name_rk = m.last_kernel_info()['name']
    for rep in range(2):
        t1 = time.time(); tr2, fm = m.rk_tgls_integrate(t, ic, tg, 1, 0, RK4['b'], RK4['c'], RK4['a'], False, 1.); dt_tg = time.time() - t1
    res[kind] = (tr, fm)
    print('kind %d: rk 64 x 5 steps %.1f ms (%s); tgls 64 x 8 x 5 steps %.1f ms (%s)' % (kind, dt_rk * 1e3, name_rk, dt_tg * 1e3, m.last_kernel_info()['name']), flush=True)
print('specialised vs generic: traj %.1e, propagator %.1e' % (np.abs(res[2][0] - res[1][0]).max() / np.abs(res[1][0]).max(),
                                                              np.abs(res[2][1] - res[1][1]).max() / np.abs(res[1][1]).max()))
m.set_kernel(2)
ic = ics(16384)
tt = np.concatenate((np.arange(0., 2.0, 0.1), [2.0]))
for rep in range(2):
    t1 = time.time(); m.rk_integrate(tt, ic, 1, 0, RK4['b'], RK4['c'], RK4['a']); el = time.time() - t1
print('stepper 16384 members x 20 steps (host API): %.1f ms = %.2e traj-steps/s' % (el * 1e3, 16384 * 20 / el))
