"""GPU box: cost of SMALL host-layout calls -- the reference's scripts integrate one trajectory in a loop of short `integrate`
calls (qgs_rp.py:102-108) -- microseconds per call of HipModel.rk_integrate and of RungeKuttaIntegrator.integrate + get_trajectories."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from qgs_amd.functions.tendencies import tendencies_from_tensor
from qgs_amd.integrators.integrator import RungeKuttaIntegrator
g = np.load(os.path.join(REPO, 'tests', 'golden', 'm36.npz')); ndim = int(g['ndim'])
f, Df = tendencies_from_tensor(ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
m = f.hip_model()
c = np.array([0., .5, .5, 1.]); b = np.array([1 / 6, 1 / 3, 1 / 3, 1 / 6]); a = np.zeros((4, 4)); a[1, 0] = .5; a[2, 1] = .5; a[3, 2] = 1.
for n, steps, ws in ((1, 10, 1), (1, 100, 0), (64, 10, 1), (4096, 10, 0)):
    ic = np.random.RandomState(0).rand(n, ndim) * 0.01
    def call(k):
        t = np.concatenate((np.arange(0., steps * 0.1, 0.1), [steps * 0.1]))[:steps + 1] + 0.1 * (k % 7)    # a new grid every call, like a loop over t0
        return m.rk_integrate(t, ic, 1, ws, b, c, a)
    for k in range(20): call(k)
    t0 = time.perf_counter()
    for k in range(500): call(k)
    us = (time.perf_counter() - t0) / 500 * 1e6
    integ = RungeKuttaIntegrator(num_threads=1); integ.set_func(f)
    for k in range(20): integ.integrate(0.1 * k, 0.1 * k + steps * 0.1, 0.1, ic=ic, write_steps=ws); integ.get_trajectories()
    t0 = time.perf_counter()
    for k in range(500):
        integ.integrate(0.1 * k, 0.1 * k + steps * 0.1, 0.1, ic=ic, write_steps=ws); integ.get_trajectories()
    us2 = (time.perf_counter() - t0) / 500 * 1e6
    print('%5d members x %4d steps, write_steps %d: HipModel.rk_integrate %7.1f us per call; class integrate + get_trajectories %7.1f us  (%s)'
          % (n, steps, ws, us, us2, m.last_kernel_info()['name']))
