"""3/8-rule RK4 (dense lower-triangular tableau) at 65 536 members x 1000 steps, MAOOAM-36: the specialised general-tableau
stepper (qgs_spec_rkd_s4) against the generic kernel it replaces and the classic-RK4 stepper."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qgs_amd import _lib
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'm36.npz')); ndim = int(g['ndim'])
m = _lib.HipModel(ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
rk4 = (np.array([1 / 6, 1 / 3, 1 / 3, 1 / 6]), np.array([0., .5, .5, 1.]), np.array([[0., 0, 0, 0], [.5, 0, 0, 0], [0, .5, 0, 0], [0, 0, 1., 0]]))
r38 = (np.array([1., 3., 3., 1.]) / 8., np.array([0., 1. / 3, 2. / 3, 1.]), np.array([[0., 0, 0, 0], [1. / 3, 0, 0, 0], [-1. / 3, 1., 0, 0], [1., -1., 1., 0]]))
n, steps = 65536, 1000
t = np.concatenate((np.arange(0., steps * 0.1, 0.1), [steps * 0.1]))[:steps + 1]
ic = torch.from_numpy(np.random.RandomState(1).rand(ndim, n) * 0.01).cuda(); rec = torch.empty((1, ndim, n), dtype=torch.float64, device='cuda')
st = torch.cuda.current_stream().cuda_stream
for label, (b, c, a), kind, st_n in (('classic RK4', rk4, 2, steps), ('3/8 rule, specialised', r38, 2, steps), ('3/8 rule, generic', r38, 1, 20)):
    m.set_kernel(kind)
    tt = t[:st_n + 1]
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.rk_integrate_device(n, n, ic.data_ptr(), tt, 1, 0, b, c, a, rec.data_ptr(), st)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    el = min(ts[1:])
    print('%-24s %8.3f ms for %4d steps  %.3e traj-steps/s  (%s)' % (label, el * 1e3, st_n, n * st_n / el, m.last_kernel_info()['name']), flush=True)
