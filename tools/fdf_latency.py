"""GPU box: latency of the single-state callables f(t, x) / Df(t, x) (what SciPy / DiffEq solvers call, user_guide.rst:502-517).

Reports, per model, microseconds per call: through the C-ABI alone (ctypes call on preallocated arrays) and through the Python
callables of create_tendencies (array conversion + result allocation included); `QGS_HIP_FDF_BATCH_PATH=1` in the environment of a
second run is not needed: the old route (n_traj == 1 through pack / kernel / unpack with two blocking copies) is what
`kernel_kind == 1` still takes, so it is timed here as 'batched route' for comparison.
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from qgs_amd import _lib                                                      # noqa: E402
from qgs_amd.functions.tendencies import tendencies_from_tensor              # noqa: E402


def per_call_us(fn, n=2000, warm=50):
    for _ in range(warm):
        fn()
    best = 1e30
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        best = min(best, (time.perf_counter() - t0) / n)
    return best * 1e6


for name in ('m36', 't228', 'd38'):
    g = np.load(os.path.join(REPO, 'tests', 'golden', name + '.npz'))
    ndim = int(g['ndim'])
    f, Df = tendencies_from_tensor(ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    m = f.hip_model()
    L = _lib.lib()
    x = np.random.RandomState(0).rand(ndim) * 0.01
    dx, J = np.empty(ndim), np.empty((ndim, ndim))
    res = {}
    res['f  C-ABI'] = per_call_us(lambda: L.qgs_tendencies(m._h, 1, x, dx))
    kf = m.last_kernel_info()['name']
    res['Df C-ABI'] = per_call_us(lambda: L.qgs_jacobian(m._h, 1, x, J), n=1000)
    kj = m.last_kernel_info()['name']
    res['f  Python callable'] = per_call_us(lambda: f(0., x))
    res['Df Python callable'] = per_call_us(lambda: Df(0., x), n=1000)
    m.set_kernel(1)                                                           # generic family: n_traj == 1 takes the batched route
    res['f  batched route (round 2)'] = per_call_us(lambda: L.qgs_tendencies(m._h, 1, x, dx), n=500)
    res['Df batched route (round 2)'] = per_call_us(lambda: L.qgs_jacobian(m._h, 1, x, J), n=300)
    m.set_kernel(0)
    print('%s (ndim %d): kernels %s / %s' % (name, ndim, kf, kj))
    for k, v in res.items():
        print('    %-28s %8.2f us per call' % (k, v))
    f.operands.release()
