"""Tangent model (MAOOAM-36, full 36-column propagator, 10 RK4 steps) at several ensemble sizes: the shared-stage-state
kernel (qgs_spec_tglx4) against the one-wavefront-per-column kernel (QGS_HIP_TGL_VARIANT=plain)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qgs_amd import _lib  # noqa: E402

c = np.array([0., .5, .5, 1.]); b = np.array([1 / 6, 1 / 3, 1 / 3, 1 / 6]); a = np.zeros((4, 4)); a[1, 0] = .5; a[2, 1] = .5; a[3, 2] = 1.
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'm36.npz'))
ndim = int(g['ndim'])
m = _lib.HipModel(ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
m.set_kernel(2)
dev = torch.device('cuda', 0)
st = torch.cuda.current_stream().cuda_stream
steps, n_tg = 10, 36
t = np.concatenate((np.arange(0., steps * 0.01 - 1e-12, 0.01), [steps * 0.01]))
for n in (1024, 4096, 16384, 65536, 131072):
    ic = torch.from_numpy(np.random.RandomState(2).rand(ndim, n) * 0.01).to(dev)
    tg = torch.zeros((ndim, n_tg, n), dtype=torch.float64, device=dev)
    for d in range(ndim):
        tg[d, d, :] = 1.0
    rec = torch.empty((1, ndim, n), dtype=torch.float64, device=dev)
    recm = torch.empty((1, ndim, n_tg, n), dtype=torch.float64, device=dev)
    res = {}
    for variant in ('shared', 'plain'):
        if variant == 'plain':
            os.environ['QGS_HIP_TGL_VARIANT'] = 'plain'
        else:
            os.environ.pop('QGS_HIP_TGL_VARIANT', None)
        m.set_kernel(2)                        # the selection knobs are read at set_kernel / model creation
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m.rk_tgls_integrate_device(n, n, n_tg, ic.data_ptr(), tg.data_ptr(), t, 1, 0, b, c, a, False, 1., rec.data_ptr(), recm.data_ptr(), st)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res[variant] = (min(ts[1:]), m.last_kernel_info()['name'])
    print('%7d members x 36 columns x 10 steps: %-18s %8.3f ms | %-16s %8.3f ms | %.2fx' %
          (n, res['shared'][1], res['shared'][0] * 1e3, res['plain'][1], res['plain'][0] * 1e3, res['plain'][0] / res['shared'][0]), flush=True)
    del tg, recm
