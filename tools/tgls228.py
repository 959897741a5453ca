#!/usr/bin/env python3
"""Developer script: TGLS timings at MAOOAM 6x6 (ndim 228) for a few (members, columns) shapes."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import model_configs
from qgs_amd.functions.tendencies import create_tendencies
p = model_configs.params_t228(); f, Df = create_tendencies(p); m = f.hip_model(); ndim = 228
m.set_kernel(int(os.environ.get('TGLS228_KIND', '0')))        # 2: force the specialised kernels (JIT if not cached)
c = np.array([0., .5, .5, 1.]); b = np.array([1 / 6, 1 / 3, 1 / 3, 1 / 6]); a = np.zeros((4, 4)); a[1, 0] = .5; a[2, 1] = .5; a[3, 2] = 1.
t = np.concatenate((np.arange(0., 0.1, 0.01), [0.1]))
dev = torch.device('cuda', 0); st = torch.cuda.current_stream().cuda_stream
shapes = ((1, 228), (1, 20), (64, 8), (1024, 4), (4096, 8), (16384, 8), (1024, 228))
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
for n, nv in shapes:
    ld = (n + 63) // 64 * 64
    ic = torch.from_numpy(np.random.RandomState(2).rand(ndim, ld) * 0.01).to(dev)
    q = torch.randn((ndim, nv, ld), dtype=torch.float64, device=dev)
    qn = torch.empty((1, ndim, nv, ld), dtype=torch.float64, device=dev)
    yend = torch.empty((1, ndim, ld), dtype=torch.float64, device=dev)
    fn = lambda: m.rk_tgls_integrate_device(n, ld, nv, ic.data_ptr(), q.data_ptr(), t, 1, 0, b, c, a, False, 1., yend.data_ptr(), qn.data_ptr(), st)
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); el = time.perf_counter() - t0
    # 4 stages x 55 522 Jacobian-tensor terms x 3 flop per (member, column) pair and step
    print('n=%5d n_tg=%3d 10 steps: %9.3f ms  %s  %.3e pair-steps/s  %.2f TFLOP/s' % (n, nv, el * 1e3, m.last_kernel_info()['name'], n * nv * 10 / el,
                                                                                   n * nv * 10 / el * 4 * 55522 * 3 / 1e12), flush=True)
