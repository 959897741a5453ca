import os, sys, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from qgs_amd import _lib
g = np.load('/root/repo/tests/golden/t228.npz'); ndim = int(g['ndim'])
m = _lib.HipModel(ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
dev = torch.device('cuda', 0); st = torch.cuda.current_stream().cuda_stream
for n in (64, 4096, 65536):
    x = torch.rand((ndim, n), dtype=torch.float64, device=dev) * 0.01
    dx = torch.empty_like(x)
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.tendencies_device(n, n, x.data_ptr(), dx.data_ptr(), st)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
    print('f   n=%6d: %8.3f ms  %.2e evals/s  %.2f TFLOP/s' % (n, el * 1e3, n / el, n / el * 27770 * 3 / 1e12), m.last_kernel_info()['name'])
g = np.load('/root/repo/tests/golden/m36.npz'); ndim = int(g['ndim'])
m = _lib.HipModel(ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
for n in (65536, 1048576):
    x = torch.rand((ndim, n), dtype=torch.float64, device=dev) * 0.01
    dx = torch.empty_like(x)
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.tendencies_device(n, n, x.data_ptr(), dx.data_ptr(), st)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
    print('f36 n=%6d: %8.3f ms  %.2e evals/s  %.1f GB/s' % (n, el * 1e3, n / el, n * 36 * 16 / el / 1e9), m.last_kernel_info()['name'])
