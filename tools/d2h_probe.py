"""Device-to-host cost of the host-layout API for a full record (1.9 GB): raw copies vs `HipModel.rk_integrate`."""
import time, numpy as np, torch, sys
sys.path.insert(0, '/root/repo')
from qgs_amd import _lib
n = 1900 * 1024 * 1024 // 8
d = torch.rand(n, dtype=torch.float64, device='cuda')
h_page = torch.empty(n, dtype=torch.float64)
h_pin = torch.empty(n, dtype=torch.float64, pin_memory=True)
for name, h in (('pageable', h_page), ('pinned', h_pin)):
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); h.copy_(d); torch.cuda.synchronize(); el = time.perf_counter() - t0
    print('D2H %s: %.1f ms  %.1f GB/s' % (name, el * 1e3, n * 8 / el / 1e9))
t0 = time.perf_counter(); a = np.empty(n); a[:] = h_pin.numpy(); print('pinned->numpy memcpy %.1f ms' % ((time.perf_counter() - t0) * 1e3))
g = np.load('/root/repo/tests/golden/m36.npz'); ndim = 36
m = _lib.HipModel(ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
RK4 = dict(c=np.array([0., 0.5, 0.5, 1.]), b=np.array([1. / 6, 1. / 3, 1. / 3, 1. / 6]), a=np.array([[0., 0, 0, 0], [0.5, 0, 0, 0], [0, 0.5, 0, 0], [0, 0, 1., 0]]))
ic = np.random.RandomState(1).rand(65536, ndim) * 0.01
t = np.concatenate((np.arange(0., 10. - 1e-9, 0.1), [10.]))
for ws in (0, 1):
    for it in range(4):
        tr = None
        t0 = time.perf_counter(); tr = m.rk_integrate(t, ic, 1, ws, RK4['b'], RK4['c'], RK4['a']); el = time.perf_counter() - t0
        print('   call %d: %.1f ms' % (it, el * 1e3))
    print('host API 65536 x 100 steps ws=%d: %.1f ms, result %.2f GB' % (ws, el * 1e3, tr.nbytes / 1e9))
