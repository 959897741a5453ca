"""Developer script (GPU box): the two generic batched QR kernels (matrix in LDS: rows <= 300, cols <= 64; matrix in a global
scratch copy: anything larger) at the shapes the ndim-228 Lyapunov runs use; time per call and bytes moved per second."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qgs_amd import _lib
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'm36.npz'))
m = _lib.HipModel(int(g['ndim']), g['coo'], g['val'], g['jcoo'], g['jval'])
for n, R, C in ((4096, 228, 40), (4096, 228, 10), (16384, 100, 36), (1024, 228, 228), (4096, 228, 64)):
    a = torch.randn((R, C, n), dtype=torch.float64, device='cuda')
    rd = torch.zeros((C, n), dtype=torch.float64, device='cuda')
    w = a.clone()
    m.batched_qr_device(n, n, R, C, w.data_ptr(), rd.data_ptr()); torch.cuda.synchronize()
    q = w[:, :, :2].cpu().numpy().transpose(2, 0, 1); a2 = a[:, :, :2].cpu().numpy().transpose(2, 0, 1)
    err = max(np.abs(q[i] - np.linalg.qr(a2[i])[0]).max() for i in range(2))
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w.copy_(a); e0.record()
        m.batched_qr_device(n, n, R, C, w.data_ptr(), rd.data_ptr())
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts)); gb = 2 * 8 * R * C * n / 1e9
    print('%d x %dx%d: %.3f ms, %.0f GB/s of the %.2f GB in + out, %s, err %.1e' % (n, R, C, ms, gb / ms * 1e3, gb, m.last_kernel_info()['name'] or 'batched_qr_kernel', err), flush=True)
