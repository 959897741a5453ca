"""Developer script: batched QR of 16 384 36x36 matrices (the Benettin step of BASELINE config 4), HIP events, 20 calls back to back;
RK_AB_LIB=<other build of the library> compares generator versions on one box (tools/build_prev_lib.sh)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qgs_amd import _lib
if os.environ.get('RK_AB_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['RK_AB_LIB'])
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'm36.npz')); ndim = int(g['ndim'])
m = _lib.HipModel(ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
for n, R, C in ((16384, 36, 36), (16384, 36, 10), (4096, 64, 64)):
    a = torch.randn((R, C, n), dtype=torch.float64, device='cuda')
    rd = torch.zeros((C, n), dtype=torch.float64, device='cuda')
    w = a.clone()
    m.batched_qr_device(n, n, R, C, w.data_ptr(), rd.data_ptr()); torch.cuda.synchronize()
    q = w[:, :, :4].cpu().numpy().transpose(2, 0, 1); a4 = a[:, :, :4].cpu().numpy().transpose(2, 0, 1)
    err = max(np.abs(q[i] - np.linalg.qr(a4[i])[0]).max() for i in range(4))
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            w.copy_(a)
            m.batched_qr_device(n, n, R, C, w.data_ptr(), rd.data_ptr())
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 20)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        w.copy_(a)
    e1.record(); e1.synchronize(); tc = e0.elapsed_time(e1) / 20
    print('%d x %dx%d: %.4f ms per QR (copy %.4f ms subtracted), %s, max|Q - lapack| %.1e' % (n, R, C, np.median(ts) - tc, tc, m.last_kernel_info(), err), flush=True)
