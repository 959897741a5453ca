import os, sys
import numpy as np, torch
ROOT = '/root/repo' if os.path.isdir('/root/repo/qgs_amd') else os.environ.get('GRAFT_REPO_ROOT', '.')
sys.path.insert(0, ROOT)
from qgs_amd import _lib
if os.environ.get('RK_AB_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['RK_AB_LIB'])
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'm36.npz'))
m = _lib.HipModel(int(g['ndim']), g['coo'], g['val'], g['jcoo'], g['jval'])
SHAPES = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(20, 20), (24, 24), (28, 28), (20, 5), (36, 20), (36, 5), (30, 30)]
for n, R, C in [(16384, r, c) for r, c in SHAPES]:
    a = torch.randn((R, C, n), dtype=torch.float64, device='cuda'); rd = torch.zeros((C, n), dtype=torch.float64, device='cuda'); w = a.clone()
    m.batched_qr_device(n, n, R, C, w.data_ptr(), rd.data_ptr()); torch.cuda.synchronize()
    q = w[:, :, :2].cpu().numpy().transpose(2, 0, 1); a2 = a[:, :, :2].cpu().numpy().transpose(2, 0, 1)
    err = max(np.abs(q[i] - np.linalg.qr(a2[i])[0]).max() for i in range(2))
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w.copy_(a); e0.record(); m.batched_qr_device(n, n, R, C, w.data_ptr(), rd.data_ptr()); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    i = m.last_kernel_info()
    print('%d x %dx%d: %.4f ms vgpr %d lds %d scratch %d err %.1e' % (n, R, C, float(np.median(ts)), i['vgprs'], i['lds_bytes'], i['scratch_bytes'], err), flush=True)
