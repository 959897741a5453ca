"""Developer script (GPU box): the blocked batched QR (cols > 64: `batched_qr_blocked_kernel`, dgeqrf + dorgqr with 16-column panels) at the
full-spectrum shape of MAOOAM 6x6, 228 x 228.  With a developer build (make DEV=1) QGS_HIP_QRB_SKIP=1|2|4 (sum of them) leaves out the
trailing updates / the second phase / the panel factorisations: where the time is (profiles/r05_qr.md section 11)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qgs_amd import _lib
if os.environ.get('RK_AB_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['RK_AB_LIB'])
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'm36.npz'))
m = _lib.HipModel(int(g['ndim']), g['coo'], g['val'], g['jcoo'], g['jval'])
for n, R, C in ((1024, 228, 228), (4096, 228, 228), (512, 228, 228)):
    a = torch.randn((R, C, n), dtype=torch.float64, device='cuda'); rd = torch.zeros((C, n), dtype=torch.float64, device='cuda'); w = a.clone()
    m.batched_qr_device(n, n, R, C, w.data_ptr(), rd.data_ptr()); torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w.copy_(a); e0.record(); m.batched_qr_device(n, n, R, C, w.data_ptr(), rd.data_ptr()); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    print('skip %s: %d x %dx%d: %.3f ms' % (os.environ.get('QGS_HIP_QRB_SKIP', '0'), n, R, C, float(np.median(ts))), flush=True)
