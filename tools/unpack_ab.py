"""GPU box: qgs_unpack_records (R[record][inner][member] -> (member, inner, record)) at config-2 size, optionally with another
build of the library (RK_AB_LIB=)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qgs_amd import _lib
if os.environ.get('RK_AB_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['RK_AB_LIB'])
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'm36.npz')); ndim = int(g['ndim'])
m = _lib.HipModel(ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
st = torch.cuda.current_stream().cuda_stream
for n, inner, nrec in ((65536, 36, 101), (65536, 36, 7), (16384, 36 * 36, 11), (1000, 36, 1001)):
    src = torch.rand((nrec, inner, n), dtype=torch.float64, device='cuda')
    dst = torch.empty((n, inner, nrec), dtype=torch.float64, device='cuda')
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); m.unpack_records(n, n if n % 64 == 0 else (n + 63) // 64 * 64, inner, nrec, src.data_ptr(), dst.data_ptr(), st); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ok = bool(torch.equal(dst, src.permute(2, 1, 0))) if n % 64 == 0 else None
    print('%6d members x %5d inner x %5d records: %.3f ms = %.2f TB/s moved, correct %s' % (n, inner, nrec, min(ts[1:]), 2 * src.numel() * 8 / min(ts[1:]) / 1e9, ok))
