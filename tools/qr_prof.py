"""Developer script (GPU box, under rocprofv3): the batched QR of BASELINE config 4's Benettin step, 16 384 matrices of 36 x 36,
12 launches on fresh copies of one random input (tools/r05_qr_pmc.sh collects the counters)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qgs_amd import _lib
if os.environ.get('RK_AB_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['RK_AB_LIB'])
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'm36.npz')); ndim = int(g['ndim'])
m = _lib.HipModel(ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
n, R, C = 16384, 36, 36
a = torch.randn((R, C, n), dtype=torch.float64, device='cuda')
rd = torch.zeros((C, n), dtype=torch.float64, device='cuda')
w = a.clone()
for _ in range(12):
    w.copy_(a)
    m.batched_qr_device(n, n, R, C, w.data_ptr(), rd.data_ptr())
torch.cuda.synchronize()
print(m.last_kernel_info())
