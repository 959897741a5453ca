#!/usr/bin/env python3
"""Developer script: Benettin workload at BASELINE config 4 scale: members x n_vec basis columns, one dt interval =
`sub` TGLS sub-steps + one batched QR; reports the time of each part."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import model_configs
from qgs_amd.functions.tendencies import create_tendencies
p = model_configs.params_m36(); f, Df = create_tendencies(p); m = f.hip_model(); ndim = 36
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 36
sub = 10
c = np.array([0., .5, .5, 1.]); b = np.array([1 / 6, 1 / 3, 1 / 3, 1 / 6]); a = np.zeros((4, 4)); a[1, 0] = .5; a[2, 1] = .5; a[3, 2] = 1.
ld = (n + 63) // 64 * 64
dev = torch.device('cuda', 0)
ic = torch.from_numpy(np.random.RandomState(2).rand(ndim, ld) * 0.01).to(dev)
q = torch.randn((ndim, nv, ld), dtype=torch.float64, device=dev)
qn = torch.empty((1, ndim, nv, ld), dtype=torch.float64, device=dev)
yend = torch.empty((1, ndim, ld), dtype=torch.float64, device=dev)
rd = torch.empty((nv, ld), dtype=torch.float64, device=dev)
t = np.concatenate((np.arange(0., 0.1, 0.01), [0.1]))
st = torch.cuda.current_stream().cuda_stream
def tgls(): m.rk_tgls_integrate_device(n, ld, nv, ic.data_ptr(), q.data_ptr(), t, 1, 0, b, c, a, False, 1., yend.data_ptr(), qn.data_ptr(), st)
def qr(): m.batched_qr_device(n, ld, ndim, nv, qn.data_ptr(), rd.data_ptr(), st)
for name, fn in (('tgls (%d sub-steps)' % sub, tgls), ('batched QR', qr)):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print('%-22s n=%d n_vec=%d  %.3f ms' % (name, n, nv, float(np.median(ts))), flush=True)
