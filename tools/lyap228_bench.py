import os, sys, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from qgs_amd import _lib
g = np.load('/root/repo/tests/golden/t228.npz'); ndim = 228
m = _lib.HipModel(ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
c = np.array([0., .5, .5, 1.]); b = np.array([1 / 6, 1 / 3, 1 / 3, 1 / 6]); a = np.zeros((4, 4)); a[1, 0] = .5; a[2, 1] = .5; a[3, 2] = 1.
dev = torch.device('cuda', 0); st = torch.cuda.current_stream().cuda_stream
for n, nv in ((1, 228), (64, 64), (1024, 40), (4096, 40), (4096, 10), (1024, 228)):
    ld = (n + 63) // 64 * 64
    ic = torch.from_numpy(np.random.RandomState(2).rand(ndim, ld) * 0.01).to(dev)
    q = torch.randn((ndim, nv, ld), dtype=torch.float64, device=dev)
    qn = torch.empty((1, ndim, nv, ld), dtype=torch.float64, device=dev)
    yend = torch.empty((1, ndim, ld), dtype=torch.float64, device=dev)
    rd = torch.empty((nv, ld), dtype=torch.float64, device=dev)
    t = np.concatenate((np.arange(0., 0.1 - 1e-12, 0.01), [0.1]))
    def tgls(): m.rk_tgls_integrate_device(n, ld, nv, ic.data_ptr(), q.data_ptr(), t, 1, 0, b, c, a, False, 1., yend.data_ptr(), qn.data_ptr(), st)
    def qr(): m.batched_qr_device(n, ld, ndim, nv, qn.data_ptr(), rd.data_ptr(), st)
    for name, fn in (('tgls', tgls), ('qr', qr)):
        fn(); torch.cuda.synchronize(); ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        print('n=%d n_vec=%d %-5s %.3f ms (%s)' % (n, nv, name, min(ts), m.last_kernel_info()['name']), flush=True)
