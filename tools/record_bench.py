"""Developer script: config 2 with the reference's default write_steps=1 (every step a record): kernel time per 100 steps at
65 536 members with the burst stores (qgs_spec_rk_s4) and with the stores spread over the step (qgs_spec_rkr_s4), and parity
of the two record buffers."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from qgs_amd import _lib
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'm36.npz')); ndim = int(g['ndim'])
c = np.array([0., .5, .5, 1.]); b = np.array([1 / 6, 1 / 3, 1 / 3, 1 / 6]); a = np.zeros((4, 4)); a[1, 0] = .5; a[2, 1] = .5; a[3, 2] = 1.
n, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 100
t = np.concatenate((np.arange(0., steps * 0.1, 0.1), [steps * 0.1]))[:steps + 1]
st = torch.cuda.current_stream().cuda_stream
ic = torch.from_numpy(np.random.RandomState(1).rand(ndim, n) * 0.01).cuda()
out = {}
for spread in ('0', '1'):
    os.environ['QGS_HIP_RK_SPREAD_REC'] = spread
    m = _lib.HipModel(ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    rec = torch.zeros((steps + 1, ndim, n), dtype=torch.float64, device='cuda')
    for direction in (1, -1):
        ts = []
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); m.rk_integrate_device(n, n, ic.data_ptr(), t, direction, 1, b, c, a, rec.data_ptr(), st); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        print('spread=%s direction %+d: %.3f ms (min %.3f)  %s  record writes %.2f TB/s' % (spread, direction, np.median(ts[1:]), min(ts), m.last_kernel_info()['name'],
                                                                                   rec.numel() * 8 / (np.median(ts[1:]) * 1e-3) / 1e12), flush=True)
        out[(spread, direction)] = rec.clone()
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); m.rk_integrate_device(n, n, ic.data_ptr(), t, 1, 0, b, c, a, rec.data_ptr(), st); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    print('   write_steps=0: %.3f ms' % np.median(ts[1:]))
    del m
for d in (1, -1):
    print('direction %+d max |diff| burst vs spread: %g' % (d, float((out[('0', d)] - out[('1', d)]).abs().max())))
