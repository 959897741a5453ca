import os, sys
import numpy as np, torch
ROOT='/root/repo'
sys.path.insert(0, ROOT)
from qgs_amd import _lib
if os.environ.get('RK_AB_LIB'):                    # another build of the library (tools/build_prev_lib.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ['RK_AB_LIB'])
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'm36.npz')); ndim = int(g['ndim'])
c = np.array([0., .5, .5, 1.]); b = np.array([1 / 6, 1 / 3, 1 / 3, 1 / 6]); a = np.zeros((4, 4)); a[1, 0] = .5; a[2, 1] = .5; a[3, 2] = 1.
n, steps = 65536, 100
t = np.concatenate((np.arange(0., steps * 0.1, 0.1), [steps * 0.1]))[:steps + 1]
st = torch.cuda.current_stream().cuda_stream
ic = torch.from_numpy(np.random.RandomState(1).rand(ndim, n) * 0.01).cuda()
rec = torch.zeros((steps + 1, ndim, n), dtype=torch.float64, device='cuda')
for rep in range(2):
    for var in sys.argv[1:]:
        for kv in var.split(','):
            k, v = kv.split('=', 1); os.environ[k] = v
        m = _lib.HipModel(ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
        def run():
            for _ in range(10):
                m.rk_integrate_device(n, n, ic.data_ptr(), t, 1, 1, b, c, a, rec.data_ptr(), st)
        run(); torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 10)
        print('%-40s %.4f ms (min %.4f) %s' % (var, np.median(ts), min(ts), m.last_kernel_info()['name']), flush=True)
        for kv in var.split(','):
            os.environ.pop(kv.split('=')[0], None)
        del m
