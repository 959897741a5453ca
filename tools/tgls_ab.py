"""Developer script: config 4 (16 384 members x 36 tangent vectors x 10 sub-steps), 100 calls back to back timed with HIP
events -- A/B over generator knobs given as NAME=VALUE arguments (each variant in its own model, same process)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qgs_amd import _lib
if os.environ.get('RK_AB_LIB'):                    # another build of the library (tools/build_prev_lib.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ['RK_AB_LIB'])
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'm36.npz')); ndim = int(g['ndim'])
c = np.array([0., .5, .5, 1.]); b = np.array([1 / 6, 1 / 3, 1 / 3, 1 / 6]); a = np.zeros((4, 4)); a[1, 0] = .5; a[2, 1] = .5; a[3, 2] = 1.
n, steps, n_tg, calls = 16384, 10, ndim, 100
t = np.concatenate((np.arange(0., steps * 0.01, 0.01), [steps * 0.01]))[:steps + 1]
st = torch.cuda.current_stream().cuda_stream
ic = torch.from_numpy(np.random.RandomState(2).rand(ndim, n) * 0.01).cuda()
tg = torch.zeros((ndim, n_tg, n), dtype=torch.float64, device='cuda')
for d in range(ndim):
    tg[d, d, :] = 1.0
rec = torch.empty((1, ndim, n), dtype=torch.float64, device='cuda'); recm = torch.empty((1, ndim, n_tg, n), dtype=torch.float64, device='cuda')
ref = None
for rep in range(2):
    for var in (sys.argv[1:] or ['A=0']):
        for kv in var.split(','):
            k, v = kv.split('='); os.environ[k] = v
        m = _lib.HipModel(ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
        def run():
            for _ in range(calls):
                m.rk_tgls_integrate_device(n, n, n_tg, ic.data_ptr(), tg.data_ptr(), t, 1, 0, b, c, a, False, 1., rec.data_ptr(), recm.data_ptr(), st)
        run(); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / calls)
        out = recm.clone()
        if ref is None:
            ref = out
        print('%-40s %.4f ms per call (min %.4f)  %s  max|diff| vs first %.2e' % (var, np.median(ts), min(ts), m.last_kernel_info(), float((out - ref).abs().max())), flush=True)
        for kv in var.split(','):
            os.environ.pop(kv.split('=')[0], None)
        del m
