"""Time only: qgs_spec_rklds16 on the MAOOAM 6x6 tensor, 65 536 members x 100 steps (generator knobs through the environment,
see INTEGRATION.md)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qgs_amd import _lib
if os.environ.get('RK_AB_LIB'):                    # another build of the library (A/B across generator versions)
    _lib.LIB_PATH = os.path.abspath(os.environ['RK_AB_LIB'])
c = np.array([0., .5, .5, 1.]); b = np.array([1 / 6, 1 / 3, 1 / 3, 1 / 6]); a = np.zeros((4, 4)); a[1, 0] = .5; a[2, 1] = .5; a[3, 2] = 1.
g = np.load(os.path.join(ROOT, 'tests', 'golden', 't228.npz')); ndim = int(g['ndim'])
m = _lib.HipModel(ndim, g['coo'], g['val'], g['jcoo'], g['jval']); m.set_kernel(2)
steps = 100; t = np.concatenate((np.arange(0., steps * 0.1, 0.1), [steps * 0.1]))[:steps + 1]
st = torch.cuda.current_stream().cuda_stream
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ic = torch.from_numpy(np.random.RandomState(3).rand(ndim, n) * 0.01).cuda(); rec = torch.empty((1, ndim, n), dtype=torch.float64, device='cuda')
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.rk_integrate_device(n, n, ic.data_ptr(), t, 1, 0, b, c, a, rec.data_ptr(), st)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
info = m.last_kernel_info()
clk = m.kernel_clock()                     # (shader GHz, ms) of the last launch: in-kernel probe
# numerics of the kernel just timed: 200 members x 12 steps (records every 4) against the generic tiled kernel
nc, sc = 200, 12
tc = np.concatenate((np.arange(0., sc * 0.1, 0.1), [sc * 0.1]))[:sc + 1]
icc = np.random.RandomState(5).rand(nc, ndim) * 0.01
a_spec = m.rk_integrate(tc, icc, 1, 4, b, c, a)
m.set_kernel(1)
a_gen = m.rk_integrate(tc, icc, 1, 4, b, c, a)
print('%.2f ms at %.2f GHz %s  max rel diff vs %s: %.1e' % (min(ts[1:]) * 1e3, clk[0] if clk else 0., info, m.last_kernel_info()['name'], float(np.abs(a_spec - a_gen).max() / np.abs(a_gen).max())), flush=True)
