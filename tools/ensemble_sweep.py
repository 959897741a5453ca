"""GPU box: trajectory-steps/s of the automatic kernel choice as a function of the ensemble size (MAOOAM-36 and MAOOAM 6x6),
with the kernel the library picked."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from qgs_amd import _lib
c = np.array([0., .5, .5, 1.]); b = np.array([1 / 6, 1 / 3, 1 / 3, 1 / 6]); a = np.zeros((4, 4)); a[1, 0] = .5; a[2, 1] = .5; a[3, 2] = 1.
st = torch.cuda.current_stream().cuda_stream
for name, steps, sizes in (('m36', 1000, (1, 64, 1024, 2048, 4096, 16384, 32768, 65536, 131072, 262144, 1048576)),
                           ('t228', 100, (1, 64, 512, 1024, 4096, 16384, 65536, 131072))):
    g = np.load(os.path.join(REPO, 'tests', 'golden', name + '.npz')); ndim = int(g['ndim'])
    m = _lib.HipModel(ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    t = np.concatenate((np.arange(0., steps * 0.1, 0.1), [steps * 0.1]))[:steps + 1]
    print('%s (ndim %d), %d RK4 steps per launch, write_steps=0' % (name, ndim, steps))
    for n in sizes:
        ld = (n + 63) // 64 * 64
        ic = torch.from_numpy(np.random.RandomState(1).rand(ndim, ld) * 0.01).cuda()
        rec = torch.empty((1, ndim, ld), dtype=torch.float64, device='cuda')
        ts = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); m.rk_integrate_device(n, ld, ic.data_ptr(), t, 1, 0, b, c, a, rec.data_ptr(), st); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = min(ts[1:])
        print('    %8d members  %9.3f ms  %.3e traj-steps/s  %s' % (n, ms, n * steps / ms * 1e3, m.last_kernel_info()['name']), flush=True)
    m.close()
