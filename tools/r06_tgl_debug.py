"""Developer script: the hand-scheduled tangent kernel against the compiler-scheduled one on a golden model, row by row (dev build)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qgs_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'qgs_amd', 'libqgs_hip_dev.so')
name = sys.argv[1] if len(sys.argv) > 1 else 'rp20'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz')); ndim = int(g['ndim'])
c = np.array([0., .5, .5, 1.]); b = np.array([1 / 6, 1 / 3, 1 / 3, 1 / 6]); a = np.zeros((4, 4)); a[1, 0] = .5; a[2, 1] = .5; a[3, 2] = 1.
n, n_tg = 4096, ndim
t = np.concatenate((np.arange(0., steps * 0.01, 0.01), [steps * 0.01]))[:steps + 1]
st = torch.cuda.current_stream().cuda_stream
ic = torch.from_numpy(np.random.RandomState(2).rand(ndim, n) * 0.01).cuda()
tg = torch.zeros((ndim, n_tg, n), dtype=torch.float64, device='cuda')
for d in range(ndim):
    tg[d, d, :] = 1.0
out = {}
for asm in ('0', '1'):
    os.environ['QGS_HIP_TGL_ASM'] = asm
    m = _lib.HipModel(ndim, g['coo'], g['val'], g['jcoo'], g['jval']); m.set_kernel(2)
    for adj in (False, True):
        rec = torch.empty((1, ndim, n), dtype=torch.float64, device='cuda'); recm = torch.empty((1, ndim, n_tg, n), dtype=torch.float64, device='cuda')
        m.rk_tgls_integrate_device(n, n, n_tg, ic.data_ptr(), tg.data_ptr(), t, 1, 0, b, c, a, adj, 1., rec.data_ptr(), recm.data_ptr(), st)
        torch.cuda.synchronize()
        out[(asm, adj)] = recm[0].cpu().numpy()
        print(asm, adj, m.last_kernel_info()['name'])
for adj in (False, True):
    d = np.abs(out[('1', adj)] - out[('0', adj)])
    print('adjoint', adj, 'max diff', d.max(), 'rows with diff:', [(i + 1, float(d[i].max())) for i in range(ndim) if d[i].max() > 0][:40])
    print('   columns with diff:', [(j + 1, float(d[:, j].max())) for j in range(n_tg) if d[:, j].max() > 0][:40])
    print('   members with diff: %d of %d; first lanes %s' % (int((d.max(axis=(0, 1)) > 0).sum()), n, np.nonzero(d.max(axis=(0, 1)) > 0)[0][:20].tolist()))
np.savez(os.path.join(ROOT, 'gpurun_out', 'r06_tgl_debug_%s.npz' % name), asm=out[('1', False)][:, :, :64], cpp=out[('0', False)][:, :, :64],
         asm_adj=out[('1', True)][:, :, :64], cpp_adj=out[('0', True)][:, :, :64], ic=ic.cpu().numpy()[:, :64])
