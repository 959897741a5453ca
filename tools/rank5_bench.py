"""Rank-5 (dynamic-T / T4) models on the GPU: stepper and tendencies rate of the specialised (derived-monomial) kernels
against the 4-factor generic ones.  python tools/rank5_bench.py [n_traj] [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from qgs_amd import _lib  # noqa: E402
if os.environ.get('RK_AB_LIB'):                    # another build of the library (developer knobs)
    _lib.LIB_PATH = os.path.abspath(os.environ['RK_AB_LIB'])

RK4 = dict(c=np.array([0., 0.5, 0.5, 1.]), b=np.array([1. / 6, 1. / 3, 1. / 3, 1. / 6]),
           a=np.array([[0., 0, 0, 0], [0.5, 0, 0, 0], [0, 0.5, 0, 0], [0, 0, 1., 0]]))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device('cuda', 0)
st = torch.cuda.current_stream().cuda_stream
gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')
for name in ('d38', 'q38'):
    g = np.load(os.path.join(gold, name + '.npz'))
    ndim = int(g['ndim'])
    m = _lib.HipModel(ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    print('%s: ndim %d, nnz %d, derived monomials %s' % (name, ndim, len(g['val']), m.n_derived))
    rng = np.random.RandomState(21217)
    ic = rng.rand(n, ndim) * 0.01
    ic[:, 10] += 1.5
    ic[:, 29] += 3.
    x = torch.from_numpy(np.ascontiguousarray(ic.T)).to(dev)
    rec = torch.empty_like(x)
    dx = torch.empty_like(x)
    t = np.concatenate((np.arange(0., steps * 0.1 - 1e-9, 0.1), [steps * 0.1]))
    nnz = len(g['val'])
    out = {}
    for kind, kname in ((2, 'specialised'), (1, 'generic')):
        m.set_kernel(kind)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m.rk_integrate_device(n, n, x.data_ptr(), t, 1, 0, RK4['b'], RK4['c'], RK4['a'], rec.data_ptr(), st)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        info = m.last_kernel_info()
        out[kind] = rec.clone()
        print('  %-11s rk4 %d x %d steps: %8.2f ms  %.3e traj-steps/s  (%s, %d VGPRs, %d B scratch)'
              % (kname, n, steps, best * 1e3, n * steps / best, info['name'], info['vgprs'], info['scratch_bytes']))
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m.tendencies_device(n, n, x.data_ptr(), dx.data_ptr(), st)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print('  %-11s f   %d evals: %8.3f ms  %.3e evals/s (%s)' % (kname, n, best * 1e3, n / best, m.last_kernel_info()['name']))
    print('  specialised vs generic final states: max rel diff %.1e'
          % float((out[2] - out[1]).abs().max() / out[1].abs().max()))
    m.close()
