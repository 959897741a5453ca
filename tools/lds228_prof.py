#!/usr/bin/env python3
"""Developer script for profiling: run the ndim-228 stepper once warm + `reps` times at `members` members x `steps` steps.
usage: lds228_prof.py <kind 1|2> <members> <steps> [reps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qgs_amd import _lib  # noqa: E402
if os.environ.get("RK_AB_LIB"):                    # another build of the library (developer knobs)
    _lib.LIB_PATH = os.path.abspath(os.environ["RK_AB_LIB"])

kind, n, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
c = np.array([0., .5, .5, 1.]); b = np.array([1 / 6, 1 / 3, 1 / 3, 1 / 6]); a = np.zeros((4, 4)); a[1, 0] = .5; a[2, 1] = .5; a[3, 2] = 1.
g = np.load(os.path.join(ROOT, 'tests', 'golden', 't228.npz'))
ndim = int(g['ndim'])
m = _lib.HipModel(ndim, g['coo'], g['val'], g['jcoo'], g['jval'], device=0)
m.set_kernel(kind)
dev = torch.device('cuda', 0)
t = np.concatenate((np.arange(0., steps * 0.1, 0.1), [steps * 0.1]))[:steps + 1]
ic = torch.from_numpy(np.random.RandomState(3).rand(ndim, n) * 0.01).to(dev)
rec = torch.empty((1, ndim, n), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
ts = []
for _ in range(reps + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.rk_integrate_device(n, n, ic.data_ptr(), t, 1, 0, b, c, a, rec.data_ptr(), st)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
el = min(ts[1:])
print('%s members %d steps %d: %.3f ms  %.3e traj-steps/s  fp64 frac %.3f' % (m.last_kernel_info()['name'], n, steps, el * 1e3, n * steps / el,
                                                                               n * steps / el * 336336 / 78.6e12), flush=True)
