"""GPU box: how records reach the host -- stores of the unpack kernel into the page-locked result block ('kernel') against
unpack into device staging + (strided) copy ('copy'), for BASELINE config 2 with write_steps=1 (65 536 members, 100 steps:
1.9 GB of records), in one window and cut into windows; and the plain page-locked copy of the same bytes as the PCIe floor."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from qgs_amd import _lib                                                      # noqa: E402
from bench import load_model_tensors, rk4_tableau, grid                      # noqa: E402

ndim, coo, val, jcoo, jval, _ = load_model_tensors()
b, c, a = rk4_tableau()
n, steps = 65536, (int(sys.argv[1]) if len(sys.argv) > 1 else 100)
t = grid(steps, 0.1)
ic = np.random.RandomState(21217).rand(n, ndim) * 0.01


def wall(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


nbytes = n * ndim * (steps + 1) * 8
# PCIe floor: one page-locked device-to-host copy of the record's bytes
d = torch.empty(nbytes // 8, dtype=torch.float64, device='cuda')
h = torch.empty(nbytes // 8, dtype=torch.float64).pin_memory()
ms = wall(lambda: (h.copy_(d, non_blocking=True), torch.cuda.synchronize()))
print('plain page-locked D2H of %.2f GB: %.1f ms = %.1f GB/s' % (nbytes / 1e9, ms, nbytes / ms / 1e6))
del d, h
ref = None
modes = ('kernel', 'copy') if len(sys.argv) <= 2 else tuple(sys.argv[2].split(','))
for mode in modes:
    for mb in ((None, 2048, 512, 128) if len(sys.argv) <= 3 else tuple(None if q == 'dflt' else int(q) for q in sys.argv[3].split(','))):
        os.environ['QGS_HIP_D2H'] = mode
        if mb is None:
            os.environ.pop('QGS_HIP_RECORD_WINDOW_MB', None)
        else:
            os.environ['QGS_HIP_RECORD_WINDOW_MB'] = str(mb)
        m = _lib.HipModel(ndim, coo, val, jcoo, jval)
        out = m.rk_integrate(t, ic, 1, 1, b, c, a)
        if ref is None:
            ref = np.array(out)
        same = bool(np.array_equal(out, ref))
        del out
        ms = wall(lambda: m.rk_integrate(t, ic, 1, 1, b, c, a))
        print('route %-6s window budget %-5s MB: %3d windows  %7.1f ms wall  = %5.1f GB/s of records  bitwise equal %s'
              % (mode, mb or 'dflt', m.last_windows, ms, nbytes / ms / 1e6, same))
        m.close()
