"""GPU box: the tangent model's records (trajectory + propagators) of a large ensemble through the host-pointer API into ordinary
NumPy memory: 16 384 members x 36 tangent vectors, every step a record.  Usage: tgls_big_record.py [steps, default 100]."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from qgs_amd import _lib                                                      # noqa: E402
from bench import load_model_tensors, rk4_tableau, grid                      # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n, n_tg = 16384, 36
ndim, coo, val, jcoo, jval, _ = load_model_tensors()
b, c, a = rk4_tableau()
t = grid(steps, 0.1)
rng = np.random.RandomState(5)
ic = rng.rand(n, ndim) * 0.01
tg = np.broadcast_to(np.eye(ndim)[:, :n_tg], (n, ndim, n_tg)).copy()
m = _lib.HipModel(ndim, coo, val, jcoo, jval)
m.rk_tgls_integrate(t[:3], ic[:256], tg[:256], 1, 1, b, c, a, False, 1.)          # warm-up
out = (np.empty((n, ndim, steps + 1)), np.empty((n, ndim, n_tg, steps + 1)))
nbytes = out[0].nbytes + out[1].nbytes
for rep in range(3):
    t0 = time.perf_counter()
    m.rk_tgls_integrate(t, ic, tg, 1, 1, b, c, a, False, 1., out=out)
    el = time.perf_counter() - t0
    print('run %d: %.1f GB of records, %d member group(s), %d window(s) each, %.3f s = %.1f GB/s' % (rep, nbytes / 1e9, m.last_groups, m.last_windows, el, nbytes / el / 1e9), flush=True)
print('record 0 of the propagators is the start matrix:', bool(np.array_equal(out[1][:, :, :, 0], tg)), '; finite:', bool(np.isfinite(out[1].reshape(-1)[::9973]).all()))
