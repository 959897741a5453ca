#!/usr/bin/env python3
"""Developer script: the JIT LDS-resident stepper (qgs_spec_rklds<W>) against the generic tiled kernel on the
MAOOAM 6x6 tensor (ndim 228): parity on a small run with records, then timing at 4096 / 65536 members x 100 steps."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qgs_amd import _lib  # noqa: E402

c = np.array([0., .5, .5, 1.]); b = np.array([1 / 6, 1 / 3, 1 / 3, 1 / 6]); a = np.zeros((4, 4)); a[1, 0] = .5; a[2, 1] = .5; a[3, 2] = 1.
dev = torch.device('cuda', 0)
g = np.load(os.path.join(ROOT, 'tests', 'golden', 't228.npz'))
ndim = int(g['ndim'])
m = _lib.HipModel(ndim, g['coo'], g['val'], g['jcoo'], g['jval'], device=0)


def grid(steps, dt=0.1):
    return np.concatenate((np.arange(0., steps * dt, dt), [steps * dt]))[:steps + 1]


rng = np.random.RandomState(5)
x = rng.rand(200, ndim) * 0.01
t = grid(11)
m.set_kernel(1)
ref = m.rk_integrate(t, x, 1, 3, b, c, a)
print('generic:', m.last_kernel_info()['name'], flush=True)
m.set_kernel(2)
t0 = time.perf_counter()
out = m.rk_integrate(t, x, 1, 3, b, c, a)
print('lds:', m.last_kernel_info(), 'first call %.1f s' % (time.perf_counter() - t0), flush=True)
err = float(np.abs(out - ref).max() / np.abs(ref).max())
print('parity forward rel err %.2e shape %s' % (err, out.shape), flush=True)
ref_b = None
m.set_kernel(1); ref_b = m.rk_integrate(t, x, -1, 2, b, c, a)
m.set_kernel(2); out_b = m.rk_integrate(t, x, -1, 2, b, c, a)
print('parity backward rel err %.2e' % float(np.abs(out_b - ref_b).max() / np.abs(ref_b).max()), flush=True)
b2 = np.array([0., 1.]); c2 = np.array([0., .5]); a2 = np.zeros((2, 2)); a2[1, 0] = .5
m.set_kernel(1); r2 = m.rk_integrate(t, x[:70], 1, 0, b2, c2, a2)
m.set_kernel(2); o2 = m.rk_integrate(t, x[:70], 1, 0, b2, c2, a2)
print('parity rk2 rel err %.2e' % float(np.abs(o2 - r2).max() / np.abs(r2).max()), flush=True)
assert err < 1e-12

steps = 100
t = grid(steps)
st = torch.cuda.current_stream().cuda_stream
for n in (4096, 65536):
    ld = n
    ic = torch.from_numpy(np.random.RandomState(3).rand(ndim, ld) * 0.01).to(dev)
    rec = torch.empty((1, ndim, ld), dtype=torch.float64, device=dev)
    for kind in (1, 2):
        m.set_kernel(kind)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m.rk_integrate_device(n, ld, ic.data_ptr(), t, 1, 0, b, c, a, rec.data_ptr(), st)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        el = float(np.median(ts[1:]))
        print(json.dumps({'members': n, 'kernel': m.last_kernel_info(), 'seconds': el, 'traj_steps_per_s': n * steps / el,
                          'fp64_flop_frac': n * steps / el * 336336 / 78.6e12}), flush=True)
