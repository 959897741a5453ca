#!/usr/bin/env python3
"""Developer script: measure the BASELINE.json configurations other than the headline one
(config 2 through the host-pointer API incl. PCIe, config 2 with full record, config 3 = MAOOAM-228,
config 4 = TGLS).  Prints one JSON object per measurement."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from qgs_amd import _lib  # noqa: E402
from qgs_amd.functions.tendencies import create_tendencies  # noqa: E402
import model_configs  # noqa: E402

c = np.array([0., .5, .5, 1.]); b = np.array([1 / 6, 1 / 3, 1 / 3, 1 / 6]); a = np.zeros((4, 4)); a[1, 0] = .5; a[2, 1] = .5; a[3, 2] = 1.
dev = torch.device('cuda', 0)
which = sys.argv[1:] or ['host', 'record', 'tgls', 't228', 'moments']


def grid(steps, dt=0.1):
    return np.concatenate((np.arange(0., steps * dt, dt), [steps * dt]))[:steps + 1]


def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


if 'host' in which or 'record' in which or 'tgls' in which:
    p = model_configs.params_m36()
    f, Df = create_tendencies(p)
    m = f.hip_model()
    ndim = p.ndim

if 'host' in which:
    n, steps = 65536, 1000
    ic = np.random.RandomState(21217).rand(n, ndim) * 0.01
    t = grid(steps)
    el = timeit(lambda: m.rk_integrate(t, ic, 1, 0, b, c, a))
    print(json.dumps({'case': 'config2 host-pointer API (H2D + kernel + D2H, pageable)', 'seconds': el,
                      'traj_steps_per_s': n * steps / el}))

if 'record' in which:
    n, steps = 65536, 100
    ld = n
    t = grid(steps)
    ic = torch.from_numpy(np.random.RandomState(1).rand(ndim, ld) * 0.01).to(dev)
    rec = torch.empty((steps + 1, ndim, ld), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    el = timeit(lambda: m.rk_integrate_device(n, ld, ic.data_ptr(), t, 1, 1, b, c, a, rec.data_ptr(), st))
    print(json.dumps({'case': 'config2 write_steps=1, 100 steps, 1.9 GB record (device layout)', 'seconds': el,
                      'traj_steps_per_s': n * steps / el, 'record_GBs': rec.numel() * 8 / el / 1e9,
                      'kernel': m.last_kernel_info()}))
    del rec

if 'tgls' in which:
    n, steps, n_tg = 16384, 10, 36
    ld = n
    t = grid(steps, 0.01)
    ic = torch.from_numpy(np.random.RandomState(2).rand(ndim, ld) * 0.01).to(dev)
    tg = torch.zeros((ndim, n_tg, ld), dtype=torch.float64, device=dev)
    for d in range(ndim):
        tg[d, d, :] = 1.0
    rec = torch.empty((1, ndim, ld), dtype=torch.float64, device=dev)
    recm = torch.empty((1, ndim, n_tg, ld), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for kind, kname in ((2, 'spec'), (1, 'generic')):
        m.set_kernel(kind)
        el = timeit(lambda: m.rk_tgls_integrate_device(n, ld, n_tg, ic.data_ptr(), tg.data_ptr(), t, 1, 0, b, c, a, False, 1.,
                                                       rec.data_ptr(), recm.data_ptr(), st))
        print(json.dumps({'case': 'config4 TGLS %s: 16384 members x 36 tangent vectors x 10 steps' % kname, 'seconds': el,
                          'traj_steps_per_s': n * steps / el, 'hbm_frac_algorithmic': n * steps / el * 21312 / 8e12,
                          'fp64_dense_flop_frac': n * steps / el * 4.02e5 / 78.6e12, 'kernel': m.last_kernel_info()}))
    m.set_kernel(0)

if 't228' in which:
    p = model_configs.params_t228()
    f, Df = create_tendencies(p)
    m = f.hip_model()
    ndim = p.ndim
    steps = 100
    t = grid(steps)
    st = torch.cuda.current_stream().cuda_stream
    for n in (4096, 65536):
        ld = n
        ic = torch.from_numpy(np.random.RandomState(3).rand(ndim, ld) * 0.01).to(dev)
        rec = torch.empty((1, ndim, ld), dtype=torch.float64, device=dev)
        el = timeit(lambda: m.rk_integrate_device(n, ld, ic.data_ptr(), t, 1, 0, b, c, a, rec.data_ptr(), st), n=2)
        print(json.dumps({'case': 'config3 MAOOAM-228 (%s), %d members x 100 steps' % (m.last_kernel_info()['name'], n), 'seconds': el,
                          'traj_steps_per_s': n * steps / el, 'fp64_flop_frac': n * steps / el * 336336 / 78.6e12,
                          'hbm_frac_algorithmic': n * steps / el * 3648 / 8e12, 'kernel': m.last_kernel_info()}), flush=True)

if 'moments' in which:
    # ensemble mean + variance of a device-resident record: 65 536 members x 36 modes x 101 records (1.9 GB)
    p = model_configs.params_m36()
    f, Df = create_tendencies(p)
    m = f.hip_model()
    n, ndim, nrec = 65536, p.ndim, 101
    rec = torch.rand((nrec * ndim, n), dtype=torch.float64, device=dev)
    mean = torch.empty(nrec * ndim, dtype=torch.float64, device=dev)
    var = torch.empty(nrec * ndim, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    el = timeit(lambda: m.ensemble_moments_device(n, n, nrec * ndim, rec.data_ptr(), mean.data_ptr(), var.data_ptr(), st), n=5)
    ok = float((mean - rec.mean(dim=1)).abs().max()), float((var - rec.var(dim=1, unbiased=False)).abs().max())
    print(json.dumps({'case': 'ensemble moments of a 65536 x 36 x 101 device record', 'seconds': el, 'GBs': rec.numel() * 8 / el / 1e9,
                      'hbm_frac': rec.numel() * 8 / el / 8e12, 'max_abs_err_mean_var_vs_torch': ok}), flush=True)
