#!/usr/bin/env python3
"""Developer script: time the RK4 stepper kernel alone (HIP events) for a few ensemble sizes / variants.
usage: kbench.py [members ...]   env: QGS_HIP_RK_VARIANT=plain|split, QGS_HIP_* codegen knobs"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qgs_amd import _lib

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'm36.npz'))
ndim = int(g['ndim'])
m = _lib.HipModel(ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
c = np.array([0., .5, .5, 1.]); b = np.array([1 / 6, 1 / 3, 1 / 3, 1 / 6]); a = np.zeros((4, 4)); a[1, 0] = .5; a[2, 1] = .5; a[3, 2] = 1.
steps = int(os.environ.get('KB_STEPS', '1000'))
t = np.concatenate((np.arange(0., steps * 0.1, 0.1), [steps * 0.1]))[:steps + 1]
dev = torch.device('cuda', 0)
for n in [int(x) for x in sys.argv[1:]] or [65536]:
    ld = (n + 63) // 64 * 64
    ic = torch.from_numpy(np.random.RandomState(1).rand(ndim, ld) * 0.01).to(dev)
    rec = torch.empty((1, ndim, ld), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for variant in os.environ.get('KB_VARIANTS', 'plain,split').split(','):
        os.environ['QGS_HIP_RK_VARIANT'] = variant
        for _ in range(2):
            m.rk_integrate_device(n, ld, ic.data_ptr(), t, 1, 0, b, c, a, rec.data_ptr(), st)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); m.rk_integrate_device(n, ld, ic.data_ptr(), t, 1, 0, b, c, a, rec.data_ptr(), st); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ms = float(np.median(ts))
        chk = float(rec.sum().item())
        print('%-6s n=%7d  %8.3f ms  %.3e traj-steps/s  %s  checksum %.12e' % (variant, n, ms, n * steps / ms * 1e3, m.last_kernel_info(), chk), flush=True)
