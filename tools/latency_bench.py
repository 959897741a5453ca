#!/usr/bin/env python3
"""Developer script: trajectory-steps/s as a function of the ensemble size, wave-per-trajectory kernel vs
one-member-per-lane kernels (crossover measurement for QGS_HIP_WAVE_MAX_TRAJ)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import model_configs
from qgs_amd.functions.tendencies import create_tendencies
name = sys.argv[1] if len(sys.argv) > 1 else 'm36'
p = dict(model_configs.MAKERS, **model_configs.MAKERS_RANK5)[name]()
f, Df = create_tendencies(p); m = f.hip_model(); ndim = p.ndim
c = np.array([0., .5, .5, 1.]); b = np.array([1 / 6, 1 / 3, 1 / 3, 1 / 6]); a = np.zeros((4, 4)); a[1, 0] = .5; a[2, 1] = .5; a[3, 2] = 1.
steps = int(os.environ.get('KB_STEPS', '2000'))
t = np.concatenate((np.arange(0., steps * 0.1, 0.1), [steps * 0.1]))[:steps + 1]
dev = torch.device('cuda', 0)
for n in [int(x) for x in os.environ.get("LB_SIZES", "1,16,64,256,1024,2048,4096,8192,16384").split(",")]:
    ld = (n + 63) // 64 * 64
    ic0 = np.random.RandomState(1).rand(ndim, ld) * 0.01
    if p.dynamic_T:
        ic0[p.variables_range[0]] += 1.5
        ic0[p.variables_range[2]] += 3.
    ic = torch.from_numpy(ic0).to(dev)
    rec = torch.empty((1, ndim, ld), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for label, env in (('wave', '1000000'), ('lane', '0')):
        os.environ['QGS_HIP_WAVE_MAX_TRAJ'] = env
        m.set_kernel(0)                    # the selection knobs are read at set_kernel / model creation
        m.rk_integrate_device(n, ld, ic.data_ptr(), t, 1, 0, b, c, a, rec.data_ptr(), st); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); m.rk_integrate_device(n, ld, ic.data_ptr(), t, 1, 0, b, c, a, rec.data_ptr(), st); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        out.append('%s %8.3f ms %.3e/s (%s) chk %.10e' % (label, ms, n * steps / ms * 1e3, m.last_kernel_info()['name'], float(rec[:, :, :n].sum())))
    print('n=%6d  %s' % (n, ' | '.join(out)), flush=True)
