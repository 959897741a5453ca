"""GPU box: config 5 (1 048 576 members x 1000 RK4 steps, MAOOAM-36, write_steps = 0) through the host-pointer API, on one model and on
a device group of 8 shards that all live on GPU 0 (`device=[0] * 8`): what the group machinery itself costs -- one host thread and
one windowed pipeline per shard, 8 H2D / D2H streams -- when the shards cannot run in parallel.  (On an 8-GPU node the same call
runs the shards side by side; this box has one GPU.)  Also the 8-shard Lyapunov estimator (one Python thread per shard)."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
import bench                                          # noqa: E402
from qgs_amd import _lib                              # noqa: E402

ndim, coo, val, jcoo, jval, _ = bench.load_model_tensors()
b, c, a = bench.rk4_tableau()
n, steps = 1048576, 1000
t = bench.grid(steps, 0.1)
ic = np.random.RandomState(5).rand(n, ndim) * 0.01
out = {}
res = {}
for tag, make in (('one_model', lambda: _lib.HipModel(ndim, coo, val, jcoo, jval, device=0)),
                  ('group_of_8_on_gpu0', lambda: _lib.HipModelGroup(ndim, coo, val, jcoo, jval, devices=[0] * 8))):
    m = make()
    m.rk_integrate(t, ic[:8192], 1, 0, b, c, a)
    res[tag] = m.rk_integrate(t, ic, 1, 0, b, c, a)                     # warm-up at full size (buffers, page-locked result block)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        res[tag] = m.rk_integrate(t, ic, 1, 0, b, c, a)
        ts.append(time.perf_counter() - t0)
    out[tag] = {'median_s': float(np.median(ts)), 'all_s': ts, 'traj_steps_per_s': n * steps / float(np.median(ts))}
    m.close()
out['bitwise_equal'] = bool(np.array_equal(res['one_model'], res['group_of_8_on_gpu0']))
out['group_over_one_model'] = out['group_of_8_on_gpu0']['median_s'] / out['one_model']['median_s']
# the same through the device-layout kernel only, for scale: 1 048 576 members in one launch
out['note'] = ('host-pointer API: H2D of 302 MB + pack + stepper + unpack + D2H of 302 MB per call; the 8 shards of the group share one GPU '
               'here, so their kernels and copies serialise -- the difference to one model is the cost of 8 threads / pipelines, not a speed-up')

# ---- Lyapunov estimator on 8 shards (Python threads), 16 384 members x 36 vectors, 50 intervals ----
from qgs_amd.functions.tendencies import tendencies_from_tensor   # noqa: E402
from qgs_amd.toolbox.lyapunov import LyapunovsEstimator           # noqa: E402
f, Df = tendencies_from_tensor(ndim, coo, val, jcoo, jval)
icl = np.random.RandomState(6).rand(16384, ndim) * 0.01
ly = {}
for tag, device in (('one_model', None), ('eight_shards_on_gpu0', [0] * 8)):
    est = LyapunovsEstimator(num_threads=1, device=device)
    est.set_func(f, Df)
    np.random.seed(3)
    est.compute_lyapunovs(0., 1.0, 2.0, 0.1, 0.01, ic=icl[:1024], write_steps=0, n_vec=36)      # warm-up
    ts = []
    for _ in range(3):
        np.random.seed(3)
        t0 = time.perf_counter()
        est.compute_lyapunovs(0., 2.0, 5.0, 0.1, 0.01, ic=icl, write_steps=0, n_vec=36)
        ts.append(time.perf_counter() - t0)
    ly[tag] = {'median_s': float(np.median(ts)), 'all_s': ts, 'intervals': 50, 'ms_per_interval': float(np.median(ts)) / 50 * 1e3}
    est.terminate()
out['lyapunov_16384x36_50_intervals'] = ly
print(json.dumps(out, indent=1))
