"""GPU box: the Benettin estimator with every interval recorded at config-4 size (16 384 members x 36 vectors) -- a record far
beyond the device budget of the windows (default 8 GB) that lands in host memory window by window.
Prints the size of the record, the windows it was cut into, the wall time and the transfer rate; checks orthonormality of a few
recorded bases.  Usage: lyap_big.py [recorded intervals, default: what fits 35 % of the host's available memory, at most 400]."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, 'tests'))
import model_configs                                                  # noqa: E402
from qgs_amd.functions.tendencies import create_tendencies           # noqa: E402
from qgs_amd.toolbox import lyapunov                                  # noqa: E402

n, nv, ndim = 16384, 36, 36
per_record = 8 * n * (ndim * nv + ndim + nv)
avail = lyapunov._host_memory_available() or (64 << 30)
intervals = int(sys.argv[1]) if len(sys.argv) > 1 else int(min(400, 0.35 * avail / per_record - 1))
if per_record * (intervals + 1) > (200 << 30) and os.environ.get('QGS_LYAP_BIG_ANYWAY') != '1':
    # 180 GB is the largest record this tool has delivered; the 0.9 TB request (5 000 intervals) took a box of the pool down while
    # its result block was being set up
    sys.exit('lyap_big.py: %d intervals = %.0f GB of records; refusing above 200 GB (QGS_LYAP_BIG_ANYWAY=1 overrides)'
             % (intervals, per_record * (intervals + 1) / 1e9))
f, Df = create_tendencies(model_configs.params_m36())
est = lyapunov.LyapunovsEstimator(num_threads=1)
est.set_func(f, Df)
ic = np.random.RandomState(0).rand(n, ndim) * 0.01
np.random.seed(0)
est.compute_lyapunovs(0., 1., 1.5, 0.1, 0.01, ic=ic[:256], write_steps=1)          # warm-up (kernels, pools)
out = {'members': n, 'vectors': nv, 'recorded_intervals': intervals, 'record_gb': per_record * (intervals + 1) / 1e9,
       'host_available_gb': avail / 1e9, 'device_window_budget_mb': lyapunov._window_budget_bytes() / 1048576.0}
if os.environ.get('QGS_LYAP_BIG_NO_THP') == '1':               # experiment: result blocks without MADV_HUGEPAGE
    from qgs_amd import _lib as _l0
    _l0._advise_huge_pages = lambda a: None
np.random.seed(1)
# host-side time line of the run: when each member group was handed to `_compute_shard_on_current_device`, how long the final wait
# for the drain thread took (QGS_LYAP_BIG_TIMELINE=1)
marks = []
if os.environ.get('QGS_LYAP_BIG_TIMELINE') == '1':
    inner = est._compute_shard_on_current_device

    def timed(m, ic_, *a, **k):
        marks.append(('group of %d enqueue starts' % ic_.shape[0], time.perf_counter()))
        r = inner(m, ic_, *a, **k)
        marks.append(('group enqueued', time.perf_counter()))
        return r
    est._compute_shard_on_current_device = timed

    def wrap(obj, name, label):
        fn = getattr(obj, name)

        def w(*a, **k):
            marks.append((label + ' starts', time.perf_counter()))
            r = fn(*a, **k)
            marks.append((label + ' done', time.perf_counter()))
            return r
        setattr(obj, name, w)
    from qgs_amd import _lib as _l
    wrap(lyapunov, '_host_memory_available', 'host memory check')
    wrap(_l._RESULTS, 'empty', 'result block')
    wrap(lyapunov._fn, 'hip_model_of', 'model lookup')
    wrap(est, '_member_groups', 'group rule')
t0 = time.perf_counter()
est.compute_lyapunovs(0., 2., 2. + 0.1 * intervals, 0.1, 0.01, ic=ic, write_steps=1, n_vec=nv)
el = time.perf_counter() - t0
tt, traj, exps, vecs = est.get_lyapunovs()
out.update({'seconds': el, 'windows_base_records': est.last_windows, 'gb_per_s': out['record_gb'] / el,
            'ms_per_interval': el / (intervals + 20) * 1e3, 'vectors_shape': list(vecs.shape)})
dev = 0.0
for i in (0, n // 2, n - 1):
    for r in (0, intervals // 2, intervals):
        q = vecs[i, :, :, r]
        dev = max(dev, float(np.abs(q.T @ q - np.eye(nv)).max()))
if marks:
    out['timeline_s'] = [[name, round(t - t0, 3)] for name, t in marks] + [['returned', round(el, 3)]]
out['max_orthonormality_defect'] = dev
out['lambda_1_mean'] = float(np.mean(exps[:, 0, :]))
print(json.dumps(out, indent=1))
