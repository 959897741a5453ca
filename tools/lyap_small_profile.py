"""GPU box: where the host's time goes in a Benettin run of a small ensemble (1 024 members x 36 vectors, 400 intervals, no records):
the intervals are launch-bound there.  cProfile of one run after a warm-up."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, 'tests'))
import model_configs                                                  # noqa: E402
from qgs_amd.functions.tendencies import create_tendencies           # noqa: E402
from qgs_amd.toolbox import lyapunov                                  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
f, Df = create_tendencies(model_configs.params_m36())
est = lyapunov.LyapunovsEstimator(num_threads=1)
est.set_func(f, Df)
ic = np.random.RandomState(0).rand(n, 36) * 0.01
est.compute_lyapunovs(0., 1., 2., 0.1, 0.01, ic=ic, write_steps=0)
for ws in (0, 1):
    t0 = time.perf_counter()
    est.compute_lyapunovs(0., 10., 40., 0.1, 0.01, ic=ic, write_steps=ws)
    el = time.perf_counter() - t0
    print('%d members, 400 intervals, write_steps=%d: %.3f s = %.3f ms per interval' % (n, ws, el, el / 400 * 1e3), flush=True)
for ws in (0, 1):
    pr = cProfile.Profile()
    pr.enable()
    est.compute_lyapunovs(0., 10., 40., 0.1, 0.01, ic=ic, write_steps=ws)
    pr.disable()
    print('---- write_steps = %d' % ws)
    pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
