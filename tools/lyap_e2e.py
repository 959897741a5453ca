#!/usr/bin/env python3
"""Developer script: end-to-end LyapunovsEstimator timing (single trajectory and a small ensemble)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import model_configs
from qgs_amd.functions.tendencies import create_tendencies
from qgs_amd.toolbox.lyapunov import LyapunovsEstimator
p = model_configs.params_m36(); f, Df = create_tendencies(p)
est = LyapunovsEstimator(num_threads=1); est.set_func(f, Df)
for n in ((1, 64, 1024) if len(sys.argv) < 2 else tuple(int(q) for q in sys.argv[1].split(","))):
    ic = np.random.RandomState(0).rand(n, 36) * 0.01
    for rep in range(2):
        np.random.seed(0)
        t0 = time.perf_counter()
        est.compute_lyapunovs(0., 10., 20., 0.1, 0.01, ic=ic, write_steps=10)
        el = time.perf_counter() - t0
    tt, traj, exps, vecs = est.get_lyapunovs()
    print('n=%5d  200 intervals x 10 sub-steps: %.3f s  (%.3f ms / interval)  exps shape %s  lambda_1 ~ %.4f'
          % (n, el, el / 200 * 1e3, exps.shape, float(np.mean(exps.reshape(n, 36, -1)[:, 0, :]))), flush=True)
