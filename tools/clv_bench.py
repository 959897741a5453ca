"""Where the time of a covariant-Lyapunov-vector estimation goes (MAOOAM-36, 10 + 20 + 20 intervals, 36 vectors, 21 records), for
growing ensembles.  Method 0: forward part / R matrices / backward recursion, device-resident (default) or with the records on
the host and the recursion in NumPy (CLV_BENCH_HOST=1).  CLV_BENCH_METHOD=1: the subspace-intersection method (two Benettin
runs on the GPU, batched LAPACK SVDs on the host).      python tools/clv_bench.py [members ...]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))


def main():
    from conftest import load_golden
    from qgs_amd.functions.tendencies import tendencies_from_tensor
    from qgs_amd.toolbox.lyapunov import CovariantLyapunovsEstimator
    g = load_golden('m36')
    f, Df = tendencies_from_tensor(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    est = CovariantLyapunovsEstimator()           # (num_threads = the host's cores: the SVDs of method 1)
    est.set_func(f, Df)
    out = []
    method = int(os.environ.get('CLV_BENCH_METHOD', '0'))
    host = os.environ.get('CLV_BENCH_HOST') == '1'          # the host recursion instead of the device-resident path
    est.device_resident = False if host else None
    for n in [int(a) for a in sys.argv[1:]] or [1, 64, 1024]:
        ic = np.random.RandomState(1).rand(n, g.ndim) * 0.01
        np.random.seed(2)
        for rep in range(2):                       # the second pass has the kernels loaded
            t = time.perf_counter()
            est.compute_clvs(0., 1., 3., 5., 0.1, 0.1, ic=ic, write_steps=1, method=method)
            wall = time.perf_counter() - t
        if method == 0:
            row = dict(method=0, members=n, intervals=50, n_vec=g.ndim, records=21, path=est.last_path,
                       **dict({k: round(v, 3) for k, v in est.last_timing.items()}, wall_s=round(wall, 3)))
        else:
            row = dict(method=1, members=n, intervals=50, n_vec=g.ndim, records=21, wall_s=round(wall, 3))
        print(json.dumps(row), flush=True)
        out.append(row)
    est.terminate()
    return out


if __name__ == '__main__':
    main()
