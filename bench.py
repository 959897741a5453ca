#!/usr/bin/env python3
"""bench.py -- headline benchmark: ensemble trajectory-steps/s, fp64, MAOOAM-36 (BASELINE.json config 2).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One bench "step" = one pass of the hot path over one batch: every rank takes its own batch of
`--members` (default 65 536) synthetic initial conditions that are already resident in HBM in the
reference's (n_traj, ndim) layout, packs them mode-major, integrates `--rk-steps` (default 1000) classic
RK4 steps of MAOOAM 2x2/2x4 (36 variables, the qgs_maooam.py parameter set) with write_steps=0 in ONE
fused HIP kernel, unpacks the final states to (n_traj, ndim) and (N>1) gathers them onto rank 0 with RCCL (asynchronously,
overlapping the next pass).
Members are independent, so ranks shard them with no data-path collective except that final gather
("scaling": "weak": per-GPU work is fixed).

value = (members * rk_steps * N * K) / (max over ranks of the timed region)   [trajectory-steps / s]

Extra objects on the JSON line: `roofline` (algorithmic HBM bytes of the stepper kernel / its measured
duration, against 8 TB/s) and `cpu_baseline` (the C oracle = scalar restatement of the reference's
numba loops, OpenMP over members, timed on this host on a bounded sample; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)
FP64_VALU_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz


def rk4_tableau():
    c = np.array([0., 0.5, 0.5, 1.])
    b = np.array([1. / 6, 1. / 3, 1. / 3, 1. / 6])
    a = np.zeros((4, 4))
    a[1, 0] = 0.5
    a[2, 1] = 0.5
    a[3, 2] = 1.
    return b, c, a


def load_model_tensors():
    """MAOOAM-36 tensors of the qgs_maooam.py parameter set (BASELINE config 2)."""
    try:
        from qgs_amd.params.params import QgParams
        from qgs_amd.functions.tendencies import create_tendencies
        p = QgParams()
        p.set_atmospheric_channel_fourier_modes(2, 2)
        p.set_oceanic_basin_fourier_modes(2, 4)
        p.set_params({'kd': 0.0290, 'kdp': 0.0290, 'n': 1.5, 'r': 1.e-7, 'h': 136.5, 'd': 1.1e-7})
        p.atemperature_params.set_params({'eps': 0.7, 'T0': 289.3, 'hlambda': 15.06, })
        p.gotemperature_params.set_params({'gamma': 5.6e8, 'T0': 301.46})
        p.atemperature_params.set_insolation(103.3333, 0)
        p.gotemperature_params.set_insolation(310., 0)
        f, Df = create_tendencies(p)
        return p.ndim, f.coo, f.val, Df.coo, Df.val, 'qgs_amd.create_tendencies(QgParams: qgs_maooam.py set)'
    except ImportError:
        g = np.load(os.path.join(HERE, 'tests', 'golden', 'm36.npz'))
        return int(g['ndim']), g['coo'], g['val'], g['jcoo'], g['jval'], 'tests/golden/m36.npz'


def cpu_baseline(ndim, coo, val, rk_steps, dt, target_seconds=12.0):
    """Time the CPU oracle (oracle/qgs_oracle.c) on a bounded sample of the same workload."""
    from oracle.oracle import OracleModel, max_threads
    m = OracleModel(ndim, coo, val)
    b, c, a = rk4_tableau()
    threads = max_threads()
    time_grid = np.concatenate((np.arange(0., rk_steps * dt, dt), np.full((1,), rk_steps * dt)))
    rng = np.random.RandomState(21217)
    n_cal = max(threads * 4, 64)
    ic = rng.rand(n_cal, ndim) * 0.01
    m.integrate_runge_kutta_jit(time_grid[:11], ic, 1, 0, b, c, a, threads=threads)          # warm up the pool
    t0 = time.perf_counter()
    m.integrate_runge_kutta_jit(time_grid[:101], ic, 1, 0, b, c, a, threads=threads)
    rate = n_cal * 100 / (time.perf_counter() - t0)
    n_traj = int(max(threads, min(65536, rate * target_seconds / rk_steps)))
    n_traj = max(threads, (n_traj // threads) * threads)
    ic = rng.rand(n_traj, ndim) * 0.01
    t0 = time.perf_counter()
    out = m.integrate_runge_kutta_jit(time_grid, ic, 1, 0, b, c, a, threads=threads)
    el = time.perf_counter() - t0
    model = ''
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    return {'value': n_traj * rk_steps / el, 'unit': 'traj-steps/s', 'cores': threads, 'kind': 'port',
            'sample': '%d members x %d RK4 steps of the same MAOOAM-36 workload, %.1f s wall, OpenMP over members'
                      % (n_traj, rk_steps, el),
            'cpu': model, 'per_core': n_traj * rk_steps / el / threads}, ic, out[:, :, 0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--members', type=int, default=65536, help='ensemble members per GPU')
    ap.add_argument('--rk-steps', type=int, default=1000, help='RK4 steps per pass')
    ap.add_argument('--kernel', choices=['auto', 'generic', 'spec'], default='auto')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--force-dist', action='store_true', help='initialise the RCCL process group even at world size 1 (plumbing check)')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if rank == 0:
            print('bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)' % (args.gpus, world),
                  file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        print('bench.py: no GPU visible; the HIP path has no CPU fallback', file=sys.stderr)
        sys.exit(1)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from qgs_amd import _lib

    ndim, coo, val, jcoo, jval, tensor_src = load_model_tensors()
    model = _lib.HipModel(ndim, coo, val, jcoo, jval, device=local_rank)
    model.set_kernel({'auto': 0, 'generic': 1, 'spec': 2}[args.kernel])

    n_traj, rk_steps, dt = args.members, args.rk_steps, 0.1
    ld = (n_traj + 63) // 64 * 64
    b, c, a = rk4_tableau()
    time_grid = np.concatenate((np.arange(0., rk_steps * dt, dt), np.full((1,), rk_steps * dt)))

    # synthetic initial conditions (the distribution qgs_maooam.py:108 uses), different per rank
    rng = np.random.RandomState(21217 + rank)
    ic_host = rng.rand(n_traj, ndim) * 0.01
    d_ic_rows = torch.from_numpy(ic_host).to(dev)                      # resident in HBM, reference layout
    d_ic_modes = torch.empty((ndim, ld), dtype=torch.float64, device=dev)
    d_rec = torch.empty((1, ndim, ld), dtype=torch.float64, device=dev)
    d_out = [torch.empty((n_traj, ndim), dtype=torch.float64, device=dev) for _ in range(2)]   # double buffer
    from qgs_amd.parallel import ShardedEnsemble, RootGather
    ens = ShardedEnsemble(world * n_traj)                              # contiguous member blocks, one per rank
    assert ens.n_local == n_traj
    root = RootGather(ens, dst=0)
    d_all = torch.empty((world * n_traj, ndim), dtype=torch.float64, device=dev) if (world > 1 and rank == 0) else None
    pending = [None, None]                                             # in-flight gathers of d_out[0], d_out[1]
    stream = torch.cuda.current_stream().cuda_stream

    kern_events = []
    gather_events = []

    def one_pass(record_events, k=0):
        d_out_rows = d_out[k % 2]
        if pending[k % 2] is not None:                                 # the gather that still reads this buffer
            pending[k % 2].wait()
            pending[k % 2] = None
        model.pack_states(n_traj, ld, d_ic_rows.data_ptr(), d_ic_modes.data_ptr(), stream)
        if record_events:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        model.rk_integrate_device(n_traj, ld, d_ic_modes.data_ptr(), time_grid, 1, 0, b, c, a, d_rec.data_ptr(), stream)
        if record_events:
            e1.record()
            kern_events.append((e0, e1))
        model.unpack_records(n_traj, ld, ndim, 1, d_rec.data_ptr(), d_out_rows.data_ptr(), stream)
        if use_dist:
            # RCCL gather of the final states onto rank 0 over xGMI: the only collective.  It is asynchronous
            # (RCCL's own stream), so it overlaps the next pass; the timed region ends after the last one completed.
            if record_events:
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record()
            if world > 1:
                pending[k % 2], _ = root.start(d_out_rows, out=d_all, async_op=True)
            else:
                dist.all_gather_into_tensor(torch.empty_like(d_out_rows), d_out_rows)
            if record_events:
                g1.record()
                gather_events.append((g0, g1))

    def drain():
        for i in (0, 1):
            if pending[i] is not None:
                pending[i].wait()
                pending[i] = None

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        one_pass(False, k)
    drain()
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_pass(True, k)
    drain()
    barrier()
    elapsed = time.perf_counter() - t0

    el_t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(el_t, op=dist.ReduceOp.MAX)
    elapsed = float(el_t.item())

    kern_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in kern_events])) if kern_events else float('nan')
    kinfo = model.last_kernel_info()
    gather_ms = float(np.mean([g0.elapsed_time(g1) for g0, g1 in gather_events])) if gather_events else None

    # correctness guard inside the bench (rank 0): a handful of members against the CPU oracle
    result = None
    if rank == 0:
        total_traj_steps = float(n_traj) * rk_steps * world * args.steps
        value = total_traj_steps / elapsed
        bytes_per_traj_step = 2 * 8 * ndim                                 # state read once + written once per step
        flops_per_traj_step = 12 * len(val) + 14 * ndim
        alg_bytes = float(bytes_per_traj_step) * n_traj * rk_steps
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(HERE, 'profiles', 'hbm_traffic.json')
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    traffic = json.load(f).get(kinfo['name'], {}).get('hbm_bytes_per_launch')
            except (OSError, ValueError):
                traffic = None
        result = {
            'metric': 'ensemble trajectory-steps/sec fp64, MAOOAM-36',
            'value': value, 'unit': 'traj-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'MAOOAM 2x2 atm / 2x4 ocean (36 modes) fp64, %d-member ensemble per GPU, %d RK4 steps '
                                   'per pass, write_steps=0 (BASELINE configs[1])' % (n_traj, rk_steps),
                       'members_per_gpu': n_traj, 'rk_steps_per_pass': rk_steps, 'ndim': ndim, 'tensor_nnz': int(len(val)),
                       'dt': dt, 'tensor_source': tensor_src, 'parallelism': 'members sharded x%d, RCCL gather of final states onto rank 0 (async, overlapped)' % world,
                       'kernel': kinfo},
            'mode_updates_per_s': value * ndim,
            'gather_ms_per_step': gather_ms,        # host-side enqueue-to-enqueue time of the async RCCL gather (rank 0), None at N=1
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'kernel': kinfo['name'], 'kernel_ms': kern_ms,
                         'algorithmic_bytes_per_launch': alg_bytes,
                         'fp64_tflops_algorithmic': flops_per_traj_step * n_traj * rk_steps / (kern_ms * 1e-3) / 1e12,
                         'fp64_valu_peak_tflops': FP64_VALU_PEAK_TFLOPS},
        }
        if world == 1 and not args.no_cpu_baseline:
            base, ic_s, ref_final = cpu_baseline(ndim, coo, val, rk_steps, dt)
            result['cpu_baseline'] = base
            ns = min(64, ic_s.shape[0])
            got = model.rk_integrate(time_grid, ic_s[:ns], 1, 0, b, c, a)[:, :, 0]
            err = float(np.abs(got - ref_final[:ns]).max() / np.abs(ref_final[:ns]).max())
            result['parity_check'] = {'members': ns, 'rk_steps': rk_steps, 'max_rel_err_vs_oracle': err, 'tolerance': 1e-10}
            if not err < 1e-10:
                print('bench.py: PARITY FAILURE vs oracle: %g' % err, file=sys.stderr)
                result['value'] = 0.0
        print(json.dumps(result))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
