#!/usr/bin/env python3
"""bench.py -- headline benchmark: ensemble trajectory-steps/s, fp64, MAOOAM-36 (BASELINE.json config 2).

    python bench.py [--gpus N] [--steps K] [--warmup W]

works as typed for any N: with N > 1 and no launcher environment the process (which never touches a GPU)
starts N child ranks (one per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) and relays rank 0's JSON line.
It also runs as a rank under
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One bench "step" = one pass of the hot path over one batch: every rank takes its own batch of `--members`
(default 65 536) synthetic initial conditions that are already resident in HBM in the reference's (n_traj, ndim)
layout, packs them mode-major, integrates `--rk-steps` (default 1000) classic RK4 steps of MAOOAM 2x2/2x4
(36 variables, the qgs_maooam.py parameter set) with write_steps=0 in ONE fused HIP kernel, unpacks the final
states to (n_traj, ndim) and (N>1) gathers them onto rank 0 with RCCL (asynchronously, overlapping the next pass).
Members are independent, so ranks shard them with no data-path collective except that final gather
("scaling": "weak": per-GPU work is fixed).

value = (members * rk_steps * N * K) / (max over ranks of the timed region)   [trajectory-steps / s]

Extra objects on the JSON line (rank 0):
  roofline      the stepper kernel against the bound that binds it, the FP64 vector rate (the state stays in
                registers for the whole launch, so HBM sees 0.1 % of the algorithmic bytes); the SURVEY 8(d) byte
                figure and the counter traffic (a committed PMC measurement: `traffic_source`) are kept next to it
  parity_check  the output buffer of the LAST TIMED PASS (the kernel named in `roofline.kernel`), sampled members
                against the CPU oracle on the same initial conditions; a failure sets `value` to 0
  configs       (N = 1) the other single-GPU BASELINE configurations, timed in this same run with HIP events:
                config 2 with write_steps=1, config 2 through the host-pointer API (H2D + D2H included),
                config 3 (MAOOAM 6x6, ndim 228), config 4 (tangent model, 100 calls; batched QR separately),
                config 5 on one GPU -- each with its own `parity_check` of what was timed
  cold_start    (N = 1) seconds from `create_tendencies` to the first 1000-step result in a fresh process, on an empty
                kernel cache and on a structure-warm one (other parameter values, same tensor structure)
  cpu_baseline  (N = 1) the C restatement of the reference's numba loops (oracle/qgs_oracle.c) built here with
                -O3 -march=native (FMA allowed), timed on 1 thread and on all physical cores
  single_gpu_reference  (N > 1) the same members-per-GPU passes on every GPU at once WITHOUT the gather: N x this is
                the denominator of the scaling efficiency of this line
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
# (No OMP_PLACES / OMP_PROC_BIND here: set before the first OpenMP runtime starts -- PyTorch brings one -- they bind the main
# thread to one core, the process then sees 2 CPUs and the CPU-baseline leg runs on 2 threads (measured, round 3); set later
# they are never read.  The baseline's threads are left to the scheduler; its thread count follows the container's CPU quota.)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)
FP64_VALU_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
PEAK_CLOCK_GHZ = 2.4           # the clock that peak is quoted at


def clock_fields(model, achieved_tflops):
    """`effective_clock_ghz` of the kernel just timed and the FP64 fraction against the peak AT THAT CLOCK.  The clock comes from
    the kernel itself (qgs_kernel_clock: lane 0 of workgroup 0 reads the shader-clock counter and the constant 100 MHz counter
    at entry and after its last store), so it is the clock the timed launch ran at -- no profiler pass, no smi sampling.  Boxes
    of this pool differ in what they sustain under the fp64 load (2.07-2.4 GHz seen): `frac` moves with it, this figure does not."""
    try:
        clk = model.kernel_clock()
    except Exception:
        clk = None
    if not clk or not clk[0] > 0:
        return {'effective_clock_ghz': None, 'frac_at_effective_clock': None}
    ghz, ms = clk
    out = {'effective_clock_ghz': ghz, 'clock_probe_ms': ms, 'clock_source': 'in-kernel s_memtime / s_memrealtime of workgroup 0 of the last timed launch'}
    if achieved_tflops is not None:
        out['frac_at_effective_clock'] = achieved_tflops / (FP64_VALU_PEAK_TFLOPS * ghz / PEAK_CLOCK_GHZ)
    return out


def rk4_tableau():
    c = np.array([0., 0.5, 0.5, 1.])
    b = np.array([1. / 6, 1. / 3, 1. / 3, 1. / 6])
    a = np.zeros((4, 4))
    a[1, 0] = 0.5
    a[2, 1] = 0.5
    a[3, 2] = 1.
    return b, c, a


def grid(steps, dt):
    return np.concatenate((np.arange(0., steps * dt, dt), np.full((1,), steps * dt)))[:steps + 1]


def load_model_tensors(kd=0.0290, kdp=0.0290):
    """MAOOAM-36 tensors of the qgs_maooam.py parameter set (BASELINE config 2); other kd / kdp: the cold-start probe."""
    from qgs_amd.params.params import QgParams
    from qgs_amd.functions.tendencies import create_tendencies
    p = QgParams()
    p.set_atmospheric_channel_fourier_modes(2, 2)
    p.set_oceanic_basin_fourier_modes(2, 4)
    p.set_params({'kd': kd, 'kdp': kdp, 'n': 1.5, 'r': 1.e-7, 'h': 136.5, 'd': 1.1e-7})
    p.atemperature_params.set_params({'eps': 0.7, 'T0': 289.3, 'hlambda': 15.06, })
    p.gotemperature_params.set_params({'gamma': 5.6e8, 'T0': 301.46})
    p.atemperature_params.set_insolation(103.3333, 0)
    p.gotemperature_params.set_insolation(310., 0)
    f, Df = create_tendencies(p)
    return p.ndim, f.coo, f.val, Df.coo, Df.val, 'qgs_amd.create_tendencies(QgParams: qgs_maooam.py set)'


# ---------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` typed directly
# ---------------------------------------------------------------------------------------------------------------
KFD_NODES = '/sys/class/kfd/kfd/topology/nodes'


def masked_count(n, env=None):
    """`n` physical GPUs cut down by the visibility masks a HIP process honours: ROCR_VISIBLE_DEVICES (applied by the ROCr runtime:
    indices or `GPU-<uuid>` names), then HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (applied by HIP to what ROCr left).  An entry
    that is not a valid index ends the list, as in the runtimes (`0,1,-1,2` = two devices); an empty value hides every GPU."""
    env = os.environ if env is None else env
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = env.get(var)
        if v is None:
            continue
        kept, seen = 0, set()
        for q in v.split(','):
            q = q.strip()
            if q.startswith('GPU-') and len(q) > 4:
                key = q
            else:
                try:
                    key = int(q)
                except ValueError:
                    break
                if key < 0 or key >= n:
                    break
            if key in seen:
                break
            seen.add(key)
            kept += 1
        n = min(n, kept)
    return n


def visible_gpus(kfd_nodes=None, env=None):
    """Number of GPUs the child ranks will see, counted WITHOUT loading the HIP runtime (the launcher parent must stay
    GPU-free: it only starts child processes): the KFD topology nodes that have SIMDs (CPUs are nodes with simd_count 0),
    cut down by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when set."""
    import glob
    n, seen = 0, 0
    for path in glob.glob(os.path.join(kfd_nodes or KFD_NODES, '*', 'properties')):
        try:
            with open(path) as f:
                for line in f:
                    k, _, v = line.partition(' ')
                    if k == 'simd_count':
                        seen += 1
                        n += 1 if int(v) > 0 else 0
                        break
        except (OSError, ValueError):
            pass
    if seen == 0:
        # no readable KFD topology (unusual container): ask a short-lived child, so that this process still never loads HIP
        # (the child applies the masks itself)
        try:
            out = subprocess.run([sys.executable, '-c', 'import torch; print(torch.cuda.device_count())'],
                                 stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300).stdout
            return int(out.decode().strip().splitlines()[-1])
        except (OSError, ValueError, IndexError, subprocess.TimeoutExpired):
            return 0
    return masked_count(n, env)


def launch_ranks(n, argv, program=None, grace=2.0):
    """Start `n` ranks of `program` (default: this file) on 127.0.0.1, relay rank 0's standard output, return the exit code.  A rank
    that dies -- before or after the process group is up -- ends the job: its peers get `grace` seconds to report on their own,
    then exactly the processes started here are killed and the parent exits non-zero (never a hang in a collective)."""
    import socket
    m = visible_gpus()
    if m < n:
        print('bench.py: %d GPUs requested, %d visible' % (n, m), file=sys.stderr)
        return 3
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        # dmabuf IPC: the host driver of this pool supports no other kind, and RCCL's intra-node transport (and any sharing of
        # device tensors across processes) fails with `hipIpcGetMemHandle: invalid argument` under the legacy mode.  The pool
        # exports the variable already; it is set here only when the operator's environment does not say otherwise.
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, program or os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # a rank that dies would leave the others waiting in a collective: watch all of them, stop the rest (exactly the
    # processes started above) as soon as one fails
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            break
        time.sleep(0.2)
    if failed:
        time.sleep(grace)                                 # let the failing rank's peers report on their own first
        for p in procs:
            if p.poll() is None:
                p.kill()
    codes = [p.wait() for p in procs]
    reader.join(timeout=5.0)
    sys.stdout.write(b''.join(chunks).decode())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c]
    if bad:
        print('bench.py: ranks failed (rank, exit code): %r' % bad, file=sys.stderr)
        return 1
    return 0


def rccl_report(torch, dist, dev, world, rank):
    """What the communicator says about itself, for the line of an N > 1 run (and of --force-dist): RCCL's version, the number of
    ranks the process group reports, how many ranks actually answer a collective, and which GPU every rank sits on -- so that a
    SCALE record can answer "did RCCL see N ranks on N different GPUs" without a rerun.  Collectives: one all_reduce, one all_gather
    of 16 numbers; outside every timed region."""
    out = {'backend': dist.get_backend(), 'world_size_reported': dist.get_world_size(), 'world_size_requested': world}
    try:
        v = torch.cuda.nccl.version()
        out['rccl_version'] = '.'.join(str(q) for q in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception as e:
        out['rccl_version'] = 'unavailable (%r)' % (e,)
    ones = torch.ones(1, dtype=torch.float64, device=dev)
    dist.all_reduce(ones)
    out['ranks_answering_all_reduce'] = int(round(float(ones.item())))
    props = torch.cuda.get_device_properties(dev)
    bus = getattr(props, 'pci_bus_id', -1)
    mine = torch.tensor([rank, dev.index if dev.index is not None else -1, int(bus) if isinstance(bus, int) else -1,
                         int(getattr(props, 'pci_device_id', -1)), int(getattr(props, 'pci_domain_id', -1)),
                         int(props.multi_processor_count), int(props.total_memory >> 30), os.getpid()] + [0] * 8,
                        dtype=torch.int64, device=dev)
    table = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(table, mine)
    out['ranks'] = [{'rank': int(t[0]), 'device_index': int(t[1]), 'pci_bus_id': int(t[2]), 'pci_device_id': int(t[3]),
                     'pci_domain_id': int(t[4]), 'compute_units': int(t[5]), 'memory_gib': int(t[6]), 'pid': int(t[7])} for t in table]
    out['distinct_gpus'] = len({(r['pci_domain_id'], r['pci_bus_id'], r['pci_device_id'], r['device_index']) for r in out['ranks']})
    out['device_name'] = props.name
    out['env'] = {k: os.environ.get(k) for k in ('HSA_ENABLE_IPC_MODE_LEGACY', 'NCCL_DEBUG', 'ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES',
                                                 'CUDA_VISIBLE_DEVICES', 'MASTER_ADDR', 'NCCL_SOCKET_IFNAME')}
    if rank == 0:
        print('bench.py: RCCL %s, %d ranks in the process group, %d answered, %d distinct GPUs' %
              (out['rccl_version'], out['world_size_reported'], out['ranks_answering_all_reduce'], out['distinct_gpus']), file=sys.stderr)
    return out


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1)
# ---------------------------------------------------------------------------------------------------------------
def cpu_info():
    model, cores = '', set()
    try:
        phys = core = None
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name') and not model:
                    model = line.split(':', 1)[1].strip()
                elif line.startswith('physical id'):
                    phys = line.split(':', 1)[1].strip()
                elif line.startswith('core id'):
                    core = line.split(':', 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    try:
        logical = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    physical = min(len(cores), logical) if cores else logical
    return model, max(1, physical), logical


def cgroup_cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited."""
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            q, p = f.read().split()[:2]
        if q != 'max':
            return float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:
            q = float(f.read())
        with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
            p = float(f.read())
        if q > 0:
            return q / p
    except (OSError, ValueError):
        pass
    return None


def cpu_baseline(ndim, coo, val, rk_steps, dt, seconds_1=5.0, seconds_all=12.0):
    """Time the C restatement of the reference's loops on this host: same COO loop order, built HERE with
    -O3 -march=native (FMA allowed) -- a performance build next to the -O2 -ffp-contract=off parity build."""
    try:
        import numba  # noqa: F401
        have_numba = True
    except Exception:
        have_numba = False
    model_name, physical, logical = cpu_info()
    quota = cgroup_cpu_quota()
    physical_all = physical
    if quota is not None:                                 # more threads than the container's CPU quota only fight each other
        physical = max(1, min(physical, int(quota)))
    from oracle.oracle import OracleModel
    flavour = 'fast'
    try:
        m = OracleModel(ndim, coo, val, flavour='fast')
    except Exception as e:                                # no compiler on this host: fall back to the parity build
        print('bench.py: -O3 oracle build failed (%s); timing the parity build' % e, file=sys.stderr)
        m = OracleModel(ndim, coo, val)
        flavour = 'parity'
    parity = OracleModel(ndim, coo, val)
    b, c, a = rk4_tableau()
    tg = grid(rk_steps, dt)
    rng = np.random.RandomState(21217)

    def timed(n_traj, threads):
        ic = rng.rand(n_traj, ndim) * 0.01
        t0 = time.perf_counter()
        out = m.integrate_runge_kutta_jit(tg, ic, 1, 0, b, c, a, threads=threads)
        return time.perf_counter() - t0, ic, out

    m.integrate_runge_kutta_jit(tg[:11], rng.rand(physical * 2, ndim) * 0.01, 1, 0, b, c, a, threads=physical)   # pool warm-up
    el, _, _ = timed(8, 1)
    rate1 = 8 * rk_steps / el
    n1 = int(max(8, min(4096, rate1 * seconds_1 / rk_steps)))
    el1, _, _ = timed(n1, 1)
    rate1 = n1 * rk_steps / el1
    n_all = int(max(physical, min(65536, rate1 * physical * seconds_all / rk_steps)))
    n_all = max(physical, n_all // physical * physical)
    el_all, ic, out = timed(n_all, physical)
    rate_all = n_all * rk_steps / el_all
    # thread scaling in between (bounded samples), so that the all-cores figure can be judged
    scaling = {1: rate1}
    for th in (4, 16, 64):
        if th < physical:
            n_th = max(th, int(rate1 * th * 2.0 / rk_steps) // th * th)
            el_th, _, _ = timed(n_th, th)
            scaling[th] = n_th * rk_steps / el_th
    scaling[physical] = rate_all
    # the performance build must still be the same algorithm: check it against the parity build
    ns = min(16, n_all)
    ref = parity.integrate_runge_kutta_jit(tg, ic[:ns], 1, 0, b, c, a, threads=min(ns, physical))
    dev = float(np.abs(out[:ns] - ref).max() / np.abs(ref).max())
    return {'value': rate_all, 'unit': 'traj-steps/s', 'cores': physical, 'kind': 'port',
            'sample': '%d members x %d RK4 steps of the same MAOOAM-36 workload on %d threads (one per physical core, capped by the container CPU quota), %.1f s wall; '
                      '1 thread: %d members, %.1f s' % (n_all, rk_steps, physical, el_all, n1, el1),
            'cpu': model_name, 'physical_cores': physical_all, 'logical_cpus': logical, 'cgroup_cpu_quota': quota,
            'one_thread': rate1, 'per_core': rate_all / physical, 'rate_by_threads': {str(k): v for k, v in scaling.items()},
            'build': 'oracle/qgs_oracle.c, gcc -O3 -march=native -fopenmp (FMA contraction allowed)' if flavour == 'fast'
                     else 'oracle/qgs_oracle.c, gcc -O2 -ffp-contract=off (parity build)',
            'fast_vs_parity_build_rel_diff': dev, 'numba_importable': have_numba,
            'published_numba_single_core': 1.5e5}, ic[:ns], ref[:, :, 0]


# ---------------------------------------------------------------------------------------------------------------
# the other single-GPU BASELINE configurations (rank 0, N = 1)
# ---------------------------------------------------------------------------------------------------------------
def event_ms(torch, fn, n, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), ts


TRAFFIC_SOURCE = ('profiles/hbm_traffic.json (rocprofv3 PMC passes, tools/r06_profiles.sh, one entry per kernel / grid / duration class as in '
                  'profiles/r06_pmc_by_class.csv; a committed constant looked up by (kernel, threads of the launch, duration), not measured in this run)')


def _traffic_table():
    try:
        with open(os.path.join(HERE, 'profiles', 'hbm_traffic.json')) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def traffic_entry(kernel, threads, ms=None, table=None):
    """The committed PMC entry of `kernel` launched with `threads` work-items (rocprofv3's Grid_Size), or None.  Entries are keyed
    `<kernel>@grid<threads>[/class<k>]`: one kernel name at two grid sizes (the headline stepper at 65 536 and at 1 048 576 members),
    or at one grid size with launches of very different length (100 / 1000 steps), are DIFFERENT entries.  With several duration
    classes the one closest to `ms` (within a factor 2) is taken; a (kernel, grid) the table does not hold gives None -- never
    another grid's figure."""
    table = _traffic_table() if table is None else table
    prefix = '%s@grid%d' % (kernel, int(threads))
    cands = [e for k, e in table.items() if k == prefix or k.startswith(prefix + '/class')]
    if not cands:
        return None
    if ms is None or len(cands) == 1 and not cands[0].get('mean_ms'):
        return cands[0] if len(cands) == 1 else None
    best = min(cands, key=lambda e: abs(np.log(max(e.get('mean_ms', 0.0), 1e-9) / ms)))
    return best if 0.5 <= best.get('mean_ms', 0.0) / ms <= 2.0 else None


def measured_traffic(kernel, threads, ms=None):
    """HBM bytes per launch (FETCH_SIZE / WRITE_SIZE counters, corrected as guides/MI355X_MICROARCH.md prescribes) or None."""
    e = traffic_entry(kernel, threads, ms)
    return e.get('hbm_bytes_per_launch') if e else None


def launch_threads(kernel, members, columns=1):
    """Work-items of one launch of a kernel of this library, as rocprofv3 reports them (Grid_Size): what keys the PMC tables."""
    wg = (int(members) + 63) // 64
    if kernel.startswith('qgs_spec_rkldsa') or kernel.startswith('qgs_spec_rklds') or kernel.startswith('qgs_spec_tendlds'):
        digits = ''.join(ch for ch in kernel.split('lds')[-1] if ch.isdigit())
        return wg * 64 * int(digits or 16)
    if kernel.startswith('qgs_spec_qr_'):
        return (int(members) + 15) // 16 * 256
    if kernel.startswith('qgs_spec_tgl'):
        return wg * 64 * int(columns)
    return wg * 64


def executed_view(kernel, threads, ms, steps, waves_per_simd_note=None):
    """What the kernel EXECUTES, from the committed PMC pass of this (kernel, grid, duration) class: VALU instructions per wavefront
    and step, and the share of the issue slots they fill -- SQ_INSTS_VALU x 4 cycles (a wave64 fp64 / 32-bit VALU instruction
    occupies its SIMD for 4 cycles) over SQ_BUSY_CYCLES-free arithmetic: waves x steps x cycles per step of the profiled launch at its
    measured clock (GRBM_GUI_ACTIVE).  A committed constant of the profiled launch, not of this run."""
    try:
        with open(os.path.join(HERE, 'profiles', 'pmc_by_class.json')) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None
    e = traffic_entry(kernel, threads, ms, table)
    if not e or not e.get('SQ_INSTS_VALU') or not e.get('SQ_WAVES') or not e.get('grbm_clock_ghz'):
        return None
    valu_per_wave_step = e['SQ_INSTS_VALU'] / e['SQ_WAVES'] / steps
    cycles = e['mean_ms'] * 1e-3 * e['grbm_clock_ghz'] * 1e9                # shader cycles of the profiled launch
    simds = 1024.0
    occupancy = e['SQ_INSTS_VALU'] * 4.0 / (cycles * simds)
    return {'valu_instr_per_wave_step': valu_per_wave_step, 'valu_issue_occupancy': occupancy,
            'wait_any_frac_of_wave_cycles': (e['SQ_WAIT_ANY'] / e['SQ_WAVE_CYCLES']) if e.get('SQ_WAVE_CYCLES') else None,
            'profiled_ms': e['mean_ms'], 'profiled_clock_ghz': e['grbm_clock_ghz'],
            'source': 'profiles/pmc_by_class.json (= profiles/r06_pmc_by_class.csv): SQ_INSTS_VALU x 4 cycles / (launch cycles at the GRBM clock x 1024 SIMDs)'}


# ---------------------------------------------------------------------------------------------------------------
# parity of what was timed (rank 0): sampled members against the CPU oracle
# ---------------------------------------------------------------------------------------------------------------
SAMPLE_MEMBERS = (0, 63, 64, 4097, 32768, 65535)


def sample_members(n, count=16):
    """Member indices to check: fixed ones at wavefront / workgroup / XCD boundaries plus seeded random ones."""
    idx = [q for q in SAMPLE_MEMBERS if q < n]
    rng = np.random.RandomState(4)
    while len(idx) < min(count, n):
        q = int(rng.randint(0, n))
        if q not in idx:
            idx.append(q)
    return np.array(sorted(idx[:count]))


def rel_err(got, ref):
    return float(np.abs(np.asarray(got) - np.asarray(ref)).max() / max(float(np.abs(ref).max()), 1e-300))


def parity_entry(kernel, what, err, tol, members):
    return {'kernel': kernel, 'checked': what, 'members': [int(q) for q in members], 'max_rel_err_vs_oracle': err, 'tolerance': tol,
            'ok': bool(err < tol)}


def extra_configs(torch, dev, model, ndim, nnz, jnnz, tensors):
    """The other single-GPU BASELINE configurations.  Every entry carries a `parity_check`: sampled members of the very buffers
    the timed launches wrote, against the CPU oracle on the same initial conditions."""
    from qgs_amd import _lib
    from oracle.oracle import OracleModel
    coo, val, jcoo, jval = tensors
    ora = OracleModel(ndim, coo, val, jcoo, jval)
    out = {}
    b, c, a = rk4_tableau()
    st = torch.cuda.current_stream().cuda_stream
    flops36 = 12 * nnz + 14 * ndim

    # -- config 2 with the reference's default write_steps = 1 (integrate.py:210-212): 100 steps, full record ----
    n, steps = 65536, 100
    t = grid(steps, 0.1)
    ic_h = np.random.RandomState(1).rand(ndim, n) * 0.01
    ic = torch.from_numpy(ic_h).to(dev)
    rec = torch.empty((steps + 1, ndim, n), dtype=torch.float64, device=dev)
    reps = 10                                                     # launches back to back per sample: steady clocks, as in a run of many windows

    def rec_runs(ws):
        for _ in range(reps):
            model.rk_integrate_device(n, n, ic.data_ptr(), t, 1, ws, b, c, a, rec.data_ptr(), st)
    ms0 = event_ms(torch, lambda: rec_runs(0), 4)[0] / reps
    ms = event_ms(torch, lambda: rec_runs(1), 4)[0] / reps       # (last: `rec` holds the write_steps = 1 record of the timed launches)
    rec_bytes = float(rec.numel() * 8)
    kname_rec = model.last_kernel_info()['name']
    clk_rec = clock_fields(model, flops36 * n * steps / (ms * 1e-3) / 1e12)
    idx = sample_members(n, 8)
    got = rec[:, :, torch.from_numpy(idx).to(dev)].cpu().numpy().transpose(2, 1, 0)          # (member, mode, record)
    ref = ora.integrate_runge_kutta_jit(t, np.ascontiguousarray(ic_h[:, idx].T), 1, 1, b, c, a)
    pc = parity_entry(kname_rec, 'first, middle and last record of the timed write_steps=1 record',
                      max(rel_err(got[:, :, k], ref[:, :, k]) for k in (0, steps // 2, steps)), 1e-12, idx)
    out['config2_write_steps_1'] = {
        'workload': 'MAOOAM-36, 65 536 members, 100 RK4 steps, write_steps=1: 101 records = %.2f GB (device layout); per launch, %d launches '
                    'back to back per sample' % (rec_bytes / 1e9, reps),
        'kernel': kname_rec, 'ms': ms, 'ms_same_run_without_records': ms0,
        'traj_steps_per_s': n * steps / (ms * 1e-3),
        'roofline': {'bound': 'hbm', 'achieved': rec_bytes / (ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': rec_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     'traffic': measured_traffic(kname_rec, launch_threads(kname_rec, n), ms), 'traffic_source': TRAFFIC_SOURCE,
                     'note': 'record bytes actually written / kernel time; the kernel also does the fp64 work of the steps',
                     'fp64_valu_frac': flops36 * n * steps / (ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                     # a plateau, reported as one: the floor of THIS formulation (issue slots of 2 076 fp64 instructions + 36 row
                     # stores per member-step on a lone wavefront per SIMD), profiles/r03_record_path.txt
                     'floor_ms': 0.546, 'frac_of_floor': 0.546 / ms,
                     'effective_clock_ghz': clk_rec.get('effective_clock_ghz'),
                     'fp64_valu_frac_at_effective_clock': clk_rec.get('frac_at_effective_clock')},
        'parity_check': pc}
    del rec

    # -- config 2 end to end through the host-pointer API: H2D + pack + kernel + unpack + D2H ----------------------
    n, steps = 65536, 1000
    t = grid(steps, 0.1)
    ic_h = np.random.RandomState(21217).rand(n, ndim) * 0.01
    model.rk_integrate(t, ic_h, 1, 0, b, c, a)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        res = model.rk_integrate(t, ic_h, 1, 0, b, c, a)
        ts.append(time.perf_counter() - t0)
    el = float(np.median(ts))
    idx = sample_members(n, 8)
    ref = ora.integrate_runge_kutta_jit(t, ic_h[idx], 1, 0, b, c, a)
    pc = parity_entry(model.last_kernel_info()['name'], 'final states returned by the last timed qgs_rk_integrate call',
                      rel_err(res[idx], ref), 1e-10, idx)
    del res
    out['config2_end_to_end_host_api'] = {
        'workload': 'MAOOAM-36, 65 536 members, 1000 RK4 steps, write_steps=0 through qgs_rk_integrate (NumPy in, NumPy out: '
                    'H2D + D2H over PCIe included)',
        'kernel': model.last_kernel_info()['name'], 'ms': el * 1e3, 'traj_steps_per_s': n * steps / el,
        'roofline': {'bound': 'fp64_valu', 'achieved': flops36 * n * steps / el / 1e12, 'peak': FP64_VALU_PEAK_TFLOPS,
                     'unit': 'TFLOP/s', 'frac': flops36 * n * steps / el / 1e12 / FP64_VALU_PEAK_TFLOPS},
        'parity_check': pc}

    # -- the same run through the host-pointer API: records delivered window by window into a page-locked block ----------
    n, steps = 65536, 100
    t = grid(steps, 0.1)
    ic_h = np.random.RandomState(21217).rand(n, ndim) * 0.01
    out_h = model.rk_integrate(t, ic_h, 1, 1, b, c, a)
    rec_bytes_h = float(out_h.nbytes)
    del out_h
    ts = []
    for k in range(4):
        t0 = time.perf_counter()
        out_h = model.rk_integrate(t, ic_h, 1, 1, b, c, a)
        ts.append(time.perf_counter() - t0)
        if k < 3:
            del out_h
    el = float(np.median(ts))
    idx = sample_members(n, 4)
    ref = ora.integrate_runge_kutta_jit(t, ic_h[idx], 1, 1, b, c, a)
    pc = parity_entry(model.last_kernel_info()['name'], 'full records (101) of sampled members as delivered to the host block by the last timed call',
                      rel_err(out_h[idx], ref), 1e-12, idx)
    del out_h
    # the PCIe floor of the same bytes: one page-locked device-to-host copy
    d_probe = torch.empty(int(rec_bytes_h) // 8, dtype=torch.float64, device=dev)
    h_probe = torch.empty(int(rec_bytes_h) // 8, dtype=torch.float64).pin_memory()
    tp = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h_probe.copy_(d_probe, non_blocking=True)
        torch.cuda.synchronize()
        tp.append(time.perf_counter() - t0)
    del d_probe, h_probe
    out['config2_write_steps_1_host_api'] = {
        'workload': 'MAOOAM-36, 65 536 members, 100 RK4 steps, write_steps=1 through qgs_rk_integrate: %.2f GB of records into a '
                    'page-locked host block in the reference layout (n_traj, ndim, n_records)' % (rec_bytes_h / 1e9),
        'kernel': model.last_kernel_info()['name'], 'record_windows': model.last_windows, 'ms': el * 1e3,
        'traj_steps_per_s': n * steps / el, 'host_gb_per_s': rec_bytes_h / el / 1e9,
        'plain_pinned_d2h_copy_ms': float(np.median(tp)) * 1e3,
        'note': 'PCIe-bound: the wall time is the device-to-host transfer; compute (0.6 ms) and layout conversion run under it',
        'parity_check': pc}

    # -- config 5 on ONE GPU: all 1 048 576 members x 1000 steps in one launch (the denominator of the 8-GPU curve) -------
    n, steps = 1048576, 1000
    t = grid(steps, 0.1)
    ic_h = np.random.RandomState(5).rand(ndim, n) * 0.01
    ic = torch.from_numpy(ic_h).to(dev)
    rec = torch.empty((1, ndim, n), dtype=torch.float64, device=dev)
    ms, _ = event_ms(torch, lambda: model.rk_integrate_device(n, n, ic.data_ptr(), t, 1, 0, b, c, a, rec.data_ptr(), st), 3)
    clk5 = clock_fields(model, flops36 * n * steps / (ms * 1e-3) / 1e12)
    idx = np.array(sorted(set(sample_members(65536, 6).tolist() + [524287, 1048575])))
    got = rec[0][:, torch.from_numpy(idx).to(dev)].cpu().numpy().T
    ref = ora.integrate_runge_kutta_jit(t, np.ascontiguousarray(ic_h[:, idx].T), 1, 0, b, c, a)[:, :, 0]
    pc = parity_entry(model.last_kernel_info()['name'], 'final states of sampled members of the timed 1 048 576-member launch', rel_err(got, ref), 1e-10, idx)
    out['config5_one_gpu'] = {
        'workload': 'BASELINE configs[4] on one GPU: MAOOAM-36, 1 048 576 members, 1000 RK4 steps, write_steps=0, one launch',
        'kernel': model.last_kernel_info()['name'], 'ms': ms, 'traj_steps_per_s': n * steps / (ms * 1e-3),
        'per_gpu_share_of_8': {'members': n // 8, 'note': 'one eighth of this launch is what each of 8 GPUs integrates in configs[4]; '
                               'the 8-GPU line carries its own measured single_gpu_reference'},
        'roofline': {'bound': 'fp64_valu', 'achieved': flops36 * n * steps / (ms * 1e-3) / 1e12, 'peak': FP64_VALU_PEAK_TFLOPS,
                     'unit': 'TFLOP/s', 'frac': flops36 * n * steps / (ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                     'traffic': measured_traffic(model.last_kernel_info()['name'], launch_threads(model.last_kernel_info()['name'], n), ms), 'traffic_source': TRAFFIC_SOURCE,
                     'effective_clock_ghz': clk5.get('effective_clock_ghz'), 'frac_at_effective_clock': clk5.get('frac_at_effective_clock'),
                     'clock_note': 'workgroup 0 lives for one sixteenth of this launch (16 waves of workgroups): the clock of its first wave'},
        'parity_check': pc}
    del ic, rec

    # -- config 4: tangent model, 16 384 members x 36 columns, 10 sub-steps per call, 100 calls; QR separately -------
    n, steps, n_tg, calls = 16384, 10, ndim, 100
    t = grid(steps, 0.01)
    ic_h = np.random.RandomState(2).rand(ndim, n) * 0.01
    ic = torch.from_numpy(ic_h).to(dev)
    tg = torch.zeros((ndim, n_tg, n), dtype=torch.float64, device=dev)
    for d in range(ndim):
        tg[d, d, :] = 1.0
    rec = torch.empty((1, ndim, n), dtype=torch.float64, device=dev)
    recm = torch.empty((1, ndim, n_tg, n), dtype=torch.float64, device=dev)
    rdiag = torch.empty((n_tg, n), dtype=torch.float64, device=dev)

    def tgls_calls():
        for _ in range(calls):
            model.rk_tgls_integrate_device(n, n, n_tg, ic.data_ptr(), tg.data_ptr(), t, 1, 0, b, c, a, False, 1.,
                                           rec.data_ptr(), recm.data_ptr(), st)
    ms, _ = event_ms(torch, tgls_calls, 3)
    ms_call = ms / calls
    kname = model.last_kernel_info()
    clk4 = model.kernel_clock()
    # the propagators and end states the timed calls left behind, before the QR overwrites them
    idx = sample_members(n, 4)
    sel = torch.from_numpy(idx).to(dev)
    got_fm = recm[0][:, :, sel].cpu().numpy().transpose(2, 0, 1)                             # (member, mode, column)
    got_y = rec[0][:, sel].cpu().numpy().T
    eye = np.repeat(np.eye(ndim)[np.newaxis], len(idx), axis=0)
    ref_y, ref_fm = ora.integrate_runge_kutta_tgls_jit(t, np.ascontiguousarray(ic_h[:, idx].T), eye, 1, 0, b, c, a, False, 1.)
    pc = parity_entry(kname['name'], 'propagators (36 x 36) and end states of sampled members after the timed calls',
                      max(rel_err(got_fm, ref_fm[..., 0]), rel_err(got_y, ref_y[..., 0])), 1e-11, idx)
    # the QR of the Benettin step on those propagators, checked against LAPACK member by member (np.linalg.qr is what the
    # reference calls, lyapunov.py:599-628), then timed on them (Q of an orthonormal matrix costs the same as the first one)
    model.batched_qr_device(n, n, ndim, n_tg, recm.data_ptr(), rdiag.data_ptr(), st)
    torch.cuda.synchronize()
    q_dev = recm[0][:, :, sel].cpu().numpy().transpose(2, 0, 1)
    r_dev = rdiag[:, sel].cpu().numpy().T
    qr_err = 0.0
    for k in range(len(idx)):
        q_ref, r_ref = np.linalg.qr(got_fm[k])
        qr_err = max(qr_err, float(np.abs(q_dev[k] - q_ref).max()), float(np.abs(r_dev[k] - np.diag(r_ref)).max() / max(1.0, np.abs(np.diag(r_ref)).max())))
    ms_qr, _ = event_ms(torch, lambda: model.batched_qr_device(n, n, ndim, n_tg, recm.data_ptr(), rdiag.data_ptr(), st), 9)
    clk_qr = model.kernel_clock()
    qr_kernel = model.last_kernel_info()
    qr_bytes = 2.0 * 8 * ndim * n_tg * n + 8.0 * n_tg * n                             # A in, Q out, diag(R) out
    qr_traffic = measured_traffic(qr_kernel['name'], launch_threads(qr_kernel['name'], n), ms_qr)
    flops_tgls = 4 * (2 * jnnz) + 4 * 2 * ndim ** 3 + 7 * 2 * ndim * ndim + flops36     # 4.02e5 at ndim 36
    rate = n * steps / (ms_call * 1e-3)
    # what the two kernels of a call EXECUTE: fp64 instructions of their step loops, counted in the ISA (tools/kisa.py ->
    # profiles/r06_kernel_isa.json; the tangent kernel's loop holds the tangent and the adjoint branch: half of it runs)
    executed = None
    try:
        with open(os.path.join(HERE, 'profiles', 'r06_kernel_isa.json')) as f:
            isa = json.load(f)
        tg_i = isa['bench:' + kname['name']]['hot_loop']['fp64'] / 2.0
        st_i = isa['bench:qgs_spec_rkstagesp_s4']['hot_loop']['fp64']
        executed = {'fp64_instr_per_column_step': tg_i, 'fp64_instr_per_member_step_trajectory_pass': st_i,
                    'flops_per_traj_step': 2.0 * (n_tg * tg_i + st_i)}
    except (OSError, KeyError, ValueError):
        pass
    out['config4_tgls'] = {
        'workload': 'MAOOAM-36 tangent model, 16 384 members x 36 tangent vectors (identity), 10 sub-steps per call, '
                    '%d calls timed together (trajectory pass + tangent pass per call)' % calls,
        'kernel': kname['name'], 'kernel_info': kname, 'ms_per_call': ms_call, 'traj_steps_per_s': rate,
        'qr_kernel': qr_kernel['name'], 'qr_kernel_info': qr_kernel, 'qr_ms': ms_qr,
        'qr_roofline': {'bound': 'hbm', 'bytes': qr_bytes, 'achieved': qr_bytes / (ms_qr * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': qr_bytes / (ms_qr * 1e-3) / 1e9 / HBM_PEAK_GBS, 'floor_ms': qr_bytes / (HBM_PEAK_GBS * 1e9) * 1e3,
                        'traffic': qr_traffic, 'traffic_source': TRAFFIC_SOURCE,
                        'traffic_over_algorithmic': (qr_traffic / qr_bytes) if qr_traffic else None,
                        'effective_clock_ghz': clk_qr[0] if clk_qr else None,
                        'max_abs_err_vs_lapack': qr_err, 'lapack_tolerance': 1e-12, 'lapack_ok': bool(qr_err < 1e-12),
                        'note': '16 384 matrices of 36 x 36 in, Q and diag(R) out, four matrices per wavefront; the time is the fp64 '
                                'instruction stream of the 70 dependent reflector steps at two wavefronts per SIMD (the matrices fill '
                                'the registers), not the memory system: profiles/r05_qr.md'},
        'roofline': {'bound': 'fp64_valu', 'achieved': rate * flops_tgls / 1e12, 'peak': FP64_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': rate * flops_tgls / 1e12 / FP64_VALU_PEAK_TFLOPS,
                     'flops_per_traj_step': flops_tgls,
                     'note': 'dense-matrix flop count of SURVEY 8(a) a8 (the kernel evaluates the sparse J w directly and '
                             'executes fewer)',
                     'traffic': (measured_traffic(kname['name'], launch_threads(kname['name'], n, n_tg), ms_call) or 0) +
                                (measured_traffic('qgs_spec_rkstagesp_s4', launch_threads('qgs_spec_rkstagesp_s4', n)) or 0) or None,
                     'traffic_source': TRAFFIC_SOURCE,
                     'algorithmic_bytes_per_call': 2 * 8 * (ndim + ndim * n_tg) * n * steps,
                     'executed': executed,
                     'executed_fp64_frac': (rate * executed['flops_per_traj_step'] / 1e12 / FP64_VALU_PEAK_TFLOPS) if executed else None,
                     'hbm_algorithmic_frac': rate * 2 * 8 * (ndim + ndim * n_tg) / 1e9 / HBM_PEAK_GBS,
                     # a plateau, reported as one: instruction floor of the two kernels of a call (0.57 + 0.07 ms,
                     # profiles/r03_tgls.txt: one wavefront per SIMD at 380 VGPRs, 405 accumulation-register moves per column-step)
                     'floor_ms': 0.64, 'frac_of_floor': 0.64 / ms_call,
                     'effective_clock_ghz': clk4[0] if clk4 else None,
                     'executed_fp64_frac_at_effective_clock': (rate * executed['flops_per_traj_step'] / 1e12 /
                                                               (FP64_VALU_PEAK_TFLOPS * clk4[0] / PEAK_CLOCK_GHZ)) if (executed and clk4) else None},
        'parity_check': pc}
    del tg, recm, rdiag

    # -- config 3: MAOOAM 6x6 (ndim 228), 65 536 members x 100 steps ----------------------------------------------------
    g = np.load(os.path.join(HERE, 'tests', 'golden', 't228.npz'))
    nd3 = int(g['ndim'])
    m3 = _lib.HipModel(nd3, g['coo'], g['val'], g['jcoo'], g['jval'], device=dev.index or 0)
    n, steps = 65536, 100
    t = grid(steps, 0.1)
    ic_h = np.random.RandomState(3).rand(nd3, n) * 0.01
    ic = torch.from_numpy(ic_h).to(dev)
    rec = torch.empty((1, nd3, n), dtype=torch.float64, device=dev)
    ms, _ = event_ms(torch, lambda: m3.rk_integrate_device(n, n, ic.data_ptr(), t, 1, 0, b, c, a, rec.data_ptr(), st), 3)
    clk3 = m3.kernel_clock()
    idx = sample_members(n, 8)
    got = rec[0][:, torch.from_numpy(idx).to(dev)].cpu().numpy().T
    ref = OracleModel(nd3, g['coo'], g['val']).integrate_runge_kutta_jit(t, np.ascontiguousarray(ic_h[:, idx].T), 1, 0, b, c, a)[:, :, 0]
    pc = parity_entry(m3.last_kernel_info()['name'], 'final states of sampled members of the timed launch (100 steps)', rel_err(got, ref), 1e-12, idx)
    flops228 = 12 * len(g['val']) + 14 * nd3
    rate = n * steps / (ms * 1e-3)
    k3 = m3.last_kernel_info()
    thr3 = launch_threads(k3['name'], n)
    # what the kernel executes: the generator's own count of fp64 instructions per workgroup (64 members) and stage, in the header
    # of the generated source; its floor: that many instructions x 4 cycles on 4 SIMDs, 4 stages x `steps` x 4 workgroups per CU
    instr = None
    try:
        import re
        mm = re.search(r'per stage and 64 members:.*?(\d+) fp64 instructions', m3.kernel_source())
        instr = int(mm.group(1)) if mm else None
    except Exception:
        instr = None
    floor_ms = (instr * 4.0 / 4.0) * (4 * steps * (n / 64.0) / 256.0) / (PEAK_CLOCK_GHZ * 1e9) * 1e3 if instr else None
    traffic3 = measured_traffic(k3['name'], thr3, ms)
    out['config3_maooam228'] = {
        'workload': 'MAOOAM 6x6 atm / 6x6 ocean (ndim 228, %d tensor entries; tests/golden/t228.npz), 65 536 members, 100 RK4 steps, '
                    'write_steps=0' % len(g['val']),
        'kernel': k3['name'], 'kernel_info': k3, 'ms': ms, 'traj_steps_per_s': rate,
        'mode_updates_per_s': rate * nd3,
        'roofline': {'bound': 'fp64_valu', 'achieved': rate * flops228 / 1e12, 'peak': FP64_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': rate * flops228 / 1e12 / FP64_VALU_PEAK_TFLOPS, 'flops_per_traj_step': flops228,
                     'traffic': traffic3, 'traffic_source': TRAFFIC_SOURCE,
                     'algorithmic_bytes_per_launch': 2 * 8 * nd3 * n * steps,
                     'traffic_over_algorithmic': (traffic3 / (2.0 * 8 * nd3 * n * steps)) if traffic3 else None,
                     'hbm_algorithmic_frac': rate * 2 * 8 * nd3 / 1e9 / HBM_PEAK_GBS,
                     # executed view: the generator's count of fp64 instructions (an FMA = 2 flop) against the same peak
                     # (an instruction of a wavefront is one lane-operation per member: 4 stages x the count per workgroup-stage)
                     'executed_fp64_instr_per_step': (instr * 4) if instr else None,
                     'executed_fp64_frac': (rate * (instr * 4) * 2 / 1e12 / FP64_VALU_PEAK_TFLOPS) if instr else None,
                     'executed': executed_view(k3['name'], thr3, ms, steps),
                     # the instruction floor of THIS formulation: the generator's count at full issue rate, no waits
                     'floor_ms': floor_ms, 'frac_of_floor': (floor_ms / ms) if floor_ms else None,
                     'effective_clock_ghz': clk3[0] if clk3 else None,
                     'frac_at_effective_clock': (rate * flops228 / 1e12 / (FP64_VALU_PEAK_TFLOPS * clk3[0] / PEAK_CLOCK_GHZ)) if clk3 else None},
        'parity_check': pc}
    # -- the kernels behind SURVEY 8(f) at their hard sizes: the Benettin interval of the reference's default n_vec = n_dim at ndim 228
    #    (tangent model of 228 vectors + the QR of 228 x 228 matrices, lyapunov.py:599-628) and the rank-5 steppers (sparse_mul5,
    #    sparse_mul.py:121-158) -- each timed, set against its bound, and checked against the oracle / LAPACK
    try:
        out['f_rows'] = f_row_configs(torch, dev, m3, g, (b, c, a), st)
    except Exception as e:                                                   # a side measurement never costs the entries above
        out['f_rows'] = {'error': repr(e)}
    m3.close()
    return out


def f_row_configs(torch, dev, m3, g3, tableau, st):
    from qgs_amd import _lib
    from oracle.oracle import OracleModel
    b, c, a = tableau
    res = {}
    nd = int(g3['ndim'])
    jnnz = len(g3['jval'])
    # -- one Benettin interval at ndim 228 with the full basis: 1 024 members x 228 tangent vectors, 10 sub-steps ---------------------
    n, nv, steps = 1024, nd, 10
    ld = n
    t = grid(steps, 0.01)
    ic_h = np.random.RandomState(2).rand(nd, ld) * 0.01
    ic = torch.from_numpy(ic_h).to(dev)
    q = torch.zeros((nd, nv, ld), dtype=torch.float64, device=dev)
    for d in range(nd):
        q[d, d, :] = 1.0
    qn = torch.empty((1, nd, nv, ld), dtype=torch.float64, device=dev)
    yend = torch.empty((1, nd, ld), dtype=torch.float64, device=dev)
    rd = torch.empty((nv, ld), dtype=torch.float64, device=dev)

    def tgls():
        m3.rk_tgls_integrate_device(n, ld, nv, ic.data_ptr(), q.data_ptr(), t, 1, 0, b, c, a, False, 1., yend.data_ptr(), qn.data_ptr(), st)
    ms_t, _ = event_ms(torch, tgls, 3)
    k_t = m3.last_kernel_info()
    idx = np.array([0, 63, 64, 1023])
    sel = torch.from_numpy(idx).to(dev)
    got_fm = qn[0][:, :, sel].cpu().numpy().transpose(2, 0, 1)
    got_y = yend[0][:, sel].cpu().numpy().T
    ora = OracleModel(nd, g3['coo'], g3['val'], g3['jcoo'], g3['jval'])
    eye = np.repeat(np.eye(nd)[np.newaxis], len(idx), axis=0)
    ref_y, ref_fm = ora.integrate_runge_kutta_tgls_jit(t, np.ascontiguousarray(ic_h[:, idx].T), eye, 1, 0, b, c, a, False, 1.)
    pc_t = parity_entry(k_t['name'], 'propagators (228 x 228) and end states of sampled members after the timed calls',
                        max(rel_err(got_fm, ref_fm[..., 0]), rel_err(got_y, ref_y[..., 0])), 1e-11, idx)
    flops_pair_step = 12 * jnnz + 14 * nd                   # 3 flop per Jacobian-tensor term and stage + the 7 axpys, as SURVEY 8(d) counts the stepper
    pair_rate = n * nv * steps / (ms_t * 1e-3)
    alg_bytes_t = 2.0 * 8 * (nd + nd * nv) * n * steps
    mw = re.search(r'lds[a-z]*(\d+)$', k_t['name'])
    waves_t = int(mw.group(1)) if mw else 16
    thr_t = (n + 15) // 16 * ((nv + 3) // 4) * 64 * waves_t     # workgroups of 16 members x 4 columns; 16 wavefronts each, hand-scheduled: 8
    tr_t = measured_traffic(k_t['name'], thr_t, None)
    res['tgls228_full_basis'] = {
        'workload': 'MAOOAM 6x6 tangent model: 1 024 members x 228 tangent vectors (identity), 10 sub-steps of 0.01: one Benettin interval of the '
                    "reference's default n_vec = n_dim (trajectory pass with stage store + tangent pass)",
        'kernel': k_t['name'], 'kernel_info': k_t, 'ms': ms_t, 'pair_steps_per_s': pair_rate,
        'roofline': {'bound': 'fp64_valu', 'achieved': pair_rate * flops_pair_step / 1e12, 'peak': FP64_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': pair_rate * flops_pair_step / 1e12 / FP64_VALU_PEAK_TFLOPS, 'flops_per_pair_step': flops_pair_step,
                     'algorithmic_bytes_per_call': alg_bytes_t, 'hbm_algorithmic_frac': alg_bytes_t / (ms_t * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     'traffic': tr_t, 'traffic_source': TRAFFIC_SOURCE},
        'parity_check': pc_t}
    # -- its QR: 1 024 matrices of 228 x 228 (blocked Householder, 16-column panels) ------------------------------------------------------
    a_host = qn[0][:, :, sel].cpu().numpy().transpose(2, 0, 1).copy()
    m3.batched_qr_device(n, ld, nd, nv, qn.data_ptr(), rd.data_ptr(), st)
    torch.cuda.synchronize()
    k_q = m3.last_kernel_info()
    q_dev = qn[0][:, :, sel].cpu().numpy().transpose(2, 0, 1)
    r_dev = rd[:, sel].cpu().numpy().T
    qr_err = 0.0
    for k in range(len(idx)):
        q_ref, r_ref = np.linalg.qr(a_host[k])
        qr_err = max(qr_err, float(np.abs(q_dev[k] - q_ref).max()), float(np.abs(r_dev[k] - np.diag(r_ref)).max() / max(1.0, np.abs(np.diag(r_ref)).max())))
    ms_q, _ = event_ms(torch, lambda: m3.batched_qr_device(n, ld, nd, nv, qn.data_ptr(), rd.data_ptr(), st), 5)
    qr_bytes = 2.0 * 8 * nd * nv * n + 8.0 * nv * n
    qr_flops = (2.0 * nd * nv * nv - 2.0 * nv ** 3 / 3.0) * 2.0 * n          # dgeqrf + dorgqr of an m x n matrix: 2 (2 m n^2 - 2 n^3 / 3)
    thr_q = 8 * ((n + 7) // 8) * 256
    tr_q = measured_traffic(k_q['name'], thr_q, ms_q)
    res['qr228_full_basis'] = {
        'workload': '1 024 matrices of 228 x 228 (the propagated basis above): Q and diag(R), LAPACK sign convention',
        'kernel': k_q['name'], 'kernel_info': k_q, 'ms': ms_q,
        'roofline': {'bound': 'neither: latency (panel factorisation chains, LDS, barriers)',
                     'achieved': qr_bytes / (ms_q * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': qr_bytes / (ms_q * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     'bytes': qr_bytes, 'floor_ms_hbm': qr_bytes / (HBM_PEAK_GBS * 1e9) * 1e3,
                     'fp64_flops': qr_flops, 'fp64_valu_frac': qr_flops / (ms_q * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                     'floor_ms_fp64': qr_flops / (FP64_VALU_PEAK_TFLOPS * 1e12) * 1e3,
                     'traffic': tr_q, 'traffic_source': TRAFFIC_SOURCE, 'traffic_over_algorithmic': (tr_q / qr_bytes) if tr_q else None,
                     'max_abs_err_vs_lapack': qr_err, 'lapack_tolerance': 1e-11, 'lapack_ok': bool(qr_err < 1e-11)},
        'parity_check': {'kernel': k_q['name'], 'checked': 'Q and diag(R) of sampled members against np.linalg.qr', 'members': [int(v) for v in idx],
                         'max_abs_err_vs_lapack': qr_err, 'tolerance': 1e-11, 'ok': bool(qr_err < 1e-11)}}
    del q, qn, yend, rd, ic
    # -- rank-5 steppers: the dynamic-T (d38) and T4 (q38) MAOOAM models, 65 536 members x 100 steps ---------------------------------------
    for name in ('d38', 'q38'):
        g = np.load(os.path.join(HERE, 'tests', 'golden', name + '.npz'))
        nd5 = int(g['ndim'])
        m5 = _lib.HipModel(nd5, g['coo'], g['val'], g['jcoo'], g['jval'], device=dev.index or 0)
        n, steps = 65536, 100
        t = grid(steps, 0.1)
        rng = np.random.RandomState(21217)
        ic_h = rng.rand(n, nd5) * 0.01
        ic_h[:, 10] += 1.5                                   # (the temperature anomalies these models are built around, tools/rank5_bench.py)
        ic_h[:, 29] += 3.
        x = torch.from_numpy(np.ascontiguousarray(ic_h.T)).to(dev)
        rec = torch.empty((1, nd5, n), dtype=torch.float64, device=dev)
        ms5, _ = event_ms(torch, lambda: m5.rk_integrate_device(n, n, x.data_ptr(), t, 1, 0, b, c, a, rec.data_ptr(), st), 3)
        k5 = m5.last_kernel_info()
        clk5 = m5.kernel_clock()
        idx5 = sample_members(n, 6)
        got = rec[0][:, torch.from_numpy(idx5).to(dev)].cpu().numpy().T
        ref = OracleModel(nd5, g['coo'], g['val']).integrate_runge_kutta_jit(t, ic_h[idx5], 1, 0, b, c, a)[:, :, 0]
        nnz5 = len(g['val'])
        # flops of one trajectory step: as the reference writes the contraction (sparse_mul5: 4 multiplications + 1 addition per entry
        # and stage) and in the bilinear form the library evaluates (products shared between monomials computed once as derived
        # monomials: 3 flop per reduced term + 1 per derived monomial) -- the fraction of the peak is quoted on the SECOND: it is the
        # work of the algorithm that runs, the first over-counts what any implementation with common sub-expressions has to do
        flops5_ref = 4 * 5 * nnz5 + 14 * nd5
        red_t = m5.n_reduced_terms[0]
        flops5 = 4 * (3 * red_t + m5.n_derived[0]) + 14 * nd5
        rate5 = n * steps / (ms5 * 1e-3)
        thr5 = launch_threads(k5['name'], n)
        res['rank5_' + name] = {
            'workload': '%s MAOOAM (%s, ndim %d, %d rank-5 tensor entries = %d bilinear terms + %s derived monomials), 65 536 members, 100 RK4 steps, write_steps=0'
                        % ('dynamic-T' if name == 'd38' else 'T4', name, nd5, nnz5, red_t, m5.n_derived[0]),
            'kernel': k5['name'], 'kernel_info': k5, 'ms': ms5, 'traj_steps_per_s': rate5,
            'roofline': {'bound': 'fp64_valu', 'achieved': rate5 * flops5 / 1e12, 'peak': FP64_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': rate5 * flops5 / 1e12 / FP64_VALU_PEAK_TFLOPS, 'flops_per_traj_step': flops5,
                         'flops_per_traj_step_as_the_reference_writes_it': flops5_ref,
                         'frac_on_the_reference_form': rate5 * flops5_ref / 1e12 / FP64_VALU_PEAK_TFLOPS,
                         'note': 'frac: flops of the bilinear form the kernel evaluates (3 per reduced term and stage + the derived monomials); '
                                 'frac_on_the_reference_form counts 5 per rank-5 tensor entry and stage and can exceed 1: the reduction removes work',
                         'hbm_algorithmic_frac': rate5 * 2 * 8 * nd5 / 1e9 / HBM_PEAK_GBS,
                         'traffic': measured_traffic(k5['name'], thr5, ms5), 'traffic_source': TRAFFIC_SOURCE,
                         'effective_clock_ghz': clk5[0] if clk5 else None},
            'parity_check': parity_entry(k5['name'], 'final states of sampled members of the timed launch (100 steps)', rel_err(got, ref), 1e-10, idx5)}
        m5.close()
        del x, rec
    return res


# ---------------------------------------------------------------------------------------------------------------
# cold start (rank 0, N = 1): create_tendencies -> first 1000-step result in a fresh process
# ---------------------------------------------------------------------------------------------------------------
_COLD_CHILD = r"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, %(here)r)
import torch
torch.zeros(1, device='cuda'); torch.cuda.synchronize()          # process start-up (imports, HIP context) is not part of the figure
import bench
from qgs_amd import _lib
_lib.lib()
t0 = time.perf_counter()
ndim, coo, val, jcoo, jval, _ = bench.load_model_tensors(kd=%(kd)r)
t1 = time.perf_counter()
m = _lib.HipModel(ndim, coo, val, jcoo, jval, device=0)
b, c, a = bench.rk4_tableau()
ic = np.random.RandomState(1).rand(65536, ndim) * 0.01
res = m.rk_integrate(bench.grid(1000, 0.1), ic, 1, 0, b, c, a)
t2 = time.perf_counter()
files = len([f for f in os.listdir(os.environ['QGS_HIP_CACHE_DIR']) if f.endswith('.hsaco')])
print('COLD ' + json.dumps({'seconds': t2 - t0, 'create_tendencies_s': t1 - t0, 'kernel': m.last_kernel_info()['name'],
                            'cache_entries': files, 'checksum': float(res.sum())}))
"""


def cold_start():
    """Seconds from `create_tendencies` to the first 65 536-member x 1000-step result, each in a FRESH process: on an empty
    kernel cache with the compiler's own cache switched off (everything that run needs is generated and really compiled), then
    with other parameter values on the cache the first run left (no compilation: the code objects do not depend on values),
    and with the shipped parameter set on the cache that ships with the tree."""
    import tempfile
    out = {}
    with tempfile.TemporaryDirectory(prefix='qgs_cold_') as d:
        for tag, kd, cache, comgr in (('empty_cache', 0.0291, d, '0'), ('structure_warm_cache', 0.0296, d, None),
                                      ('shipped_cache', 0.0290, os.path.join(HERE, 'qgs_amd', 'kcache'), None)):
            env = dict(os.environ, QGS_HIP_CACHE_DIR=cache)
            if comgr is not None:
                env['AMD_COMGR_CACHE'] = comgr            # hiprtc's own on-disk cache (~/.cache/comgr) would hide the compilation
            try:
                p = subprocess.run([sys.executable, '-c', _COLD_CHILD % {'here': HERE, 'kd': kd}], stdout=subprocess.PIPE,
                                   stderr=subprocess.PIPE, timeout=600, env=env)
                line = [ln for ln in p.stdout.decode().splitlines() if ln.startswith('COLD ')]
                out[tag] = json.loads(line[0][5:]) if line else {'error': p.stderr.decode()[-400:]}
                out[tag]['kd'] = kd
            except (OSError, subprocess.TimeoutExpired, ValueError) as e:
                out[tag] = {'error': repr(e)}
    out['note'] = ('fresh process each; process start-up (imports, HIP context) excluded; empty_cache: empty kernel cache and '
                   'AMD_COMGR_CACHE=0, i.e. generation + hiprtc compilation of the packing-free path (one code object: the fused stepper); '
                   'structure_warm_cache: another kd on that cache (the code object does not depend on parameter values)')
    return out


_COLD_CHILD_228 = r"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, %(here)r)
import torch
torch.zeros(1, device='cuda'); torch.cuda.synchronize()
import bench
from qgs_amd import _lib
_lib.lib()
g = np.load(os.path.join(%(here)r, 'tests', 'golden', 't228.npz'))
ndim = int(g['ndim'])
b, c, a = bench.rk4_tableau()
ic = np.random.RandomState(1).rand(%(members)d, ndim) * 0.01
t0 = time.perf_counter()
m = _lib.HipModel(ndim, g['coo'], g['val'], g['jcoo'], g['jval'], device=0)
m.set_kernel(%(kind)d)
res = m.rk_integrate(bench.grid(%(steps)d, 0.1), ic, 1, 0, b, c, a)
t1 = time.perf_counter()
res2 = m.rk_integrate(bench.grid(%(steps)d, 0.1), ic, 1, 0, b, c, a)
t2 = time.perf_counter()
files = len([f for f in os.listdir(os.environ['QGS_HIP_CACHE_DIR']) if f.endswith('.hsaco')])
print('COLD ' + json.dumps({'seconds': t1 - t0, 'second_call_seconds': t2 - t1, 'kernel': m.last_kernel_info()['name'],
                            'cache_entries': files, 'checksum': float(res.sum())}))
"""


def cold_start_228():
    """BASELINE configs[2] (MAOOAM 6x6, ndim 228) from the tensors to the first result of 65 536 members x 100 steps, each in a FRESH
    process (host tensor assembly excluded: 5 s of NumPy, not the GPU path's): on an empty kernel cache with the compiler's own
    cache off (automatic mode: a call of this size pays for the one-second compilation of the hand-scheduled LDS stepper), the same
    with the specialised kernel requested outright, the cache that run left, the cache that ships with the tree, and the generic
    kernels on request (what a call too short to compile for gets).  `auto_small_run_empty_cache`: 4 096 members x 20 steps in
    automatic mode on an empty cache -- a short first run takes the generic kernel instead of waiting."""
    import tempfile
    out = {}
    with tempfile.TemporaryDirectory(prefix='qgs_cold228_') as d, tempfile.TemporaryDirectory(prefix='qgs_cold228b_') as d2:
        for tag, kind, members, steps, cache, comgr in (('empty_cache', 0, 65536, 100, d, '0'),
                                                        ('empty_cache_specialised_kernel_requested', 2, 65536, 100, d, '0'),
                                                        ('same_cache_again', 0, 65536, 100, d, None),
                                                        ('shipped_cache', 0, 65536, 100, os.path.join(HERE, 'qgs_amd', 'kcache'), None),
                                                        ('generic_kernels', 1, 65536, 100, d, None),
                                                        ('auto_small_run_empty_cache', 0, 4096, 20, d2, '0')):
            env = dict(os.environ, QGS_HIP_CACHE_DIR=cache)
            if comgr is not None:
                env['AMD_COMGR_CACHE'] = comgr
            try:
                p = subprocess.run([sys.executable, '-c', _COLD_CHILD_228 % {'here': HERE, 'kind': kind, 'members': members, 'steps': steps}],
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
                line = [ln for ln in p.stdout.decode().splitlines() if ln.startswith('COLD ')]
                out[tag] = json.loads(line[0][5:]) if line else {'error': p.stderr.decode()[-400:]}
                out[tag].update({'members': members, 'rk_steps': steps})
            except (OSError, subprocess.TimeoutExpired, ValueError) as e:
                out[tag] = {'error': repr(e)}
    out['note'] = ('fresh process each, start-up excluded; seconds = model creation + first qgs_rk_integrate (NumPy in, NumPy out), '
                   'second_call_seconds = the same call again in that process')
    return out


def default_members(n_gpus):
    """Ensemble members per GPU when --members is not given: BASELINE configs[4] (1 048 576 members over 8 GPUs) at 8 GPUs,
    configs[1] (65 536 members on one GPU) per GPU otherwise."""
    return 131072 if n_gpus == 8 else 65536


# ---------------------------------------------------------------------------------------------------------------
# the rank body: what every rank does between the barriers, written against an `engine` (the HIP engine below; a host stand-in
# in tests/test_bench_rank_body_cpu.py runs the same control flow at world size 8 on gloo)
# ---------------------------------------------------------------------------------------------------------------
def rank_body(engine, dist, use_dist, steps, warmup, reference_passes=0):
    """Warm-up passes, then exactly `steps` timed passes bracketed by barrier + device synchronisation; returns
    (elapsed seconds: max over ranks, seconds of the gather-free reference passes: max over ranks or None, one gather in ms or None).

    A pass computes into output buffer k % 2 and (with a process group) starts the asynchronous gather of that buffer onto rank 0;
    before a buffer is computed into again, the gather that still reads it is waited for; the timed region ends when the
    last gathers have completed on every rank."""
    pending = [None, None]                                             # in-flight gathers of output buffers 0 and 1

    def one_pass(record, k, gather=True):
        q = k % 2
        if pending[q] is not None:                                     # the gather that still reads this buffer
            pending[q].wait()
            pending[q] = None
        engine.compute(q, record)
        if use_dist and gather:
            # the only collective: gather of the final states onto rank 0.  It is asynchronous (RCCL's own stream), so it
            # overlaps the next pass
            pending[q] = engine.start_gather(q)

    def drain():
        for i in (0, 1):
            if pending[i] is not None:
                pending[i].wait()
                pending[i] = None

    def barrier():
        if use_dist:
            dist.barrier()
        engine.synchronize()

    def max_over_ranks(x):
        return engine.max_over_ranks(x) if use_dist else x

    for k in range(warmup):
        one_pass(False, k)
    drain()
    # the same passes on every GPU at once without the gather: the per-GPU rate the N-GPU value is to be compared with
    reference = None
    if reference_passes > 0:
        barrier()
        t0 = time.perf_counter()
        for k in range(reference_passes):
            one_pass(False, k, gather=False)
        barrier()
        reference = max_over_ranks(time.perf_counter() - t0)
    barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        one_pass(True, k)
    drain()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)

    # duration of ONE gather, outside the timed region: from its start to its completion on the compute stream
    gather_ms = None
    if use_dist:
        gts = []
        for _ in range(5):
            barrier()
            gts.append(engine.timed_gather(0))
        gather_ms = float(np.median(gts))
    return elapsed, reference, gather_ms


class HipEngine(object):
    """One rank's share of the headline workload on its GPU: pack -> fused stepper -> unpack into one of two output buffers,
    asynchronous RCCL gather of a buffer onto rank 0."""

    def __init__(self, torch, dist, dev, model, ndim, n_traj, ic_host, time_grid, tableau, world, rank, use_dist):
        from qgs_amd.parallel import ShardedEnsemble, RootGather
        self.torch, self.dist, self.dev, self.model = torch, dist, dev, model
        self.ndim, self.n_traj, self.time_grid, self.tableau = ndim, n_traj, time_grid, tableau
        self.ld = (n_traj + 63) // 64 * 64
        self.d_ic_rows = torch.from_numpy(ic_host).to(dev)                      # resident in HBM, reference layout
        self.d_ic_modes = torch.empty((ndim, self.ld), dtype=torch.float64, device=dev)
        self.d_rec = torch.empty((1, ndim, self.ld), dtype=torch.float64, device=dev)
        self.d_out = [torch.empty((n_traj, ndim), dtype=torch.float64, device=dev) for _ in range(2)]   # double buffer
        ens = ShardedEnsemble(world * n_traj)                              # contiguous member blocks, one per rank
        assert ens.n_local == n_traj
        self.root = RootGather(ens, dst=0)
        self.d_all = torch.empty((world * n_traj, ndim), dtype=torch.float64, device=dev) if (use_dist and rank == 0) else None
        self.stream = torch.cuda.current_stream().cuda_stream
        self.kern_events = []

    def compute(self, q, record):
        torch, model, b, c, a = self.torch, self.model, *self.tableau
        model.pack_states(self.n_traj, self.ld, self.d_ic_rows.data_ptr(), self.d_ic_modes.data_ptr(), self.stream)
        if record:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        model.rk_integrate_device(self.n_traj, self.ld, self.d_ic_modes.data_ptr(), self.time_grid, 1, 0, b, c, a, self.d_rec.data_ptr(), self.stream)
        if record:
            e1.record()
            self.kern_events.append((e0, e1))
        model.unpack_records(self.n_traj, self.ld, self.ndim, 1, self.d_rec.data_ptr(), self.d_out[q].data_ptr(), self.stream)

    def start_gather(self, q):
        work, _ = self.root.start(self.d_out[q], out=self.d_all, async_op=True)
        return work

    def timed_gather(self, q):
        torch = self.torch
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g0.record()
        work = self.start_gather(q)
        if work is not None:
            work.wait()                                            # the compute stream now waits for RCCL's stream
        g1.record()
        g1.synchronize()
        return g0.elapsed_time(g1)

    def synchronize(self):
        self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


# ---------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200, help='timed passes (default 200: 0.9 s of GPU work, long enough for an smi sampler to see it)')
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--members', type=int, default=None,
                    help='ensemble members per GPU (default: 131 072 with --gpus 8 = BASELINE configs[4], 1 048 576 members over 8 GPUs; '
                         '65 536 = configs[1] otherwise)')
    ap.add_argument('--rk-steps', type=int, default=1000, help='RK4 steps per pass')
    ap.add_argument('--kernel', choices=['auto', 'generic', 'spec'], default='auto')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra-configs', action='store_true', help='skip the configs 3 / 4 / write_steps=1 / host-API entries')
    ap.add_argument('--no-cold-start', action='store_true', help='skip the cold-start probe (two fresh processes, one of which compiles)')
    ap.add_argument('--busy-fill', action='store_true', help='after the timed region, repeat the same passes untimed until the process has kept the GPU '
                    'busy for ~1.2 s (makes a short run visible to a 1 Hz smi sampler; changes no reported number)')
    ap.add_argument('--force-dist', action='store_true', help='initialise the RCCL process group even at world size 1 (plumbing check)')
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error('--gpus must be >= 1')
    if args.members is None:
        args.members = default_members(args.gpus)

    under_launcher = 'RANK' in os.environ and 'WORLD_SIZE' in os.environ
    if not under_launcher and args.gpus > 1:
        # typed directly: this process stays off the GPU and starts one fresh child per rank
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if rank == 0:
            print('bench.py: --gpus %d but the launcher started %d ranks' % (args.gpus, world), file=sys.stderr)
        sys.exit(4)
    n_vis = torch.cuda.device_count()
    if n_vis < 1 or not torch.cuda.is_available():
        print('bench.py: no GPU visible; the HIP path has no CPU fallback', file=sys.stderr)
        sys.exit(1)
    if local_rank >= n_vis:
        print('bench.py: %d GPUs requested, %d visible' % (world, n_vis), file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    rccl = rccl_report(torch, dist, dev, world, rank) if use_dist else None

    from qgs_amd import _lib

    ndim, coo, val, jcoo, jval, tensor_src = load_model_tensors()
    model = _lib.HipModel(ndim, coo, val, jcoo, jval, device=local_rank)
    model.set_kernel({'auto': 0, 'generic': 1, 'spec': 2}[args.kernel])

    n_traj, rk_steps, dt = args.members, args.rk_steps, 0.1
    b, c, a = rk4_tableau()
    time_grid = grid(rk_steps, dt)

    # synthetic initial conditions (the distribution qgs_maooam.py:108 uses), different per rank
    rng = np.random.RandomState(21217 + rank)
    ic_host = rng.rand(n_traj, ndim) * 0.01
    engine = HipEngine(torch, dist, dev, model, ndim, n_traj, ic_host, time_grid, (b, c, a), world, rank, use_dist)

    ref_passes = min(args.steps, 50) if use_dist else 0     # (with --force-dist also at world size 1: the N > 1 code path, exercised on one GPU)
    elapsed, ref_elapsed, gather_ms = rank_body(engine, dist, use_dist, args.steps, args.warmup, reference_passes=ref_passes)

    kern_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in engine.kern_events])) if engine.kern_events else float('nan')
    kinfo = model.last_kernel_info()
    clock = None
    try:
        clock = model.kernel_clock()                     # of the last timed pass's stepper launch (in-kernel probe, qgs_kernel_clock)
    except Exception:
        clock = None
    fma_rate = None
    if rank == 0 and world == 1:                          # (behind the timed region; ~30 ms)
        try:
            fma_rate = _lib.fp64_fma_rate(dev.index or 0, 30.0)
        except Exception:
            fma_rate = None
    # --busy-fill (off by default): a short timed region (the driver's --steps 20 is 90 ms) is invisible to an smi sampler; with the
    # flag the same passes are run on, untimed and outside every statistic above, until this process has kept the GPU busy for ~1.2 s.
    busy_fill_passes = 0
    if args.busy_fill and args.steps > 0 and elapsed < 1.0:
        per_pass = elapsed / args.steps
        busy_fill_passes = int(min(2000, max(1, (1.2 - elapsed) / max(per_pass, 1e-5))))
        for k in range(busy_fill_passes):
            engine.compute(k % 2, False)
        engine.synchronize()

    result = None
    if rank == 0:
        total_traj_steps = float(n_traj) * rk_steps * world * args.steps
        value = total_traj_steps / elapsed
        bytes_per_traj_step = 2 * 8 * ndim                                 # state read once + written once per step
        flops_per_traj_step = 12 * len(val) + 14 * ndim                    # 4 716 at MAOOAM-36
        alg_bytes = float(bytes_per_traj_step) * n_traj * rk_steps
        alg_flops = float(flops_per_traj_step) * n_traj * rk_steps
        tflops = alg_flops / (kern_ms * 1e-3) / 1e12
        alg_gbs = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic = measured_traffic(kinfo['name'], launch_threads(kinfo['name'], n_traj), kern_ms)
        # what the kernel executes per trajectory-step: fp64 instructions of its step loop, counted in the ISA (tools/kisa.py ->
        # profiles/r06_kernel_isa.json, a committed constant), an FMA = 2 flop
        fp64_instr = None
        try:
            with open(os.path.join(HERE, 'profiles', 'r06_kernel_isa.json')) as f:
                fp64_instr = json.load(f)['bench:' + kinfo['name']]['hot_loop']['fp64']
        except (OSError, KeyError, ValueError):
            pass
        result = {
            'metric': 'ensemble trajectory-steps/sec fp64, MAOOAM-36',
            'value': value, 'unit': 'traj-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'MAOOAM 2x2 atm / 2x4 ocean (36 modes) fp64, %d-member ensemble per GPU, %d RK4 steps '
                                   'per pass, write_steps=0 (%s)'
                                   % (n_traj, rk_steps,
                                      'BASELINE configs[4]: %d members sharded across %d GPUs, RCCL gather only' % (n_traj * world, world)
                                      if (world == 8 and n_traj == 131072) else
                                      ('BASELINE configs[1]' if n_traj == 65536 else 'configs[1] model at a non-default ensemble size')),
                       'members_per_gpu': n_traj, 'members_total': n_traj * world, 'rk_steps_per_pass': rk_steps, 'ndim': ndim,
                       'tensor_nnz': int(len(val)), 'dt': dt, 'tensor_source': tensor_src,
                       'parallelism': 'members sharded x%d, RCCL gather of final states onto rank 0 (async, overlapped)' % world,
                       'kernel': kinfo},
            'mode_updates_per_s': value * ndim,
            'untimed_busy_fill_passes': busy_fill_passes,   # extra passes after the timed region so that a 1 Hz smi sampler sees the GPU busy
            'gather_ms': gather_ms,        # one RCCL gather of the final states onto rank 0, start to completion (None without a process group)
            'roofline': {'bound': 'fp64_valu', 'achieved': tflops, 'peak': FP64_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': tflops / FP64_VALU_PEAK_TFLOPS,
                         # what THIS box sustains on independent fp64 FMAs alone (qgs_fp64_fma_rate, ~30 ms behind the timed region: the
                         # board lowers the clock under fp64 load, so the nominal peak is not reachable by any instruction stream) and the
                         # EXECUTED flops of the kernel as a fraction of it
                         'sustained_fma_tflops': fma_rate[0] if fma_rate else None, 'sustained_fma_ms': fma_rate[1] if fma_rate else None,
                         'executed_frac_of_sustained_fma': (fp64_instr * 2.0 * n_traj * rk_steps / (kern_ms * 1e-3) / 1e12 / fma_rate[0])
                                                           if (fp64_instr and fma_rate) else None,
                         # the clock the timed launch ran at (in-kernel probe: shader-clock counter over the 100 MHz counter of
                         # workgroup 0, which lives for the whole launch) and the fraction of the FP64 peak AT that clock
                         'effective_clock_ghz': clock[0] if clock else None,
                         'frac_at_effective_clock': (tflops / (FP64_VALU_PEAK_TFLOPS * clock[0] / PEAK_CLOCK_GHZ)) if clock else None,
                         'clock_probe_ms': clock[1] if clock else None, 'peak_clock_ghz': PEAK_CLOCK_GHZ,
                         'kernel': kinfo['name'], 'kernel_ms': kern_ms,
                         'algorithmic_flops_per_launch': alg_flops, 'flops_per_traj_step': flops_per_traj_step,
                         # executed view: `frac` is on ALGORITHMIC flops; the kernel shares common factors and executes fewer
                         'executed_fp64_instr_per_step': fp64_instr,
                         'executed_fp64_frac': (fp64_instr * 2.0 * n_traj * rk_steps / (kern_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS) if fp64_instr else None,
                         'executed': executed_view(kinfo['name'], launch_threads(kinfo['name'], n_traj), kern_ms, rk_steps),
                         'traffic': traffic,                                  # HBM bytes per launch from the PMC counters (profiles/)
                         'traffic_source': TRAFFIC_SOURCE,
                         'hbm_algorithmic_bytes_per_launch': alg_bytes,
                         'hbm_algorithmic_frac': alg_gbs / HBM_PEAK_GBS,     # SURVEY 8(d) byte figure / kernel time / 8 TB/s
                         'traffic_over_algorithmic': (traffic / alg_bytes) if traffic else None,
                         'note': 'the state stays in VGPRs for all steps of a launch, so HBM moves the initial and final states '
                                 'only; the kernel is bound by fp64 VALU issue'},
        }
        if rccl is not None:
            result['rccl'] = rccl
        if ref_elapsed:
            # the denominator of this line's scaling efficiency: what one GPU does with the same members-per-GPU when nothing is gathered
            per_gpu = float(n_traj) * rk_steps * ref_passes / ref_elapsed
            result['single_gpu_reference'] = {
                'value': per_gpu, 'unit': 'traj-steps/s per GPU', 'members_per_gpu': n_traj, 'passes': ref_passes,
                'note': 'the same passes on all %d GPUs at once without the gather (max over ranks); '
                        'scaling efficiency of this line = value / (n_gpus x this)' % world,
                'value_over_n_times_reference': value / (world * per_gpu)}
        if not result['roofline']['frac'] <= 1.0:
            print('bench.py: roofline fraction above 1 (%.3f): check the clock / flop count' % result['roofline']['frac'], file=sys.stderr)

        # ---- parity of what was timed: the output buffer of the last timed pass, sampled members against the CPU oracle ----
        failures = []
        try:
            from oracle.oracle import OracleModel
            idx = sample_members(n_traj, 16)
            last = engine.d_out[(args.steps - 1) % 2] if args.steps > 0 else engine.d_out[(args.warmup - 1) % 2]
            got = last[torch.from_numpy(idx).to(dev)].cpu().numpy()
            ref = OracleModel(ndim, coo, val).integrate_runge_kutta_jit(time_grid, ic_host[idx], 1, 0, b, c, a, threads=min(16, os.cpu_count() or 1))[:, :, 0]
            result['parity_check'] = parity_entry(kinfo['name'], 'output buffer of the last timed pass (final states after %d RK4 steps)' % rk_steps,
                                                  rel_err(got, ref), 1e-10, idx)
            result['parity_check']['rk_steps'] = rk_steps
            if not result['parity_check']['ok']:
                failures.append('headline')
        except Exception as e:                                               # an oracle that cannot be built is reported, not hidden
            result['parity_check'] = {'error': repr(e), 'ok': False}
            failures.append('headline (oracle unavailable)')
        if world == 1 and not args.no_extra_configs:
            try:
                result['configs'] = extra_configs(torch, dev, model, ndim, len(val), len(jval), (coo, val, jcoo, jval))
                for k, v in result['configs'].items():
                    entries = v.items() if k == 'f_rows' else [(k, v)]
                    failures += [kk for kk, vv in entries if not (isinstance(vv, dict) and vv.get('parity_check', {}).get('ok', False))]
                    for kk, vv in entries:                                   # fp64-bound entries also against what this box sustains on FMAs alone
                        rf = vv.get('roofline') if isinstance(vv, dict) else None
                        if fma_rate and isinstance(rf, dict) and rf.get('bound') == 'fp64_valu' and rf.get('achieved'):
                            rf['sustained_fma_tflops'] = fma_rate[0]
                            rf['frac_of_sustained_fma'] = rf['achieved'] / fma_rate[0]
            except Exception as e:                                           # never lose the headline line to a side measurement
                result['configs'] = {'error': repr(e)}
        if world == 1 and not args.no_cold_start:
            try:
                result['cold_start'] = cold_start()
                result['cold_start']['config3_maooam228'] = cold_start_228()
            except Exception as e:
                result['cold_start'] = {'error': repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            base, _, _ = cpu_baseline(ndim, coo, val, rk_steps, dt)
            result['cpu_baseline'] = base
        if failures:
            print('bench.py: PARITY FAILURE vs oracle in: %s' % ', '.join(failures), file=sys.stderr)
            result['parity_failures'] = failures
            result['value'] = 0.0
        print(json.dumps(result))
        sys.stdout.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
