"""Multi-GPU ensembles: one process per GPU, members sharded, one final gather.

The reference's only parallelism is one task per trajectory handed to `multiprocessing` workers through
pickling queues (qgs/integrators/integrator.py:133-142, 388-395).  Ensemble members never interact, so
here every rank (one process per GPU, `torch.distributed`, backend "nccl" = RCCL over xGMI) integrates a
contiguous block of members with the HIP engine and the only communication is the final gather of the
results.  Nothing in the data path needs a collective.

PyTorch is used for process-group plumbing and device buffers only.
"""
import numpy as np

from qgs_amd import _lib

# host <-> device transfers go through the library (its bounce blocks; plain torch copies for CPU tensors, e.g. the gloo tests)
_to_host, _to_device = _lib.to_host, _lib.to_device


def shard_bounds(n_total, world_size):
    """Contiguous blocks of members per rank, remainder to the first ranks: list of (start, stop)."""
    base, rem = divmod(int(n_total), int(world_size))
    out, start = [], 0
    for r in range(world_size):
        n = base + (1 if r < rem else 0)
        out.append((start, start + n))
        start += n
    return out


class ShardedEnsemble(object):
    """Bookkeeping of an ensemble of `n_total` members split over the ranks of a process group."""

    def __init__(self, n_total, process_group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = process_group
        self.distributed = bool(dist.is_available() and dist.is_initialized())
        if self.distributed:
            self.rank = dist.get_rank(process_group)
            self.world_size = dist.get_world_size(process_group)
        else:
            self.rank, self.world_size = 0, 1
        self.n_total = int(n_total)
        self.bounds = shard_bounds(self.n_total, self.world_size)
        self.counts = [b - a for a, b in self.bounds]

    @property
    def local_slice(self):
        a, b = self.bounds[self.rank]
        return slice(a, b)

    @property
    def n_local(self):
        return self.counts[self.rank]

    def gather(self, local):
        """All-gather the per-rank results (a torch tensor whose first axis is the local member axis) into the
        full ensemble, in member order, on every rank.  Equal shards take the single-buffer fast path
        (`all_gather_into_tensor`, one RCCL call); ragged shards are padded to the largest shard."""
        import torch
        dist = self._dist
        if not self.distributed:
            return local
        local = local.contiguous()                 # (a one-rank group still goes through the collective: plumbing check)
        if len(set(self.counts)) == 1:
            out = torch.empty((self.n_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            dist.all_gather_into_tensor(out, local, group=self.group)
            return out
        nmax = max(self.counts)
        padded = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded[:local.shape[0]] = local
        parts = [torch.empty_like(padded) for _ in range(self.world_size)]
        dist.all_gather(parts, padded, group=self.group)
        return torch.cat([p[:n] for p, n in zip(parts, self.counts)], dim=0)


def gather_to_host(ens, local, max_bytes=None):
    """The full (n_total, n_dim, n_records) result as a NumPy array on every rank, gathered from the per-rank blocks `local`
    (torch tensors, first axis = local members).  The gathered tensor lives where `local` lives; when the whole result would
    exceed `max_bytes` there (default: QGS_GATHER_BYTES or 32 GiB -- every rank of an all-gather holds the full result), it is
    gathered and copied to the host in chunks of records instead, so that no rank ever holds more than `max_bytes` of gathered
    data in device memory."""
    import os
    import torch
    if max_bytes is None:
        max_bytes = int(os.environ.get('QGS_GATHER_BYTES', str(32 << 30)))
    n_rec = int(local.shape[2])
    per_record = ens.n_total * int(local.shape[1]) * local.element_size()
    if not ens.distributed or per_record * n_rec <= max_bytes or n_rec <= 1:
        return _to_host(ens.gather(local))
    chunk = max(1, int(max_bytes // max(1, per_record)))
    out = np.empty((ens.n_total, int(local.shape[1]), n_rec))
    for r0 in range(0, n_rec, chunk):
        r1 = min(n_rec, r0 + chunk)
        out[:, :, r0:r1] = _to_host(ens.gather(local[:, :, r0:r1].contiguous()))
    return out


class RootGather(object):
    """Gather of equal per-rank blocks onto one rank, optionally asynchronous so that it overlaps the next
    compute step (RCCL runs it on its own stream).  xGMI is point-to-point: N-1 ranks sending their block
    straight to the root uses N-1 links in parallel, while a ring all-gather would push (N-1) blocks through
    every link -- so the benchmark's "final trajectory gather" is a gather, not an all-gather."""

    def __init__(self, ens, dst=0):
        if len(set(ens.counts)) != 1:
            raise ValueError('RootGather needs equal shards')
        self.ens, self.dst = ens, dst

    def start(self, local, out=None, async_op=True):
        """Returns (work handle or None, list of per-rank views of `out` on the root)."""
        import torch
        dist = self.ens._dist
        if not self.ens.distributed:
            return None, [local]
        parts = None
        if self.ens.rank == self.dst:
            if out is None:
                out = torch.empty((self.ens.n_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            parts = list(out.split(self.ens.counts[0], dim=0))
        work = dist.gather(local, gather_list=parts, dst=self.dst, group=self.ens.group, async_op=async_op)
        return work, parts


def _integrate_shard_on_device(f, device, local_ic, time, forward, write_steps, b, c, a):
    """One rank's block on its own GPU, results left in HBM: (n_local, n_dim, n_records) torch tensor on `device`.
    `qgs_rk_integrate_rows_device`: pack -> fused stepper -> record unpack on the device, the mode-major records held one
    window at a time (QGS_HIP_RECORD_WINDOW_MB), so the only whole-record buffer in HBM is the result itself."""
    import torch
    from qgs_amd import _lib
    model = f.hip_model(device=device.index)
    n, ndim = local_ic.shape
    nrec = _lib.n_records(time, write_steps)
    with torch.cuda.device(device):
        d_rows = _lib.to_device(local_ic, device)
        d_out = torch.empty((n, ndim, nrec), dtype=torch.float64, device=device)
        torch.cuda.current_stream(device).synchronize()          # the library works on its own streams
        model.rk_integrate_rows_device(n, d_rows.data_ptr(), time, 1 if forward else -1, write_steps, b, c, a, d_out.data_ptr())
    return d_out


def integrate_ensemble(f, t0, t, dt, ic, forward=True, write_steps=0, b=None, c=None, a=None, process_group=None,
                       device=None, integrator_factory=None):
    """Integrate a (n_traj, n_dim) ensemble sharded over the ranks of `process_group`; every rank returns the
    full ``(time, traj)`` with traj of shape (n_traj, n_dim, n_records) (not squeezed).

    On GPUs (backend "nccl", or no process group at all) each rank integrates `ic[shard]` on ITS OWN device --
    `device`, default the current CUDA device, i.e. LOCAL_RANK after `torch.cuda.set_device` -- and the results stay
    in HBM until they have been gathered: H2D of the shard, pack, fused stepper, record unpack, one RCCL all-gather,
    one D2H.  Tendencies that are a plain Python callable, and process groups on a CPU backend (gloo), take the host
    route: each rank runs `RungeKuttaIntegrator` on its block and the NumPy results are gathered through host tensors.
    `integrator_factory` (a callable returning an object with the `RungeKuttaIntegrator` interface) forces that route
    with another engine (test seam).
    """
    import torch
    from qgs_amd.integrators.integrate import record_times, resolve_tableau, time_grid
    ic = np.asarray(ic, dtype=np.float64)
    if ic.ndim == 1:
        ic = ic.reshape((1, -1))
    ens = ShardedEnsemble(ic.shape[0], process_group)
    local_ic = np.ascontiguousarray(ic[ens.local_slice])
    grid = time_grid(t0, t, dt)
    time = record_times(grid, write_steps, forward)
    backend = ens._dist.get_backend(process_group) if ens.distributed else None
    on_gpu = getattr(f, 'hip_model', None) is not None and backend in (None, 'nccl')
    if integrator_factory is None and on_gpu:
        # ---- device-resident route ----
        from qgs_amd import _lib
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        device = torch.device(device)
        if device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        b, c, a = resolve_tableau(b, c, a)
        nrec = _lib.n_records(grid, write_steps)
        if ens.n_local > 0:
            local = _integrate_shard_on_device(f, device, local_ic, grid, forward, write_steps, b, c, a)
        else:
            local = torch.zeros((0, ic.shape[1], nrec), dtype=torch.float64, device=device)
        return time, gather_to_host(ens, local)
    if integrator_factory is None:
        # a user-written Python callable (host stepper), or a CPU process group (gloo): every rank integrates its block through
        # the integrator class and the NumPy results are gathered through host tensors
        from qgs_amd.integrators.integrator import RungeKuttaIntegrator
        integrator_factory = RungeKuttaIntegrator
        if backend is not None and backend != 'nccl':
            device = torch.device('cpu')
    # ---- host route (test seam) ----
    # the rank's own GPU, explicitly: an integrator left at device=None may spread a large block over every visible GPU
    # (integrate.resolve_device), and those belong to the other ranks
    local_gpu = None
    if getattr(f, 'hip_model', None) is not None and ens.distributed and torch.cuda.is_available():
        local_gpu = torch.cuda.current_device() if device is None or torch.device(device).type != 'cuda' or torch.device(device).index is None \
            else torch.device(device).index
    kwargs = {}
    if local_gpu is not None:
        import inspect
        try:
            params = inspect.signature(integrator_factory).parameters
            if 'device' in params or any(q.kind == q.VAR_KEYWORD for q in params.values()):
                kwargs['device'] = local_gpu
        except (TypeError, ValueError):                   # a factory whose signature cannot be read keeps its own default
            pass
    integ = integrator_factory(b=b, c=c, a=a, **kwargs)
    integ.set_func(f)
    if ens.n_local > 0:
        integ.integrate(t0, t, dt, ic=local_ic, forward=forward, write_steps=write_steps)
        integ.get_trajectories()
        local = np.asarray(integ._recorded_traj)
    else:
        local = np.zeros((0, ic.shape[1], 1))
    integ.terminate()
    if not ens.distributed:
        return time, local
    if device is None:
        backend = ens._dist.get_backend(process_group)
        device = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')
    # ranks with an empty shard still need the record count for the gather buffer
    nrec = torch.tensor([local.shape[2] if ens.n_local > 0 else 0], dtype=torch.int64, device=device)
    ens._dist.all_reduce(nrec, op=ens._dist.ReduceOp.MAX, group=process_group)
    if ens.n_local == 0:
        local = np.zeros((0, ic.shape[1], int(nrec.item())))
    return time, gather_to_host(ens, _to_device(local, device))
