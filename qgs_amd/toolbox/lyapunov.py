"""Lyapunov exponents and backward / forward Lyapunov vectors with the Benettin algorithm
(reference: qgs/toolbox/lyapunov.py, class LyapunovsEstimator and the loops at :471-632).

Same API as the reference (`set_func(f, fjac)`, `compute_lyapunovs(t0, tw, t, dt, mdt, ...)`,
`get_lyapunovs()`), whole ensemble on the GPU:

* the base trajectory is integrated by the fused RK stepper with every step recorded;
* per `dt` interval the tangent model is integrated over the `mdt` sub-steps by the TGLS kernels.  The
  reference propagates the identity and multiplies (`prop @ q`, lyapunov.py:546, 624); the tangent model is
  linear, so the `n_vec` columns of `q` are propagated directly (same result to rounding, n_dim/n_vec
  times less work);
* the re-orthonormalisation `np.linalg.qr` (lyapunov.py:547, 625) is the batched Householder QR kernel
  `qgs_batched_qr_device` (LAPACK sign convention, so the vectors match the reference's);
* exponents `log|diag R| / dt` are formed on the host at the end from the recorded diagonals.

What the device holds is bounded (round 4).  The reference keeps one trajectory's records at a time in host memory
(lyapunov.py:232-358, 555-632): the size of a run is limited by the host, not by a device.  Here, within the budget of
QGS_HIP_RECORD_WINDOW_MB (default 8192, the budget of the windowed integrations of qgs_hip_api.hip):

* the base trajectory is held one WINDOW of steps at a time (`_BaseTrajectory`): backward vectors consume it in time order
  and integrate it window by window; forward vectors consume it against time and recompute each window from a stored
  checkpoint (its first state), with the same kernel and the same time grid values, hence bitwise the same states;
* the recorded vectors / states / diag(R) go into one of two windows of records (`_RecordWindows`); a full window leaves
  for the host block on a copy stream (`qgs_unpack_window`: the kernel's own stores into the page-locked result block, in
  the reference's layout) while the next intervals run into the other window.
A run whose records fit the budget is one window of each: the same code path.  The result blocks themselves are host
memory, allocated for the whole ensemble before anything runs; a run that does not fit the host says so ("host memory").

The random initial basis is drawn exactly like the reference does (one `np.random.random((n_dim, n_vec))`
per trajectory, in trajectory order), so seeded runs are reproducible against it.
PyTorch is used for device buffers, streams and events only.
"""
import multiprocessing
import os
import threading

import numpy as np

from qgs_amd import _lib
from qgs_amd.integrators import integrate as _fn
from qgs_amd.functions.util import normalize_matrix_columns as _normalize_columns, reverse, solve_triangular_matrix as _solve_triangular


def _window_budget_bytes():
    """Device memory the estimator spends on its windows (same knob and default as the windowed integrations)."""
    try:
        mb = float(os.environ.get('QGS_HIP_RECORD_WINDOW_MB', '8192'))
    except ValueError:
        mb = 8192.0
    return int(max(1.0, mb * 1048576.0))


def _stat_sum(stat_path, keys):
    """Sum of the named counters of a cgroup memory.stat file (0 when it cannot be read)."""
    total = 0
    try:
        with open(stat_path) as f:
            for line in f:
                parts = line.split()
                if len(parts) == 2 and parts[0] in keys:
                    total += int(parts[1])
    except (OSError, ValueError):
        return 0
    return total


def _cgroup_memory_room(proc_cgroup='/proc/self/cgroup', sys_root='/sys/fs/cgroup'):
    """Bytes this process' control groups still allow (limit - usage, the tightest of the groups found), or None when no
    group sets a limit.  /proc/meminfo describes the machine, not the container: a container limit is only visible here.
    Usage is counted without the group's file cache (memory.stat: active_file + inactive_file): after reading goldens, the
    kernel cache or earlier large results a container reports little room while nearly all of it can be had back."""
    rooms = []

    def read(path):
        try:
            with open(path) as f:
                v = f.read().strip()
            return None if v in ('', 'max') else int(v)
        except (OSError, ValueError):
            return None
    paths = {'': ''}
    try:
        with open(proc_cgroup) as f:
            for line in f:
                parts = line.strip().split(':', 2)
                if len(parts) == 3:
                    if parts[0] == '0' and parts[1] == '':
                        paths['v2'] = parts[2]
                    elif 'memory' in parts[1].split(','):
                        paths['v1'] = parts[2]
    except OSError:
        pass
    candidates = []
    for root, sub, lim, cur in ((sys_root, paths.get('v2', ''), 'memory.max', 'memory.current'),
                                (os.path.join(sys_root, 'unified'), paths.get('v2', ''), 'memory.max', 'memory.current'),
                                (os.path.join(sys_root, 'memory'), paths.get('v1', ''), 'memory.limit_in_bytes', 'memory.usage_in_bytes')):
        # the group itself and every ancestor up to the mount point (a limit anywhere on the way applies)
        sub = sub.strip('/')
        chain = [''] + ['/'.join(sub.split('/')[:k]) for k in range(1, len(sub.split('/')) + 1)] if sub else ['']
        for c in chain:
            d = os.path.join(root, c) if c else root
            candidates.append((os.path.join(d, lim), os.path.join(d, cur)))
    for lim_path, cur_path in candidates:
        limit = read(lim_path)
        if limit is None or limit >= (1 << 60):                   # "max" / the v1 spelling of no limit
            continue
        used = read(cur_path) or 0
        stat = os.path.join(os.path.dirname(cur_path), 'memory.stat')
        # (v1 lists both `inactive_file` and `total_inactive_file`: the hierarchical totals are the ones that match usage_in_bytes)
        if os.path.basename(cur_path) == 'memory.usage_in_bytes':
            cache = _stat_sum(stat, ('total_inactive_file', 'total_active_file')) or _stat_sum(stat, ('inactive_file', 'active_file'))
        else:
            cache = _stat_sum(stat, ('inactive_file', 'active_file'))
        rooms.append(max(0, limit - max(0, used - cache)))
    return min(rooms) if rooms else None


def _host_memory_available():
    """Host memory a result can still take: MemAvailable of /proc/meminfo, cut down to what the process' control groups
    allow; None when neither can be read."""
    avail = None
    try:
        with open('/proc/meminfo') as f:
            for line in f:
                if line.startswith('MemAvailable:'):
                    avail = int(line.split()[1]) * 1024
                    break
    except (OSError, ValueError, IndexError):
        pass
    room = _cgroup_memory_room()
    if room is not None:
        avail = room if avail is None else min(avail, room)
    return avail


class _BaseTrajectory(object):
    """States of the base trajectory on the device, a window of steps at a time.

    `state(i)` returns the (n_dim, ld) device view of the state at grid index i; the indices must come in monotone order
    (ascending, or descending with `descending=True`).  Window k holds the states [k W, min((k + 1) W, G - 1)]; it is the
    record of one `rk_integrate_device` launch over that piece of the grid with write_steps = 1, started from the last state
    of window k - 1.  A run cut into ranges is bitwise the run in one piece (the steppers take dt from the grid values).
    Descending: a first sweep integrates window after window and keeps the first state of each (checkpoints); the windows
    are then recomputed from their checkpoints in reverse order -- the last one is still resident after the sweep."""

    def __init__(self, torch, m, n, ld, grid, ic_modes, budget, descending, tableau, stream):
        self.torch, self.m, self.n, self.ld, self.grid, self.stream = torch, m, n, ld, grid, stream
        self.b, self.c, self.a = tableau
        ndim = ic_modes.shape[0]
        steps = len(grid) - 1
        state_bytes = ndim * ld * 8
        self.W = int(max(1, min(max(1, steps), budget // state_bytes - 1)))
        self.n_windows = max(1, -(-steps // self.W))
        self.buf = torch.empty((min(self.W, steps) + 1, ndim, ld), dtype=torch.float64, device=ic_modes.device)
        self.starts = [ic_modes]                 # first state of window k (descending: all of them; ascending: the current one)
        self.k = -1
        self.descending = descending
        self._load(0, ic_modes)
        if descending:
            for k in range(1, self.n_windows):
                start = self.buf[self.W].clone()
                self.starts.append(start)
                self._load(k, start)

    def _bounds(self, k):
        return k * self.W, min((k + 1) * self.W, len(self.grid) - 1)

    def _load(self, k, start):
        lo, hi = self._bounds(k)
        if hi > lo:
            self.m.rk_integrate_device(self.n, self.ld, start.data_ptr(), self.grid[lo:hi + 1], 1, 1, self.b, self.c, self.a,
                                       self.buf.data_ptr(), self.stream)
        else:                                    # a grid of one point: the state is the initial condition
            self.buf[0].copy_(start)
        self.k = k

    def state(self, i):
        if i < 0 or i > len(self.grid) - 1:
            raise IndexError('base trajectory: grid index out of range')
        lo, hi = self._bounds(self.k)
        # a consumer may skip several windows between two calls (the Ginelli / CLV runs ask for one state every dt / mdt
        # steps, and a window can be shorter than that): move window by window until i is resident
        while i > hi or i < lo:
            if self.descending:
                if i > hi:
                    raise RuntimeError('base trajectory states must be consumed in monotone (descending) order')
                self._load(self.k - 1, self.starts[self.k - 1])
            else:
                if i < lo:
                    raise RuntimeError('base trajectory states must be consumed in monotone (ascending) order')
                start = self.buf[hi - lo].clone()             # the last state of the window that is about to be overwritten
                self._load(self.k + 1, start)
            lo, hi = self._bounds(self.k)
        return self.buf[i - lo]


class _RecordWindows(object):
    """The records of a run (vectors, states, diag(R)) on their way to the host blocks, a window of records at a time.

    `slot(iw)` returns the device views record iw is to be written into (on the compute stream); record indices must come in
    monotone order.  When a record falls outside the current window, that window is handed to the copy stream
    (`qgs_unpack_window`: mode-major window -> records [first, first + count) of the host blocks in the reference's layout) and
    the other buffer becomes current -- after its own previous drain has completed."""

    def __init__(self, torch, m, n, ld, inner, hosts, n_records, budget, dev):
        self.torch, self.m, self.n, self.ld, self.inner, self.hosts, self.n_records = torch, m, n, ld, inner, hosts, n_records
        per_record = sum(inner) * ld * 8
        self.W = int(max(1, min(n_records, budget // (2 * per_record))))
        self.n_windows = -(-n_records // self.W)
        nbuf = 1 if self.n_windows == 1 else 2
        self.bufs = [[torch.empty((self.W, q, ld), dtype=torch.float64, device=dev) for q in inner] for _ in range(nbuf)]
        self.done = [None] * nbuf                             # event: the buffer's last drain has completed
        self.compute = torch.cuda.current_stream(dev)
        self.copy = torch.cuda.Stream(dev) if nbuf > 1 else self.compute
        self.j, self.which, self.used = None, 0, None
        self.flushed = 0

    def slot(self, iw):
        j = iw // self.W
        if j != self.j:
            if self.j is not None:
                self._flush()
                self.which = (self.which + 1) % len(self.bufs)
            if self.done[self.which] is not None:
                self.compute.wait_event(self.done[self.which])
            self.j, self.used = j, None
        self.used = (iw, iw) if self.used is None else (min(self.used[0], iw), max(self.used[1], iw))
        return [t[iw - j * self.W] for t in self.bufs[self.which]]

    def _flush(self):
        if self.used is None:
            return
        torch = self.torch
        first, count = self.used[0], self.used[1] - self.used[0] + 1
        if self.copy is not self.compute:
            ready = torch.cuda.Event()
            ready.record(self.compute)
            self.copy.wait_event(ready)
        for t, q, host in zip(self.bufs[self.which], self.inner, self.hosts):
            # (a pageable host block: staged on the device, then brought over by the library's drain thread while the next
            # window is computed -- `finish` waits for it)
            self.m.unpack_window_enqueue(self.n, self.ld, q, count, self.n_records, first, t[first - self.j * self.W].data_ptr(),
                                         host.ctypes.data, self.copy.cuda_stream)
        ev = torch.cuda.Event()
        ev.record(self.copy)
        self.done[self.which] = ev
        self.used = None
        self.flushed += 1

    def abort(self):
        """After an exception inside a run: wait for every stream of the device and for the drain thread (a secondary error is
        swallowed -- the first one is what the caller sees)."""
        for wait in (lambda: self.torch.cuda.synchronize(self.compute.device), self.m.drain_wait):
            try:
                wait()
            except Exception:
                pass

    def finish(self, wait=True):
        """Hand the last window over; `wait=False`: the caller waits for the drain itself (`HipModel.drain_wait`), after it has
        enqueued more work -- the member groups of `LyapunovsEstimator._compute_shard`."""
        self._flush()
        if wait:
            self.copy.synchronize()
            self.compute.synchronize()
            self.m.drain_wait()


def _in_threads(work, count):
    """work(i) for i in range(count): on the calling thread when there is one piece, else one host thread per piece (the work
    of a shard is a chain of kernel launches on its own GPU); the first exception is re-raised on the calling thread."""
    if count == 1:
        work(0)
        return
    errors = []

    def run(i):
        try:
            work(i)
        except Exception as e:
            errors.append(e)
    threads = [threading.Thread(target=run, args=(i,)) for i in range(count)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    if errors:
        raise errors[0]


class LyapunovsEstimator(object):
    """Estimate the Lyapunov exponents and the Backward (default) or Forward Lyapunov Vectors.

    ``LyapunovsEstimator(num_threads=None, b=None, c=None, a=None, number_of_dimensions=None)``; attributes
    ``num_threads, b, c, a, n_dim, n_vec, n_traj, n_records, ic, func, func_jac`` as in the reference.
    """

    def __init__(self, num_threads=None, b=None, c=None, a=None, number_of_dimensions=None, device=None):
        self.num_threads = multiprocessing.cpu_count() if num_threads is None else num_threads
        # GPU(s): an index, a list of indices or 'all' (members sharded over them, one host thread per shard), None = the
        # device the tendencies were created for (as for the integrator classes, qgs_amd/integrators/integrator.py)
        self.device = device
        self.b, self.c, self.a = _fn.resolve_tableau(b, c, a)
        self.ic = None
        self._time = None
        self._pretime = None
        self._recorded_traj = None
        self._recorded_exp = None
        self._recorded_vec = None
        self.n_traj = 0
        self.n_dim = number_of_dimensions
        self.n_records = 0
        self.n_vec = 0
        self.write_steps = 0
        self._adjoint = False
        self._forward = -1
        self._inverse = 1.
        self.func = None
        self.func_jac = None
        self._model = None
        self._recorded_pre = None         # (_run(pre_qr=True)) the propagated matrices before their QR, per record
        self._recorded_r = None           # (host path) the R of the QR step that follows each record
        self._junction = None             # (_run(junction=True)) the states at the first recorded time
        self._fine_base = False
        self.last_windows = None          # (base-trajectory windows, record windows) of the last run, per shard

    def terminate(self):
        self._model = None

    def start(self):
        self.terminate()
        if self.func is not None and _fn.on_device(self.func):       # (a user-written callable has no model: host_lyapunov.py)
            self._model = _fn.hip_model_of(self.func, device=_fn.resolve_device(self.device))
            if self.func_jac is not None and _fn.hip_model_of(self.func_jac, 'fjac', device=_fn.resolve_device(self.device)) is not self._model:
                raise TypeError('f and fjac must come from the same create_tendencies() call')

    def set_bca(self, b=None, c=None, a=None, ic_init=True):
        if a is not None:
            self.a = a
        if b is not None:
            self.b = b
        if c is not None:
            self.c = c
        if ic_init:
            self.ic = None
        self.start()

    def set_func(self, f, fjac):
        self.func = f
        self.func_jac = fjac
        self.start()

    # ------------------------------------------------------------------------------------------------
    def compute_lyapunovs(self, t0, tw, t, dt, mdt, ic=None, write_steps=1, n_vec=None, forward=False, adjoint=False,
                          inverse=False):
        """Benettin algorithm.  Backward vectors (`forward=False`): QR-propagate a random basis from `t0` to `tw`
        (spin-up), then record from `tw` to `t`.  Forward vectors (`forward=True`): propagate backward in time from
        `t` to `tw`, then record from `tw` back to `t0`.  `dt` is the re-orthonormalisation interval, `mdt` the
        integration time step inside it; `n_vec` the number of vectors (default all)."""
        if self.func is None or self.func_jac is None:
            print('No function to integrate defined!')
            return 0
        if self._model is None and _fn.on_device(self.func):
            self.start()
        ic = np.zeros(_fn.dimension_of(self.func)) if ic is None else ic
        self._run(_fn.time_grid(t0, tw, dt), _fn.time_grid(tw, t, dt), mdt, ic, write_steps, n_vec, forward, adjoint, inverse)

    def _run(self, pretime, time, mdt, ic, write_steps, n_vec, forward, adjoint, inverse, a0=None, pre_qr=False, junction=False,
             fine_base=False):
        """The Benettin run on explicit grids (`compute_lyapunovs` builds them from t0, tw, t, dt; the covariant estimator below
        hands over its own).  `a0`: the (n_traj, n_dim, n_vec) start matrices instead of the reference's uniform draws.
        `pre_qr`: also keep, per record, the propagated matrix BEFORE its QR step (`_recorded_pre`; R of that step is
        Q(next record)^T times it).  `junction`: also keep the state at the first recorded time (`_junction`).  `fine_base`:
        the base trajectory is integrated with the `mdt` sub-steps of the intervals instead of one step per interval (the
        reference's Ginelli loop advances its state with the tangent integrator itself, lyapunov.py:1203-1247)."""
        self._fine_base = fine_base
        self.ic = ic
        if len(self.ic.shape) == 1:
            self.ic = self.ic.reshape((1, -1))
        self.n_traj, self.n_dim = self.ic.shape
        self.n_vec = self.n_dim if n_vec is None else n_vec
        self._pretime = pretime
        self._time = time
        self.write_steps = write_steps
        self._forward = 1 if forward else -1
        self._adjoint = adjoint
        self._inverse = -1. if inverse else 1.
        rec_grid = self._pretime if forward else self._time
        if write_steps == 0:
            self.n_records = 1
        else:
            tot = rec_grid[::write_steps]
            self.n_records = len(tot) + (1 if tot[-1] != rec_grid[-1] else 0)

        if not (_fn.on_device(self.func) and _fn.on_device(self.func_jac)):
            # user-written Python callables: the same loops in NumPy on the host steppers (host_lyapunov.py)
            from qgs_amd.toolbox import host_lyapunov
            if a0 is None:
                a0 = np.random.random((self.n_traj, self.n_dim, self.n_vec))
            res = host_lyapunov.benettin(self.func, self.func_jac, pretime, time, mdt, self.ic, self.n_vec, write_steps, forward,
                                         adjoint, self._inverse, (self.b, self.c, self.a), a0, fine_base=fine_base)
            self._recorded_traj, self._recorded_vec, self._recorded_exp = res['traj'], res['vec'], res['exp']
            self._recorded_pre, self._recorded_r, self._junction = None, res['r'], res['junction']
            self.last_windows = None
            return

        # the result blocks of the WHOLE ensemble, in host memory and in the reference's layouts; every shard fills its slice.
        # Their size is what bounds a run -- checked before anything is allocated or computed.
        nt, nd, nv, nr = self.n_traj, self.n_dim, self.n_vec, self.n_records
        need = 8 * nt * nr * (nd * nv * (2 if pre_qr else 1) + nd + nv)
        avail = _host_memory_available()
        if avail is not None and need > 0.8 * avail:
            raise MemoryError('host memory: the records of this run (%d members x %d records x (%d x %d vectors + state + exponents)) '
                              'need %.1f GB, %.1f GB are available -- raise write_steps, or lower n_vec or the number of members'
                              % (nt, nr, nd, nv, need / 1e9, avail / 1e9))
        model = _fn.hip_model_of(self.func, device=_fn.resolve_device(self.device, self.n_traj))
        shards = getattr(model, 'models', None)
        try:
            pin = need <= _lib._RESULTS.cap          # (a record beyond what the pool keeps travels through the bounce ring: no page-locking)
            # (member groups fill their pieces of the blocks front to back: no huge pages, `_lib._Store`)
            import torch
            huge = shards is not None or self._member_groups(torch, model, nt, pre_qr)[0] >= nt
            out_traj = _lib._RESULTS.empty((nt, nd, nr), pin, huge)
            out_vec = _lib._RESULTS.empty((nt, nd, nv, nr), pin, huge)
            out_exp = _lib._RESULTS.empty((nt, nv, nr), pin, huge)
            out_pre = _lib._RESULTS.empty((nt, nd, nv, nr), pin, huge) if pre_qr else None
        except MemoryError:
            raise MemoryError('host memory: could not allocate %.1f GB for the records of this run' % (need / 1e9))

        # random start bases: the matrices are drawn like the reference's (one draw per trajectory, in order:
        # `np.random.random((ndim, nv))` consumes the generator exactly as n such calls in a row do) -- for the WHOLE ensemble
        # before it is split over devices
        if a0 is None and shards is not None:
            a0 = np.random.random((self.n_traj, self.n_dim, self.n_vec))        # (one GPU: drawn by `_compute_shard`, possibly in pieces)
        self._junction = np.zeros((nt, nd)) if junction else None
        outs = (out_traj, out_vec, out_exp) + ((out_pre,) if pre_qr else ())

        def piece(a, cnt):
            return tuple(o[a:a + cnt] for o in outs) + ((self._junction[a:a + cnt],) if junction else (None,))
        if shards is None:
            self.last_windows = [self._compute_shard(model, self.ic, a0, mdt, piece(0, nt))]
        else:
            # several GPUs: contiguous member shards, one host thread each (the work of a shard is a chain of kernel launches)
            windows = [None] * len(shards)

            def run(i):
                a, cnt = model.shard(self.n_traj, i)
                if cnt > 0:
                    windows[i] = self._compute_shard(shards[i], self.ic[a:a + cnt], a0[a:a + cnt], mdt, piece(a, cnt))
            _in_threads(run, len(shards))
            self.last_windows = [w for w in windows if w is not None]
        self._recorded_traj, self._recorded_vec, self._recorded_exp, self._recorded_pre = out_traj, out_vec, out_exp, out_pre
        self._recorded_r = None

    def _compute_shard(self, m, ic, a0, mdt, outs):
        """The Benettin loops for the members `ic` (n, n_dim) with start matrices `a0` (n, n_dim, n_vec) on model `m`'s GPU;
        fills the slices `outs` = (traj, vectors, exponents[, matrices before the QR], junction state or None) of the result
        blocks (reference layouts).

        Records that do not fit the device window in one piece leave in MEMBER GROUPS where the ensemble is large enough: the
        loops run group after group, each group's whole record is one window, and since the result blocks are member-major
        (n_traj, ..., n_records) a group's window is ONE contiguous piece of each of them -- the drain thread streams it while
        the next group is computed.  Windows of records (the route for few members, or a budget set by hand) reach every page of
        the result blocks once per window in runs of 8 W bytes, which is what held the 72 GB run at 25 - 30 GB/s
        (profiles/r05_lyap_big.md).  Members are independent: a group's results are those of a run of just these members (as for
        the shards of a device list; kernels are chosen by ensemble size, so against one pass over all members they agree to
        rounding, not bitwise)."""
        import torch
        with torch.cuda.device(torch.device('cuda', m.device)):       # this thread's current device for the duration of the call only
            n = ic.shape[0]
            group, budget = self._member_groups(torch, m, n, len(outs) > 4)
            if group >= n:
                if a0 is None:
                    a0 = np.random.random((n, self.n_dim, self.n_vec))
                return self._compute_shard_on_current_device(m, ic, a0, mdt, outs)
            info = []
            try:
                for lo in range(0, n, group):
                    cnt = min(group, n - lo)
                    part = tuple(None if o is None else o[lo:lo + cnt] for o in outs)
                    # (start matrices not drawn yet -- `_run` on one GPU: drawn group by group, the same numbers in the same order,
                    # while the previous group's record is on its way)
                    a0g = a0[lo:lo + cnt] if a0 is not None else np.random.random((cnt, self.n_dim, self.n_vec))
                    info.append(self._compute_shard_on_current_device(m, ic[lo:lo + cnt], a0g, mdt, part, budget, True))
            finally:
                # every stream of the device (a group cut into record windows copies on a stream of its own), then the drain thread:
                # nothing is on its way into the result blocks when they are handed out
                torch.cuda.synchronize(torch.device('cuda', m.device))
                m.drain_wait()
            return max(i[0] for i in info), max(i[1] for i in info)

    def _member_groups(self, torch, m, n, pre_qr):
        """(members per group, record-window budget of a group); a group of `n` members or more: one pass, the default budget.
        `QGS_HIP_RECORD_GROUP_MEMBERS` sets the group size by hand (tests); a record-window budget set by hand
        (`QGS_HIP_RECORD_WINDOW_MB`) keeps the windows of records."""
        nd, nv, nr = self.n_dim, self.n_vec, self.n_records
        per_member = 8 * nr * (nd * nv * (2 if pre_qr else 1) + nd + nv)
        forced = os.environ.get('QGS_HIP_RECORD_GROUP_MEMBERS')
        if forced:
            g = max(64, (int(forced) + 63) // 64 * 64)
            return g, 2 * per_member * g + (1 << 20)
        if 'QGS_HIP_RECORD_WINDOW_MB' in os.environ:
            return n, None
        free, _total = torch.cuda.mem_get_info(torch.device('cuda', m.device))
        one_window = max(_window_budget_bytes() // 2, min(free // 16, 12 << 30)) // 2       # what one window of records may hold
        if per_member * n <= one_window:
            return n, None
        # a group's window + its staging block, two groups in flight: within a third of the free memory; eight groups or more where
        # that leaves 2 048 members per group (the tangent kernels fill the GPU from there), never fewer than 1 024
        cap = (free // 3) // (4 * per_member) // 64 * 64
        g = min(cap, max(2048, ((n + 7) // 8 + 63) // 64 * 64))
        if g < 1024 or g >= n:
            return n, None
        return int(g), 2 * per_member * int(g) + (1 << 20)

    def _compute_shard_on_current_device(self, m, ic, a0, mdt, outs, rec_budget=None, grouped=False):
        import torch
        forward, adjoint, write_steps = self._forward == 1, self._adjoint, self.write_steps
        ndim, nv, n = self.n_dim, self.n_vec, ic.shape[0]
        ld = (n + 63) // 64 * 64
        dev = torch.device('cuda', m.device)
        f64 = torch.float64
        stream = torch.cuda.current_stream(dev).cuda_stream
        out_traj, out_vec, out_exp = outs[:3]
        out_pre = outs[3] if len(outs) > 4 else None
        out_junction = outs[-1]
        budget = _window_budget_bytes()

        # base trajectory, every step recorded: R[step][mode][member], a window of steps at a time   (lyapunov.py:558 / :474)
        full_grid = np.concatenate((self._pretime[:-1], self._time))
        at = np.arange(len(full_grid))                              # interval boundary -> index in the base trajectory's grid
        if self._fine_base:
            pieces = []
            for i in range(len(full_grid) - 1):
                tt, d = full_grid[i], full_grid[i + 1] - full_grid[i]
                pieces.append(np.arange(tt, tt + d, mdt))            # (the sub-steps `propagate` is given, without the end point)
                at[i + 1] = at[i] + len(pieces[-1])
            full_grid = np.concatenate(pieces + [full_grid[-1:]])
        ic_modes = torch.zeros((ndim, ld), dtype=f64, device=dev)
        # (host -> device through the library's bounce blocks: `_lib.to_device`; no pageable pointer ever reaches the runtime, which
        # is what made concurrent uploads of shard threads unsafe in round 4)
        ic_modes[:, :n] = _lib.to_device(ic.T, dev)
        base = _BaseTrajectory(torch, m, n, ld, full_grid, ic_modes, budget // 4, forward, (self.b, self.c, self.a), stream)
        n_pre = len(self._pretime)

        # orthonormal start basis: QR of the drawn matrices with the same batched Householder kernel as in the loop
        # (np.linalg.qr on the host took 30 us per member: 0.5 s at 16 384)
        # (the drawn matrices go up as they are, (n, n_dim, n_vec), and are brought into the device layout F[mode][vector][member]
        # by the pack kernel: the host-side transpose of 170 MB at config-4 size took longer than the whole spin-up)
        q = torch.zeros((ndim, nv, ld), dtype=f64, device=dev)
        a0_rows = _lib.to_device(a0, dev)
        m.pack_tangent(n, ld, nv, a0_rows.data_ptr(), q.data_ptr(), stream)
        torch.cuda.current_stream(dev).synchronize()
        del a0_rows
        # diag(R) of that first QR: with an empty spin-up the reference's `r = qr[1]` is still this one
        # (lyapunov.py:524, 603), so the first recorded exponents come from it
        rdiag0 = torch.ones((nv, ld), dtype=f64, device=dev)
        m.batched_qr_device(n, ld, ndim, nv, q.data_ptr(), rdiag0.data_ptr(), stream)
        if ld > n:
            rdiag0[:, n:] = 1.0                                      # padding columns (log |r| of them is never used)

        q_new = torch.empty((1, ndim, nv, ld), dtype=f64, device=dev)
        y_end = torch.empty((1, ndim, ld), dtype=f64, device=dev)
        # records on their way to the host: vectors F[record][mode * vector][member], states, diag(R) of the QR before the interval
        # Their windows want to be long: a window of W records reaches the host block (records innermost) as runs of 8 W bytes, and
        # at config-4 size half of the default budget gives W = 11 -- 88-byte runs, a page apart.  Unless the budget was set by hand,
        # the record windows take up to a sixteenth of the GPU's free memory, at most 12 GiB (W = 36 there; measured over budgets
        # of 8 / 16 / 32 / 64 GiB in profiles/r05_lyap_big.md: longer windows cost more start-up and tail than their runs gain).
        if rec_budget is None:
            rec_budget = budget // 2
            if 'QGS_HIP_RECORD_WINDOW_MB' not in os.environ:
                free, _total = torch.cuda.mem_get_info(dev)
                rec_budget = max(rec_budget, min(free // 16, 12 << 30))
        rec = _RecordWindows(torch, m, n, ld, (ndim * nv, ndim, nv) + ((ndim * nv,) if out_pre is not None else ()),
                             (out_vec, out_traj, out_exp) + ((out_pre,) if out_pre is not None else ()), self.n_records, rec_budget, dev)

        def propagate(y_index, subtime, direction, pre=None):
            """q <- Q of QR( TL_{subtime}(q) ) along the trajectory started at base[y_index]; returns diag(R).
            `pre`: device view that receives TL_{subtime}(q) as it is before the QR."""
            nonlocal q, q_new
            m.rk_tgls_integrate_device(n, ld, nv, base.state(at[y_index]).data_ptr(), q.data_ptr(), subtime, direction, 0,
                                       self.b, self.c, self.a, adjoint, self._inverse, y_end.data_ptr(), q_new.data_ptr(),
                                       stream)
            if pre is not None:
                pre.copy_(q_new[0].reshape(ndim * nv, ld))
            rdiag = torch.empty((nv, ld), dtype=f64, device=dev)
            m.batched_qr_device(n, ld, ndim, nv, q_new.data_ptr(), rdiag.data_ptr(), stream)
            q, q_new = q_new[0], q.unsqueeze(0)
            return rdiag

        def record(iw, y_index, rdiag, d):
            """record iw <- (q, base[y_index], rdiag); exponents = log|rdiag| / d (None: no interval behind it -> zeros)"""
            slots = rec.slot(iw)
            s_vec, s_traj, s_rd = slots[:3]
            s_traj.copy_(base.state(at[y_index]))
            s_vec.copy_(q.reshape(ndim * nv, ld))
            if rdiag is None:
                s_rd.zero_()
            else:
                # log|diag R| / dt on the device (lyapunov.py:531, 611): the exponents leave in the window as they are
                m.local_exponents_device(nv * ld, rdiag.data_ptr(), d, s_rd.data_ptr(), stream)
            if out_pre is not None:
                slots[3].zero_()                   # (stays zero for a record no interval follows)
                return slots[3]
            return None

        try:
            if not forward:
                # ---- backward Lyapunov vectors (lyapunov.py:564-632) ----
                pre, tim = self._pretime, self._time
                rdiag = None
                for ti in range(len(pre) - 1):
                    tt, d = pre[ti], pre[ti + 1] - pre[ti]
                    sub = np.concatenate((np.arange(tt, tt + d, mdt), np.full((1,), tt + d)))
                    rdiag = propagate(ti, sub, 1)
                if rdiag is None:       # no spin-up interval
                    rdiag = rdiag0
                if out_junction is not None:
                    out_junction[...] = _lib.to_host(base.state(at[n_pre - 1])[:, :n].t())
                iw, last = 0, None
                for ti in range(len(tim) - 1):
                    tt, d = tim[ti], tim[ti + 1] - tim[ti]
                    last = (rdiag, d)                                                   # m_exp = log|diag r| / dt
                    pre = None
                    if write_steps > 0 and ti % write_steps == 0:
                        pre = record(iw, n_pre - 1 + ti, rdiag, d)
                        iw += 1
                    sub = np.concatenate((np.arange(tt, tt + d, mdt), np.full((1,), tt + d)))
                    rdiag = propagate(n_pre - 1 + ti, sub, 1, pre)
                record(self.n_records - 1, len(at) - 1, last[0] if last else None, last[1] if last else 1.0)
            else:
                # ---- forward Lyapunov vectors (lyapunov.py:480-552): integrate the tangent model backward in time ----
                tim, post = self._pretime, self._time            # the reference's (time, posttime)
                rpost, rtim = reverse(post), reverse(tim)
                n_t = len(tim)
                rdiag = None
                for ti in range(len(rpost) - 1):
                    tt, d = rpost[ti], rpost[ti + 1] - rpost[ti]
                    sub = np.concatenate((np.arange(tt + d, tt, mdt), np.full((1,), tt)))
                    rdiag = propagate(n_t - 1 + (len(post) - 1 - ti), sub, -1)          # posttraj[:, :, -1-ti]
                if rdiag is None:
                    rdiag = rdiag0
                iw, last, y_idx = self.n_records - 1, None, n_t - 1
                for ti in range(len(rtim) - 1):
                    tt, d = rtim[ti], rtim[ti + 1] - rtim[ti]
                    y_idx = n_t - 1 - ti                                                 # traj[:, :, -1-ti]
                    last = (rdiag, d)
                    if write_steps > 0 and ti % write_steps == 0:
                        record(iw, y_idx, rdiag, d)
                        iw -= 1
                    sub = np.concatenate((np.arange(tt + d, tt, mdt), np.full((1,), tt)))
                    rdiag = propagate(y_idx, sub, -1)
                record(0, y_idx, last[0] if last else None, last[1] if last else 1.0)

            rec.finish(wait=not grouped)               # (a member group: the caller waits for the drain after the last group)
        except BaseException:
            # the drain thread may still be scattering flushed windows into the result blocks: nothing of this run is on its way
            # into them (or left in the model's drain tickets / staging pool) when the exception reaches the caller, who frees or
            # recycles those blocks
            rec.abort()
            raise
        return base.n_windows, rec.n_windows

    def get_lyapunovs(self):
        """``(time, traj, exponents, vectors)``: traj (n_traj, n_dim, n_records), exponents (n_traj, n_vec, n_records),
        vectors (n_traj, n_dim, n_vec, n_records), all `np.squeeze`d; time is a scalar for ``write_steps=0``
        (lyapunov.py:360-393)."""
        tt = self._time if self._forward == -1 else self._pretime
        if self.write_steps > 0:
            kept = tt[::self.write_steps]
            if kept[-1] != tt[-1]:
                kept = np.concatenate((kept, np.full((1,), tt[-1])))
            return kept, np.squeeze(self._recorded_traj), np.squeeze(self._recorded_exp), np.squeeze(self._recorded_vec)
        return tt[-1], np.squeeze(self._recorded_traj), np.squeeze(self._recorded_exp), np.squeeze(self._recorded_vec)


def _intersect_subspaces(bvec, fvec, num_threads=1):
    """CLV j at every (member, record): the direction in which the span of the first j + 1 backward vectors meets the span of
    the last n_dim - j forward vectors -- B_j u with u the leading left singular vector of B_j^T F_j (lyapunov.py:1313-1317).
    `bvec`, `fvec`: (n_traj, n_dim, n_dim, n_records); returns (n_traj, n_records, n_dim, n_dim).  The SVDs are LAPACK's, batched
    over members and records, the members spread over `num_threads` host threads (NumPy releases the GIL inside them)."""
    nt, nd = bvec.shape[0], bvec.shape[1]
    bb = np.moveaxis(bvec, 3, 1)                                # (nt, nr, nd, nd)
    ff = np.moveaxis(fvec, 3, 1)
    clv = np.zeros(bb.shape)

    def block(lo, hi):
        for j in range(nd):
            u = np.linalg.svd(np.swapaxes(bb[lo:hi, ..., :j + 1], -1, -2) @ ff[lo:hi, ..., :nd - j])[0]
            clv[lo:hi, ..., j] = (bb[lo:hi, ..., :j + 1] @ u[..., :, :1])[..., 0]
    workers = int(max(1, min(num_threads or 1, nt, 32)))
    if workers == 1:
        block(0, nt)
    else:
        from concurrent.futures import ThreadPoolExecutor
        cuts = [nt * w // workers for w in range(workers + 1)]
        with ThreadPoolExecutor(workers) as pool:
            for fut in [pool.submit(block, cuts[w], cuts[w + 1]) for w in range(workers)]:
                fut.result()
    return clv


class CovariantLyapunovsEstimator(object):
    """Covariant Lyapunov vectors (CLVs) along the trajectories of an ensemble (reference: qgs/toolbox/lyapunov.py:635-1092
    CovariantLyapunovsEstimator, loops at :1174-1330).  Same API: ``set_func(f, fjac)``, ``compute_clvs(t0, ta, tb, tc, dt,
    mdt, ic, write_steps, n_vec, method, backward_vectors, forward_vectors)``, ``get_clvs()``, ``get_blvs()``, ``get_flvs()``.

    * ``method=0`` (Ginelli et al.): the forward part -- Benettin steps from `t0` to `tc`, keeping the backward vectors
      between `ta` and `tb` and the R matrices between `ta` and `tc` -- is one run of the estimator above on the GPU with every
      interval recorded (R of an interval = Q(next)^T times the propagated matrix before its QR; the QR kernel follows LAPACK's
      sign convention, so this is the R np.linalg.qr returns).  The backward recursion on the (n_vec, n_vec) coefficient
      matrices is sequential in time and tiny; it runs on the host, all members at once.
    * ``method=1`` (intersection of the subspaces spanned by backward and forward vectors): two runs of the estimator above
      (backward vectors over [t0, tb], forward vectors over [ta, tc]), the SVDs of (j+1) x (n_dim-j) matrices batched over
      members and records on the host, and one tangent-model step of `mdt` for the local exponents (TGLS kernels).

    Random numbers are drawn from `np.random` in the reference's order (per trajectory for method 0: start matrix, second
    matrix, noise vectors; method 1: the forward run's matrices, then the backward run's), all before anything runs.
    """

    def __init__(self, num_threads=None, b=None, c=None, a=None, number_of_dimensions=None, noise_pert=0., method=0, device=None):
        # (the reference's worker processes; here: host threads of the batched SVDs of method 1)
        self.num_threads = multiprocessing.cpu_count() if num_threads is None else num_threads
        self.device = device
        self.b, self.c, self.a = _fn.resolve_tableau(b, c, a)
        self.noise_pert = noise_pert
        self.ic = None
        self._time = None
        self._pretime = None
        self._aftertime = None
        self._recorded_traj = None
        self._recorded_exp = None
        self._recorded_vec = None
        self._recorded_bvec = None
        self._recorded_fvec = None
        self.n_traj = 0
        self.n_dim = number_of_dimensions
        self.n_records = 0
        self.n_vec = 0
        self.write_steps = 0
        self.method = method
        self.func = None
        self.func_jac = None
        self._est = None
        self.last_timing = None           # method 0: seconds spent in the GPU part / on R / in the backward recursion
        self.last_path = None             # method 0: 'device' (Q / R record resident on the GPU) or 'host'
        self.device_resident = None       # method 0: None = on the device when the Q / R record fits, else the host path; True / False force one

    def terminate(self):
        if self._est is not None:
            self._est.terminate()
        self._est = None

    def set_noise_pert(self, noise_pert):
        self.noise_pert = noise_pert
        self.start()

    def set_bca(self, b=None, c=None, a=None, ic_init=True):
        if a is not None:
            self.a = a
        if b is not None:
            self.b = b
        if c is not None:
            self.c = c
        if ic_init:
            self.ic = None
        self.start()

    def start(self):
        self.terminate()
        if self.func is not None:
            self._est = LyapunovsEstimator(num_threads=self.num_threads, b=self.b, c=self.c, a=self.a, device=self.device)
            self._est.set_func(self.func, self.func_jac)

    def set_func(self, f, fjac):
        self.func = f
        self.func_jac = fjac
        self.start()

    # ------------------------------------------------------------------------------------------------
    def compute_clvs(self, t0, ta, tb, tc, dt, mdt, ic=None, write_steps=1, n_vec=None, method=None, backward_vectors=False,
                     forward_vectors=False):
        """CLVs between `ta` and `tb` along the trajectories started from `ic` at `t0` and integrated to `tc`: [t0, ta] lets
        the backward vectors converge, [tb, tc] the backward recursion (method 0) or the forward vectors (method 1)
        (lyapunov.py:863-988)."""
        if self.func is None or self.func_jac is None:
            print('No function to integrate defined!')
            return 0
        if self._est is None:
            self.start()
        self.ic = np.zeros(_fn.dimension_of(self.func)) if ic is None else ic
        if len(self.ic.shape) == 1:
            self.ic = self.ic.reshape((1, -1))
        self.n_traj, self.n_dim = self.ic.shape
        self.n_vec = self.n_dim if n_vec is None else n_vec
        if method is not None:
            self.method = method
        self._pretime = _fn.time_grid(t0, ta, dt)
        self._time = _fn.time_grid(ta, tb, dt)
        self._aftertime = _fn.time_grid(tb, tc, dt)
        self.write_steps = write_steps
        if write_steps == 0:
            self.n_records = 1
        else:
            tot = self._time[::write_steps]
            self.n_records = len(tot) + (1 if tot[-1] != self._time[-1] else 0)
        self._recorded_bvec = self._recorded_fvec = None
        if self.method == 0:
            self._ginelli(mdt)
        else:
            self._subspaces(mdt, backward_vectors, forward_vectors)

    def _ginelli(self, mdt):
        """lyapunov.py:1174-1288."""
        nt, nd, nv, ws = self.n_traj, self.n_dim, self.n_vec, self.write_steps
        tw = len(self._time) - 1
        tew = len(self._time) + len(self._aftertime) - 2
        # every draw of the reference's loop, trajectory by trajectory
        # (start matrix, second matrix, one noise vector of n_dim entries per step behind tb, one of n_vec entries per step on
        # [ta, tb]; np.random.randn delivers the same stream however the calls are cut, so this is one call)
        k_after = max(0, tew - 1 - tw)
        cuts = np.cumsum([nd * nv, nd * nv, k_after * nd, (tw + 1) * nv])
        stream = np.random.randn(nt, cuts[-1])
        a0 = stream[:, :cuts[0]].reshape(nt, nd, nv)
        a1 = stream[:, cuts[0]:cuts[1]].reshape(nt, nd, nv)
        noise_after = stream[:, cuts[1]:cuts[2]].reshape(nt, k_after, nd)
        noise_time = stream[:, cuts[2]:].reshape(nt, tw + 1, nv)
        if len(self._aftertime) < 2:
            raise ValueError('method 0 needs tc > tb: the backward recursion starts behind the window the vectors are kept on')
        draws = (a0, a1, noise_after, noise_time)
        want = self.device_resident if _fn.on_device(self.func) else False      # (user-written callables: host loops)
        if want is None or want:
            done = self._ginelli_device(mdt, draws)
            if want and not done:
                raise MemoryError('the Q / R record of this run does not fit the device(s)')
            if done:
                return
        self._ginelli_host(mdt, draws)

    def _ginelli_host(self, mdt, draws):
        """The forward part on the GPU with its records delivered to the host (device windows: any length the host can hold),
        R = Q^T A and the backward recursion in NumPy, all members at once."""
        import time as _clock
        nt, nd, nv, ws = self.n_traj, self.n_dim, self.n_vec, self.write_steps
        tw = len(self._time) - 1
        tew = len(self._time) + len(self._aftertime) - 2
        a0, a1, noise_after, noise_time = draws
        # parts one to three: Benettin steps over [t0, ta] (spin-up) and [ta, tc], every interval recorded
        t_start = _clock.perf_counter()
        est = self._est
        est._run(self._pretime, np.concatenate((self._time, self._aftertime[1:])), mdt, self.ic, 1, nv, False, False, False,
                 a0=a0, pre_qr=True, fine_base=True)
        q_all, a_all, traj_all = est._recorded_vec, est._recorded_pre, est._recorded_traj     # (nt, nd, nv, tew + 1), ..., (nt, nd, tew + 1)
        # R of interval ti: Q(ti + 1)^T A(ti), (nt, tew, nv, nv)
        t_device = _clock.perf_counter()
        if a_all is not None:
            r_all = np.triu(np.matmul(np.transpose(q_all[..., 1:], (0, 3, 2, 1)), np.transpose(a_all[..., :-1], (0, 3, 1, 2))))
        else:                                                   # the host loops keep the R of np.linalg.qr themselves
            r_all = np.moveaxis(est._recorded_r[..., :-1], 3, 1)
        t_r = _clock.perf_counter()
        # parts four and five: the backward recursion on the coefficient matrices, all members at once
        am, _ = _normalize_columns(np.stack([np.linalg.qr(a1[i])[1] for i in range(nt)]))
        diag = np.arange(nv)
        for k, ti in enumerate(range(tew - 1, tw, -1)):
            am_new = _solve_triangular(r_all[:, ti], am)
            am_new[:, diag, diag] += noise_after[:, k, :nv] * self.noise_pert
            am, _ = _normalize_columns(am_new)
        dte = np.concatenate((np.diff(self._time), np.full((1,), self._aftertime[1] - self._aftertime[0])))
        rec_traj = np.zeros((nt, nd, self.n_records))
        rec_vec = np.zeros((nt, nd, nv, self.n_records))
        rec_exp = np.zeros((nt, nv, self.n_records))
        iw = 1
        mloc = np.ones((nt, nv))
        for k, ti in enumerate(range(tw, -1, -1)):
            am_new = _solve_triangular(r_all[:, ti], am)
            am_new[:, diag, diag] += noise_time[:, k] * self.noise_pert
            am, mloc = _normalize_columns(am_new)
            if ws > 0 and (tw - ti) % ws == 0:
                rec_traj[:, :, -iw] = traj_all[:, :, ti]
                rec_exp[:, :, -iw] = -np.log(np.abs(mloc)) / dte[ti]
                rec_vec[:, :, :, -iw] = q_all[:, :, :, ti] @ am
                iw += 1
        rec_traj[:, :, 0] = traj_all[:, :, 0]
        rec_exp[:, :, 0] = -np.log(np.abs(mloc)) / dte[0]
        rec_vec[:, :, :, 0] = q_all[:, :, :, 0] @ am
        self._recorded_traj, self._recorded_exp, self._recorded_vec = rec_traj, rec_exp, rec_vec
        # where the time went: the GPU part (Benettin run incl. its records reaching the host), R = Q^T A, the backward recursion
        self.last_path = 'host' if a_all is not None else 'host (user-written callables)'
        self.last_timing = {'benettin_run_s': t_device - t_start, 'r_matrices_s': t_r - t_device,
                            'backward_recursion_s': _clock.perf_counter() - t_r}

    def _ginelli_device(self, mdt, draws):
        """Everything on the GPU(s): the Q of [ta, tb] and the R of [ta, tc] stay in device memory between the forward part and
        the backward recursion (`qgs_batched_matmul_device` for R = Q^T A and for the vectors Q a, `qgs_clv_backstep_device` for
        a <- normalise(R^-1 a)); only the records the caller asked for travel to the host.  Returns False when that Q / R
        record does not fit the free device memory of a shard's GPU (the host path has no such limit)."""
        import time as _clock
        import torch
        nt, nd, nv = self.n_traj, self.n_dim, self.n_vec
        tw = len(self._time) - 1
        tew = len(self._time) + len(self._aftertime) - 2
        model = _fn.hip_model_of(self.func, device=_fn.resolve_device(self.device, nt))
        shards = getattr(model, 'models', None)
        pieces = [(model, 0, nt)] if shards is None else [(shards[i],) + tuple(model.shard(nt, i)) for i in range(len(shards))]
        pieces = [p for p in pieces if p[2] > 0]
        # does the record fit?  (shards that share a GPU share its memory)
        need = {}
        for m, a, cnt in pieces:
            ld = (cnt + 63) // 64 * 64
            need[m.device] = need.get(m.device, 0) + 8 * ld * (tew * nv * nv + (tw + 1) * (nd * nv + nd) + 8 * nd * nv)
        for d, bytes_needed in need.items():
            free, _total = torch.cuda.mem_get_info(torch.device('cuda', d))
            if bytes_needed + 3 * _window_budget_bytes() // 4 > 0.8 * free:
                return False
        t_start = _clock.perf_counter()
        nr = self.n_records
        rec_traj = _lib._RESULTS.empty((nt, nd, nr))
        rec_vec = _lib._RESULTS.empty((nt, nd, nv, nr))
        rec_exp = _lib._RESULTS.empty((nt, nv, nr))
        timing = [None] * len(pieces)

        def work(i):
            m, a, cnt = pieces[i]
            with torch.cuda.device(torch.device('cuda', m.device)):
                timing[i] = self._ginelli_shard(m, self.ic[a:a + cnt], mdt, [x[a:a + cnt] for x in draws],
                                                (rec_traj[a:a + cnt], rec_vec[a:a + cnt], rec_exp[a:a + cnt]))
        _in_threads(work, len(pieces))
        self._recorded_traj, self._recorded_exp, self._recorded_vec = rec_traj, rec_exp, rec_vec
        self.last_path = 'device'
        self.last_timing = {'benettin_run_s': max(t[0] for t in timing), 'r_matrices_s': 0.0,
                            'backward_recursion_s': max(t[1] for t in timing), 'wall_s': _clock.perf_counter() - t_start}
        return True

    def _ginelli_shard(self, m, ic, mdt, draws, outs):
        import time as _clock
        import torch
        a0, a1, noise_after, noise_time = draws
        nd, nv, n, ws = self.n_dim, self.n_vec, ic.shape[0], self.write_steps
        ld = (n + 63) // 64 * 64
        dev = torch.device('cuda', m.device)
        f64 = torch.float64
        stream = torch.cuda.current_stream(dev).cuda_stream
        budget = _window_budget_bytes()
        b, c, a = self.b, self.c, self.a
        tw = len(self._time) - 1
        tew = len(self._time) + len(self._aftertime) - 2
        n_pre = len(self._pretime)
        t_start = _clock.perf_counter()

        def upload(host):                                   # (n, ...) host rows -> device tensor, through the library's bounce blocks
            return _lib.to_device(host, dev)

        def basis(rows):
            """(n, nd, nv) host matrices -> (Q[mode][vector][member] of their QR, R[nv][nv][member] = Q^T A)"""
            rows_dev = upload(rows)
            mat = torch.zeros((nd, nv, ld), dtype=f64, device=dev)
            m.pack_tangent(n, ld, nv, rows_dev.data_ptr(), mat.data_ptr(), stream)
            q0 = mat.clone()
            rd = torch.empty((nv, ld), dtype=f64, device=dev)
            m.batched_qr_device(n, ld, nd, nv, q0.data_ptr(), rd.data_ptr(), stream)
            r0 = torch.zeros((nv, nv, ld), dtype=f64, device=dev)
            m.batched_matmul_device(n, ld, nv, nd, nv, q0.data_ptr(), mat.data_ptr(), r0.data_ptr(), trans_a=True, triangular=1, stream=stream)
            torch.cuda.current_stream(dev).synchronize()
            return q0, r0

        # base trajectory on the sub-step grid (the reference's loop advances the state with the tangent integrator, :1203-1247)
        coarse = np.concatenate((self._pretime[:-1], self._time, self._aftertime[1:]))
        at = np.arange(len(coarse))
        fine = []
        for i in range(len(coarse) - 1):
            tt, d = coarse[i], coarse[i + 1] - coarse[i]
            fine.append(np.arange(tt, tt + d, mdt))
            at[i + 1] = at[i] + len(fine[-1])
        ic_modes = torch.zeros((nd, ld), dtype=f64, device=dev)
        ic_modes[:, :n] = upload(ic.T)
        base = _BaseTrajectory(torch, m, n, ld, np.concatenate(fine + [coarse[-1:]]), ic_modes, budget // 4, False, (b, c, a), stream)

        # parts one to three (:1195-1247): Benettin steps; Q and the states kept on [ta, tb], R on [ta, tc]
        q, _ = basis(a0)
        q_new = torch.empty((1, nd, nv, ld), dtype=f64, device=dev)
        pre = torch.empty((nd, nv, ld), dtype=f64, device=dev)
        y_end = torch.empty((1, nd, ld), dtype=f64, device=dev)
        rdiag = torch.empty((nv, ld), dtype=f64, device=dev)
        q_all = torch.empty((tw + 1, nd * nv, ld), dtype=f64, device=dev)
        t_all = torch.empty((tw + 1, nd, ld), dtype=f64, device=dev)
        r_all = torch.empty((tew, nv * nv, ld), dtype=f64, device=dev)
        for i in range(len(coarse) - 1):
            ti = i - (n_pre - 1)                             # interval index on [ta, tc]; negative during the spin-up
            if 0 <= ti <= tw:
                q_all[ti].copy_(q.reshape(nd * nv, ld))
                t_all[ti].copy_(base.state(at[i]))
            tt, d = coarse[i], coarse[i + 1] - coarse[i]
            sub = np.concatenate((np.arange(tt, tt + d, mdt), np.full((1,), tt + d)))
            m.rk_tgls_integrate_device(n, ld, nv, base.state(at[i]).data_ptr(), q.data_ptr(), sub, 1, 0, b, c, a, False, 1.,
                                       y_end.data_ptr(), q_new.data_ptr(), stream)
            if ti >= 0:
                pre.copy_(q_new[0])
            m.batched_qr_device(n, ld, nd, nv, q_new.data_ptr(), rdiag.data_ptr(), stream)
            if ti >= 0:
                m.batched_matmul_device(n, ld, nv, nd, nv, q_new.data_ptr(), pre.data_ptr(), r_all[ti].data_ptr(), trans_a=True,
                                        triangular=1, stream=stream)
            q, q_new = q_new[0], q.unsqueeze(0)
        torch.cuda.current_stream(dev).synchronize()
        t_forward = _clock.perf_counter()

        # parts four and five (:1249-1283): a <- normalise(R^-1 a) backward in time, from the R of a second random matrix
        _, r1 = basis(a1)
        eye = torch.zeros((nv, nv, ld), dtype=f64, device=dev)
        for i in range(nv):
            eye[i, i].fill_(1.0)
        am = torch.empty((nv, nv, ld), dtype=f64, device=dev)
        am_new = torch.empty((nv, nv, ld), dtype=f64, device=dev)
        norm = torch.empty((nv, ld), dtype=f64, device=dev)
        m.clv_backstep_device(n, ld, nv, eye.data_ptr(), r1.data_ptr(), am.data_ptr(), norm.data_ptr(), stream=stream)   # = normalise(r1)
        pert = float(self.noise_pert)

        def noise_on_device(host):                           # (n, steps, >= nv) -> [step][vector][member]
            if pert == 0.0 or host.shape[1] == 0:
                return None
            t = torch.zeros((host.shape[1], nv, ld), dtype=f64, device=dev)
            t[:, :, :n] = upload(np.transpose(host[:, :, :nv], (1, 2, 0)))
            return t
        nz_after, nz_time = noise_on_device(noise_after), noise_on_device(noise_time)
        for k, ti in enumerate(range(tew - 1, tw, -1)):
            m.clv_backstep_device(n, ld, nv, r_all[ti].data_ptr(), am.data_ptr(), am_new.data_ptr(), norm.data_ptr(),
                                  nz_after[k].data_ptr() if nz_after is not None else None, pert, stream)
            am, am_new = am_new, am
        out_traj, out_vec, out_exp = outs
        rec = _RecordWindows(torch, m, n, ld, (nd * nv, nd, nv), (out_vec, out_traj, out_exp), self.n_records, budget // 2, dev)
        dte = np.concatenate((np.diff(self._time), np.full((1,), self._aftertime[1] - self._aftertime[0])))

        def record(index, ti):
            s_vec, s_traj, s_norm = rec.slot(index)
            m.batched_matmul_device(n, ld, nd, nv, nv, q_all[ti].data_ptr(), am.data_ptr(), s_vec.data_ptr(), triangular=2, stream=stream)
            s_traj.copy_(t_all[ti])
            # local exponents from the norms of the backward step, on the device: -log|norm| / dt   (lyapunov.py:1280, 1285)
            m.local_exponents_device(nv * ld, norm.data_ptr(), -dte[ti], s_norm.data_ptr(), stream)
        iw = 1
        try:
            for k, ti in enumerate(range(tw, -1, -1)):
                m.clv_backstep_device(n, ld, nv, r_all[ti].data_ptr(), am.data_ptr(), am_new.data_ptr(), norm.data_ptr(),
                                      nz_time[k].data_ptr() if nz_time is not None else None, pert, stream)
                am, am_new = am_new, am
                if ws > 0 and (tw - ti) % ws == 0:
                    record(self.n_records - iw, ti)
                    iw += 1
            record(0, 0)
            rec.finish()
        except BaseException:
            rec.abort()            # (as in LyapunovsEstimator._compute_shard_on_current_device)
            raise
        return t_forward - t_start, _clock.perf_counter() - t_forward

    def _subspaces(self, mdt, backward_vectors, forward_vectors):
        """lyapunov.py:1292-1330 (always the full basis: the reference passes n_dim vectors to both runs)."""
        nt, nd, ws = self.n_traj, self.n_dim, self.write_steps
        a0_forward = np.random.random((nt, nd, nd))            # the reference's forward run draws first, then the backward run
        a0_backward = np.random.random((nt, nd, nd))
        est = self._est
        # backward vectors on [ta, tb] after the spin-up [t0, ta]; its base trajectory also gives the states at ta
        est._run(self._pretime, self._time, mdt, self.ic, ws, nd, False, False, False, a0=a0_backward, junction=True)
        traj, bvec = est._recorded_traj, est._recorded_vec        # (result blocks stay the caller's while referenced, _lib._ResultPool)
        y_ta = est._junction
        # forward vectors on [ta, tb]: the tangent model backward in time from tc
        est._run(self._time, self._aftertime, mdt, y_ta, ws, nd, True, False, False, a0=a0_forward)
        fvec = est._recorded_vec
        nr = traj.shape[-1]
        clv = _intersect_subspaces(bvec, fvec, self.num_threads)             # (nt, nr, nd, nd)
        # local exponents: growth of every CLV over one step of the tangent model
        states = np.ascontiguousarray(np.moveaxis(traj, 2, 1).reshape(nt * nr, nd))
        _, sol = _fn.run_rk_tgls(self.func, self.func_jac, np.array([0., mdt]), states, clv.reshape(nt * nr, nd, nd), 1, 0,
                                 self.b, self.c, self.a, False, 1., None, device=self.device)
        _, growth = _normalize_columns(sol[..., 0])
        self._recorded_traj = traj
        self._recorded_exp = np.moveaxis((np.log(np.abs(growth)) / mdt).reshape(nt, nr, nd), 1, 2)
        self._recorded_vec = np.moveaxis(clv, 1, 3)
        self.n_vec = nd
        if backward_vectors:
            self._recorded_bvec = bvec
        if forward_vectors:
            self._recorded_fvec = fvec

    def _times(self):
        if self.write_steps > 0:
            kept = self._time[::self.write_steps]
            if kept[-1] != self._time[-1]:
                kept = np.concatenate((kept, np.full((1,), self._time[-1])))
            return kept
        return self._time[-1]

    def get_clvs(self):
        """``(time, traj, exponents, vectors)`` of the last estimation, `np.squeeze`d (lyapunov.py:990-1018)."""
        return self._times(), np.squeeze(self._recorded_traj), np.squeeze(self._recorded_exp), np.squeeze(self._recorded_vec)

    def get_blvs(self):
        """The backward vectors of the last ``method=1`` estimation with ``backward_vectors=True``, else None (lyapunov.py:1020-1055)."""
        if self._recorded_bvec is None:
            return None
        return self._times(), np.squeeze(self._recorded_traj), np.squeeze(self._recorded_exp), np.squeeze(self._recorded_bvec)

    def get_flvs(self):
        """The forward vectors of the last ``method=1`` estimation with ``forward_vectors=True``, else None (lyapunov.py:1057-1092)."""
        if self._recorded_fvec is None:
            return None
        return self._times(), np.squeeze(self._recorded_traj), np.squeeze(self._recorded_exp), np.squeeze(self._recorded_fvec)
