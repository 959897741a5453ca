"""Lyapunov exponents and backward / forward Lyapunov vectors with the Benettin algorithm
(reference: qgs/toolbox/lyapunov.py, class LyapunovsEstimator and the loops at :471-632).

Same API as the reference (`set_func(f, fjac)`, `compute_lyapunovs(t0, tw, t, dt, mdt, ...)`,
`get_lyapunovs()`), whole ensemble on the GPU:

* the base trajectory is one fused RK launch with every step recorded (device resident);
* per `dt` interval the tangent model is integrated over the `mdt` sub-steps by the TGLS kernels.  The
  reference propagates the identity and multiplies (`prop @ q`, lyapunov.py:546, 624); the tangent model is
  linear, so the `n_vec` columns of `q` are propagated directly (same result to rounding, n_dim/n_vec
  times less work);
* the re-orthonormalisation `np.linalg.qr` (lyapunov.py:547, 625) is the batched Householder QR kernel
  `qgs_batched_qr_device` (LAPACK sign convention, so the vectors match the reference's);
* exponents `log|diag R| / dt` are formed on the host at the end from the stored diagonals.

The random initial basis is drawn exactly like the reference does (one `np.random.random((n_dim, n_vec))`
per trajectory, in trajectory order), so seeded runs are reproducible against it.
PyTorch is used for device buffers only.
"""
import multiprocessing

import numpy as np

from qgs_amd.integrators import integrate as _fn
from qgs_amd.functions.util import reverse


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

    def terminate(self):
        self._model = None

    def start(self):
        self.terminate()
        if self.func is not None:
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
        if self._model is None:
            self.start()
        self.ic = np.zeros(_fn.dimension_of(self.func)) if ic is None else ic
        if len(self.ic.shape) == 1:
            self.ic = self.ic.reshape((1, -1))
        self.n_traj, self.n_dim = self.ic.shape
        self.n_vec = self.n_dim if n_vec is None else n_vec
        self._pretime = _fn.time_grid(t0, tw, dt)
        self._time = _fn.time_grid(tw, t, dt)
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

        # random start bases: the matrices are drawn like the reference's (one draw per trajectory, in order:
        # `np.random.random((ndim, nv))` consumes the generator exactly as n such calls in a row do) -- for the WHOLE ensemble
        # before it is split over devices
        a0 = np.random.random((self.n_traj, self.n_dim, self.n_vec))
        model = _fn.hip_model_of(self.func, device=_fn.resolve_device(self.device, self.n_traj))
        shards = getattr(model, 'models', None)
        if shards is None:
            self._recorded_traj, self._recorded_vec, self._recorded_exp = self._compute_shard(model, self.ic, a0, mdt)
            return
        # several GPUs: contiguous member shards, one host thread each (the work of a shard is a chain of kernel launches)
        import threading
        parts, errors = [None] * len(shards), []

        def run(i):
            try:
                a, cnt = model.shard(self.n_traj, i)
                if cnt > 0:
                    parts[i] = self._compute_shard(shards[i], self.ic[a:a + cnt], a0[a:a + cnt], mdt)
            except Exception as e:                       # re-raised on the calling thread
                errors.append(e)
        threads = [threading.Thread(target=run, args=(i,)) for i in range(len(shards))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        parts = [q for q in parts if q is not None]
        self._recorded_traj, self._recorded_vec, self._recorded_exp = (np.concatenate([q[k] for q in parts], axis=0) for k in range(3))

    def _compute_shard(self, m, ic, a0, mdt):
        """The Benettin loops for the members `ic` (n, n_dim) with start matrices `a0` (n, n_dim, n_vec) on model `m`'s GPU;
        returns (traj, vectors, exponents) in the reference's layouts."""
        import torch
        with torch.cuda.device(torch.device('cuda', m.device)):       # this thread's current device for the duration of the call only
            return self._compute_shard_on_current_device(m, ic, a0, mdt)

    def _compute_shard_on_current_device(self, m, ic, a0, mdt):
        import torch
        forward, adjoint, write_steps = self._forward == 1, self._adjoint, self.write_steps
        ndim, nv, n = self.n_dim, self.n_vec, ic.shape[0]
        ld = (n + 63) // 64 * 64
        dev = torch.device('cuda', m.device)
        f64 = torch.float64
        stream = torch.cuda.current_stream(dev).cuda_stream

        # base trajectory, every step recorded: R[step][mode][member]            (lyapunov.py:558 / :474)
        full_grid = np.concatenate((self._pretime[:-1], self._time))
        ic_modes = torch.zeros((ndim, ld), dtype=f64, device=dev)
        ic_modes[:, :n] = torch.from_numpy(np.ascontiguousarray(ic.T)).to(dev)
        base = torch.empty((len(full_grid), ndim, ld), dtype=f64, device=dev)
        m.rk_integrate_device(n, ld, ic_modes.data_ptr(), full_grid, 1, 1, self.b, self.c, self.a, base.data_ptr(), stream)
        n_pre = len(self._pretime)

        # orthonormal start basis: QR of the drawn matrices with the same batched Householder kernel as in the loop
        # (np.linalg.qr on the host took 30 us per member: 0.5 s at 16 384)
        # (the drawn matrices go up as they are, (n, n_dim, n_vec), and are brought into the device layout F[mode][vector][member]
        # by the pack kernel: the host-side transpose of 170 MB at config-4 size took longer than the whole spin-up)
        q = torch.zeros((ndim, nv, ld), dtype=f64, device=dev)
        a0_rows = torch.from_numpy(np.ascontiguousarray(a0)).to(dev)
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
        rec_vec = torch.zeros((self.n_records, ndim, nv, ld), dtype=f64, device=dev)
        rec_traj = torch.zeros((self.n_records, ndim, ld), dtype=f64, device=dev)

        def propagate(y_index, subtime, direction):
            """q <- Q of QR( TL_{subtime}(q) ) along the trajectory started at base[y_index]; returns diag(R)."""
            nonlocal q, q_new
            m.rk_tgls_integrate_device(n, ld, nv, base[y_index].data_ptr(), q.data_ptr(), subtime, direction, 0,
                                       self.b, self.c, self.a, adjoint, self._inverse, y_end.data_ptr(), q_new.data_ptr(),
                                       stream)
            rdiag = torch.empty((nv, ld), dtype=f64, device=dev)
            m.batched_qr_device(n, ld, ndim, nv, q_new.data_ptr(), rdiag.data_ptr(), stream)
            q, q_new = q_new[0], q.unsqueeze(0)
            return rdiag

        exp_sources = []            # (record index, rdiag tensor, dt) resolved on the host at the end
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
            iw, last = 0, None
            for ti in range(len(tim) - 1):
                tt, d = tim[ti], tim[ti + 1] - tim[ti]
                last = (rdiag, d)                                                   # m_exp = log|diag r| / dt
                if write_steps > 0 and ti % write_steps == 0:
                    exp_sources.append((iw, rdiag, d))
                    rec_traj[iw].copy_(base[n_pre - 1 + ti])
                    rec_vec[iw].copy_(q)
                    iw += 1
                sub = np.concatenate((np.arange(tt, tt + d, mdt), np.full((1,), tt + d)))
                rdiag = propagate(n_pre - 1 + ti, sub, 1)
            if last is not None:
                exp_sources.append((self.n_records - 1, last[0], last[1]))
            rec_traj[self.n_records - 1].copy_(base[len(full_grid) - 1])
            rec_vec[self.n_records - 1].copy_(q)
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
                    exp_sources.append((iw, rdiag, d))
                    rec_traj[iw].copy_(base[y_idx])
                    rec_vec[iw].copy_(q)
                    iw -= 1
                sub = np.concatenate((np.arange(tt + d, tt, mdt), np.full((1,), tt)))
                rdiag = propagate(y_idx, sub, -1)
            if last is not None:
                exp_sources.append((0, last[0], last[1]))
            rec_traj[0].copy_(base[y_idx])
            rec_vec[0].copy_(q)

        # device layout -> the reference's (n_traj, n_dim[, n_vec], n_records)
        out_traj = torch.empty((n, ndim, self.n_records), dtype=f64, device=dev)
        out_vec = torch.empty((n, ndim, nv, self.n_records), dtype=f64, device=dev)
        m.unpack_records(n, ld, ndim, self.n_records, rec_traj.data_ptr(), out_traj.data_ptr(), stream)
        m.unpack_records(n, ld, ndim * nv, self.n_records, rec_vec.data_ptr(), out_vec.data_ptr(), stream)
        recorded_exp = np.zeros((n, nv, self.n_records))
        for iw, rd, d in exp_sources:
            recorded_exp[:, :, iw] = (np.log(np.abs(rd[:, :n].cpu().numpy())) / d).T
        from qgs_amd import _lib
        torch.cuda.current_stream(dev).synchronize()
        return _lib.to_host(out_traj), _lib.to_host(out_vec), recorded_exp

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
