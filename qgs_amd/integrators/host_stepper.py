"""Host steppers for what cannot run inside a HIP kernel: user-written Python callables.

The reference's integrators accept ANY (numba-jitted) callable -- its own docstring examples integrate a Lorenz-84
system written by hand (qgs/integrators/integrator.py:1237-1256) -- and the tangent model accepts a user `boundary(t, x)`
term (integrate.py:600-603, integrator.py:1285-1291).  A Python function cannot be called from a wavefront, so these two
cases are stepped here, on the host, in NumPy:

  * `f` (and `fjac`) without a tensor attached (no `hip_model`): the whole integration runs here;
  * `f` / `fjac` from `create_tendencies` together with a `boundary` callable: the Runge-Kutta combination runs here, the
    tendencies and the Jacobian are still evaluated by the HIP kernels (`f`, `Df` are called with the whole ensemble at once).

Everything else -- tensor tendencies with the default zero boundary -- goes to the device and never reaches this module.
This is not the parity checker the test suite compares the HIP kernels with, and it is not part of any timed path.

The loops follow `_integrate_runge_kutta_jit` (integrate.py:182-223) and `_integrate_runge_kutta_tgls_jit` (:555-614): same
record rule, same stage formulas, trajectories advanced together instead of one after the other (they are independent).
"""
import numpy as np


def n_records(time, write_steps):
    """integrate.py:190-196"""
    if write_steps == 0:
        return 1
    tot = time[::write_steps]
    return len(tot) + (1 if tot[-1] != time[-1] else 0)


def _batched(func):
    """`func(t, X)` for X of shape (n_traj, n_dim): tensor functions take the batch in one (GPU) call, a user callable is
    applied row by row, as the reference does."""
    if getattr(func, 'hip_model', None) is not None:
        return lambda t, X: np.asarray(func(t, X))
    return lambda t, X: np.stack([np.asarray(func(t, x), dtype=np.float64) for x in X])


def discover_dimension(func):
    """The reference's way of finding n_dim when neither `number_of_dimensions` nor `ic` is given (integrator.py:346-359):
    call `func` with zeros of growing length until it stops raising."""
    i = 1
    while True:
        try:
            func(0., np.zeros(i))
        except Exception:
            i += 1
            if i > 100000:
                raise TypeError('cannot determine the system dimension from the callable')
        else:
            break
    return len(func(0., np.zeros(i)))


def integrate_runge_kutta(f, time, ic, time_direction, write_steps, b, c, a):
    """(n_traj, n_dim, n_records), direction-corrected -- `_integrate_runge_kutta_jit` for a Python callable."""
    time, ic = np.asarray(time, dtype=np.float64), np.asarray(ic, dtype=np.float64)
    b, c, a = np.asarray(b, dtype=np.float64), np.asarray(c, dtype=np.float64), np.asarray(a, dtype=np.float64)
    n_traj, n_dim = ic.shape
    s = len(b)
    nrec = n_records(time, write_steps)
    rec = np.zeros((n_traj, n_dim, nrec))
    directed = time[::-1] if time_direction == -1 else time
    fb = _batched(f)
    y = ic.copy()
    k = np.zeros((s, n_traj, n_dim))
    iw = 0
    for ti in range(len(directed) - 1):
        tt, dt = directed[ti], directed[ti + 1] - directed[ti]
        if write_steps > 0 and ti % write_steps == 0:
            rec[:, :, iw] = y
            iw += 1
        k.fill(0.)
        for i in range(s):
            y_s = y + dt * np.tensordot(a[i], k, axes=(0, 0))
            k[i] = fb(tt + c[i] * dt, y_s)
        y = y + dt * np.tensordot(b, k, axes=(0, 0))
    rec[:, :, -1] = y
    return rec[:, :, ::time_direction]


def integrate_runge_kutta_tgls(f, fjac, time, ic, tg_ic, time_direction, write_steps, b, c, a, adjoint, inverse, boundary=None):
    """`_integrate_runge_kutta_tgls_jit`: trajectories (n_traj, n_dim, n_records) and tangent / adjoint propagation
    (n_traj, n_dim, n_tg, n_records) of `tg_ic` (n_traj, n_dim, n_tg); `inverse` is the +-1 factor, `boundary(t, x)` the
    inhomogeneous term added to every column (None: zero, integrate.py:235-237)."""
    time, ic, tg_ic = np.asarray(time, dtype=np.float64), np.asarray(ic, dtype=np.float64), np.asarray(tg_ic, dtype=np.float64)
    b, c, a = np.asarray(b, dtype=np.float64), np.asarray(c, dtype=np.float64), np.asarray(a, dtype=np.float64)
    n_traj, n_dim = ic.shape
    s = len(b)
    nrec = n_records(time, write_steps)
    rec = np.zeros((n_traj, n_dim, nrec))
    recm = np.zeros((n_traj, tg_ic.shape[1], tg_ic.shape[2], nrec))
    directed = time[::-1] if time_direction == -1 else time
    fb, jb = _batched(f), _batched(fjac)
    y, fm = ic.copy(), tg_ic.copy()
    rec[:, :, 0] = ic
    recm[:, :, :, 0] = tg_ic
    k = np.zeros((s, n_traj, n_dim))
    km = np.zeros((s,) + tg_ic.shape)
    iw = 0
    for ti in range(len(directed) - 1):
        tt, dt = directed[ti], directed[ti + 1] - directed[ti]
        if write_steps > 0 and ti % write_steps == 0:
            rec[:, :, iw] = y
            recm[:, :, :, iw] = fm
            iw += 1
        k.fill(0.)
        km.fill(0.)
        for i in range(s):
            ts = tt + c[i] * dt
            y_s = y + dt * np.tensordot(a[i], k, axes=(0, 0))
            k[i] = fb(ts, y_s)
            km_s = fm.copy()
            for j in range(s):
                km_s += dt * a[i, j] * km[j]
            jac = jb(ts, y_s)                                            # (n_traj, n_dim, n_dim)
            if adjoint:
                jac = np.swapaxes(jac, 1, 2)
            hom = inverse * np.matmul(jac, km_s)
            if boundary is not None:
                for m in range(n_traj):                                   # (hom.T + inhom.T).T: a vector is added to every column
                    inhom = np.asarray(boundary(ts, y_s[m]), dtype=np.float64)
                    hom[m] += inhom[:, np.newaxis] if inhom.ndim == 1 else inhom
            km[i] = hom
        y = y + dt * np.tensordot(b, k, axes=(0, 0))
        fm_new = fm.copy()
        for j in range(s):
            fm_new += dt * b[j] * km[j]
        fm = fm_new
    rec[:, :, -1] = y
    recm[:, :, :, -1] = fm
    return rec[:, :, ::time_direction], recm[:, :, :, ::time_direction]
