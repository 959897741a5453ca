"""Functional integration API (reference: qgs/integrators/integrate.py).

`integrate_runge_kutta` / `integrate_runge_kutta_tgls` keep the reference's signatures and output
conventions (integrate.py:29-179, 240-552); the stepping itself (the reference's
`_integrate_runge_kutta_jit` / `_integrate_runge_kutta_tgls_jit`, integrate.py:182-223, 555-614) runs on
the GPU through `qgs_rk_integrate` / `qgs_rk_tgls_integrate` for the whole ensemble at once.
"""
import os

import numpy as np

from qgs_amd.functions.util import reverse
from qgs_amd.integrators import host_stepper


def default_tableau():
    """Classic RK4 (integrate.py:149-155)."""
    c = np.array([0., 0.5, 0.5, 1.])
    b = np.array([1. / 6, 1. / 3, 1. / 3, 1. / 6])
    a = np.zeros((len(c), len(b)))
    a[1, 0] = 0.5
    a[2, 1] = 0.5
    a[3, 2] = 1.
    return b, c, a


def resolve_tableau(b, c, a):
    if a is None and b is None and c is None:
        return default_tableau()
    return b, c, a


def time_grid(t0, t, dt):
    """`np.concatenate((np.arange(t0, t, dt), [t]))` (integrate.py:162): the last step may be shorter than dt."""
    return np.concatenate((np.arange(t0, t, dt), np.full((1,), t)))


def record_times(time, write_steps, forward):
    """Time axis of the records (integrate.py:166-179 / integrator.py:409-424)."""
    if write_steps > 0:
        if forward:
            kept = time[::write_steps]
            if kept[-1] == time[-1]:
                return kept
            return np.concatenate((kept, np.full((1,), time[-1])))
        rtime = reverse(time[::-write_steps])
        if rtime[0] == time[0]:
            return rtime
        return np.concatenate((np.full((1,), time[0]), rtime))
    return time[-1]


def on_device(func):
    """True for the tendencies / Jacobian objects of qgs_amd.functions.tendencies (they carry the tensor the kernels are
    generated from); False for a plain Python callable, which is integrated on the host (host_stepper.py)."""
    return getattr(func, 'hip_model', None) is not None


#: ensembles of at least this many members are spread over all visible GPUs when no device was asked for (two full
#: 65 536-member batches: below that one MI355X integrates the ensemble in about the time a second one needs to get its share)
AUTO_ALL_DEVICES_MIN_TRAJ = 2 * 65536


def in_multi_process_job():
    """True when this process is one rank of a one-process-per-GPU job (a launcher set WORLD_SIZE / LOCAL_RANK / RANK, or a
    torch.distributed process group is up): its GPU is the one the launcher gave it, the other GPUs belong to the other ranks."""
    import os
    import sys
    for var in ('LOCAL_RANK', 'RANK'):
        if os.environ.get(var, '') != '':
            return True
    try:
        if int(os.environ.get('WORLD_SIZE', '1') or '1') > 1:
            return True
    except ValueError:
        return True
    dist = sys.modules.get('torch.distributed')          # never imports torch for the question
    try:
        return bool(dist is not None and dist.is_available() and dist.is_initialized())
    except Exception:
        return False


def auto_all_devices_enabled():
    """`QGS_HIP_AUTO_ALL_DEVICES=1`: large ensembles spread over every visible GPU without being asked to (below).  Off by default:
    the single-process device group has run on one physical GPU only so far (device 0 listed several times; no multi-GPU box was
    ever available to the build), and a default must not be the first thing to execute on two devices."""
    return os.environ.get('QGS_HIP_AUTO_ALL_DEVICES', '0') == '1'


def resolve_device(device, n_traj=None):
    """The `device` argument of the integrators: a GPU index, a list of indices, 'all', or None = the device the tendencies
    were created for.  With `QGS_HIP_AUTO_ALL_DEVICES=1`, None means all GPUs of the node when the ensemble has at least
    AUTO_ALL_DEVICES_MIN_TRAJ members, the node has several GPUs and this process has them to itself (the reference's default is
    every core of the machine, integrator.py:79-82).  A rank of a multi-process job never spreads by itself: there `None` keeps
    meaning the tendencies' own device, and all GPUs must be asked for explicitly (device='all')."""
    if (device is None and n_traj is not None and n_traj >= AUTO_ALL_DEVICES_MIN_TRAJ and auto_all_devices_enabled()
            and not in_multi_process_job()):
        from qgs_amd import _lib
        if len(_lib.visible_devices()) > 1:
            return 'all'
    return device


def hip_model_of(func, what='f', device=None):
    """The GPU handle behind a tendencies callable (on `device`, default the one the callable was created for)."""
    get = getattr(func, 'hip_model', None)
    if get is None:
        raise TypeError("%s carries no tensor: plain Python callables are integrated by qgs_amd.integrators.host_stepper, "
                        "not on the GPU" % what)
    return get(device)


def dimension_of(func):
    nd = getattr(func, 'ndim', None)
    if nd is None:
        return host_stepper.discover_dimension(func)      # the reference's probing loop (integrator.py:346-359)
    return int(nd)


def normalise_ic(ic, n_dim):
    if ic is None:
        ic = np.zeros(n_dim)
    ic = np.asarray(ic, dtype=np.float64)
    if ic.ndim == 1:
        ic = ic.reshape((1, -1))
    return ic


def normalise_tg_ic(tg_ic, n_traj, n_dim):
    """Bring the tangent initial conditions to (n_traj, n_dim, n_tg) (integrate.py:472-497)."""
    if tg_ic is None:
        tg_ic = np.eye(n_dim)
    tg_ic = np.asarray(tg_ic, dtype=np.float64)
    if tg_ic.ndim == 1:                                   # one vector, shared by all trajectories
        out = np.repeat(tg_ic.reshape((1, -1, 1)), n_traj, axis=0)
    elif tg_ic.ndim == 2:
        if tg_ic.shape[0] == n_traj:                      # one vector per trajectory
            out = tg_ic[..., np.newaxis]
        else:                                             # (n_tg, n_dim) shared by all trajectories
            out = np.repeat(np.swapaxes(tg_ic[np.newaxis, ...], 1, 2), n_traj, axis=0)
    elif tg_ic.ndim == 3:
        out = np.swapaxes(tg_ic, 1, 2) if tg_ic.shape[1] != n_dim else tg_ic
    else:
        raise ValueError('tg_ic must be 1-D, 2-D or 3-D')
    return np.ascontiguousarray(out)


def restore_fmatrix_axes(recorded_fmatrix, tg_ic_user, n_dim):
    """Undo the axis swap for user-side shapes (integrate.py:527-534 / integrator.py:998-1005)."""
    if tg_ic_user.ndim == 2:
        if recorded_fmatrix.shape[1:3] != tg_ic_user.shape:
            recorded_fmatrix = np.swapaxes(recorded_fmatrix, 1, 2)
    elif tg_ic_user.ndim == 3:
        if tg_ic_user.shape[1] != n_dim:
            if recorded_fmatrix.shape[:3] != tg_ic_user.shape:
                recorded_fmatrix = np.swapaxes(recorded_fmatrix, 1, 2)
    return recorded_fmatrix


def run_rk(f, time, ic, time_direction, write_steps, b, c, a, device=None):
    """One ensemble integration, (n_traj, n_dim, n_records): the fused HIP stepper for tensor tendencies, the host stepper
    for a user-written callable."""
    if on_device(f):
        device = resolve_device(device, np.shape(ic)[0])
        return hip_model_of(f, device=device).rk_integrate(time, ic, time_direction, write_steps, b, c, a)
    return host_stepper.integrate_runge_kutta(f, time, np.ascontiguousarray(ic, dtype=np.float64), time_direction, write_steps, b, c, a)


def run_rk_tgls(f, fjac, time, ic, tg_ic, time_direction, write_steps, b, c, a, adjoint, inverse, boundary, device=None):
    """Trajectories + tangent / adjoint model.  On the device when `f`, `fjac` come from one create_tendencies() call and
    the boundary term is the default zero; otherwise on the host (a `boundary` callable, or user-written `f` / `fjac`)."""
    if on_device(f) and on_device(fjac) and boundary is None:
        device = resolve_device(device, np.shape(ic)[0])
        model = hip_model_of(f, device=device)
        if hip_model_of(fjac, 'fjac', device=device) is not model:
            raise TypeError('f and fjac must come from the same create_tendencies() call')
        return model.rk_tgls_integrate(time, ic, tg_ic, time_direction, write_steps, b, c, a, adjoint, inverse)
    return host_stepper.integrate_runge_kutta_tgls(f, fjac, time, np.ascontiguousarray(ic, dtype=np.float64), tg_ic, time_direction,
                                                   write_steps, b, c, a, adjoint, inverse, boundary)


def integrate_runge_kutta(f, t0, t, dt, ic=None, forward=True, write_steps=1, b=None, c=None, a=None, device=None):
    """Integrate dx/dt = f(t, x) for one state or an ensemble of states; returns ``(time, traj)`` with the
    reference's conventions: traj is ``np.squeeze`` of (n_traj, n_dim, n_records); time is a scalar when
    ``write_steps == 0``.  One keyword beyond the reference: ``device`` (a GPU index, a list of indices or 'all'; see
    `resolve_device`)."""
    ic = normalise_ic(ic, None if ic is not None else dimension_of(f))
    b, c, a = resolve_tableau(b, c, a)
    time = time_grid(t0, t, dt)
    recorded = run_rk(f, time, ic, 1 if forward else -1, write_steps, b, c, a, device=device)
    return record_times(time, write_steps, forward), np.squeeze(recorded)


def integrate_runge_kutta_tgls(f, fjac, t0, t, dt, ic=None, tg_ic=None, forward=True, adjoint=False, inverse=False,
                               boundary=None, write_steps=1, b=None, c=None, a=None, device=None):
    """Integrate the trajectory together with its tangent linear (or adjoint) model; returns
    ``(time, traj, fmatrix)`` like integrate.py:240-552."""
    ic = normalise_ic(ic, None if ic is not None else dimension_of(f))
    n_dim = ic.shape[1]
    tg_user = np.eye(n_dim) if tg_ic is None else np.asarray(tg_ic, dtype=np.float64)
    tg = normalise_tg_ic(tg_user, ic.shape[0], n_dim)
    b, c, a = resolve_tableau(b, c, a)
    time = time_grid(t0, t, dt)
    traj, fm = run_rk_tgls(f, fjac, time, ic, tg, 1 if forward else -1, write_steps, b, c, a, adjoint,
                           -1. if inverse else 1., boundary, device=device)
    fm = restore_fmatrix_axes(fm, tg_user, n_dim)
    return record_times(time, write_steps, forward), np.squeeze(traj), np.squeeze(fm)
