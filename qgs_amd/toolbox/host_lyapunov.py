"""Benettin steps for user-written Python systems (reference: qgs/toolbox/lyapunov.py:471-632 and the forward part of
:1174-1247).  The reference's estimators take any numba-jitted pair `f`, `fjac` -- its own example is the Lorenz-84 system written
by hand (lyapunov.py:1334-1397).  A Python function cannot be called from a wavefront, so for such a pair the loops run here in
NumPy on the host steppers of qgs_amd/integrators/host_stepper.py, all trajectories together; tensor tendencies never reach
this module (qgs_amd/toolbox/lyapunov.py runs them on the GPU).  Not the parity checker of the test suite, not part of any timed
path.
"""
import numpy as np

from qgs_amd.integrators import host_stepper
from qgs_amd.functions.util import reverse


def _sub(tt, d, mdt):
    return np.concatenate((np.arange(tt, tt + d, mdt), np.full((1,), tt + d)))


def benettin(f, fjac, pretime, time, mdt, ic, n_vec, write_steps, forward, adjoint, inverse, tableau, a0, fine_base=False):
    """Backward (`forward=False`) or forward Lyapunov vectors of every trajectory started from `ic` (n, n_dim): spin-up over
    `pretime` (backward vectors) or backward in time over `time` (forward vectors), records on the other grid.

    Returns a dict: traj (n, n_dim, n_records), vec (n, n_dim, n_vec, n_records), exp (n, n_vec, n_records), and for the
    backward vectors r (n, n_vec, n_vec, n_records): the R of the QR step that FOLLOWS each record (zeros behind the last), and
    junction (n, n_dim): the states at the first recorded time.  `a0`: the (n, n_dim, n_vec) start matrices.  `fine_base`: the
    base trajectory advances with the `mdt` sub-steps (lyapunov.py:1203-1247) instead of one step per interval (:474, :558)."""
    b, c, a = tableau
    ic = np.asarray(ic, dtype=np.float64)
    n, nd = ic.shape
    nv = n_vec
    rec_grid = pretime if forward else time
    nr = host_stepper.n_records(rec_grid, write_steps)
    coarse = np.concatenate((pretime[:-1], time))
    at = np.arange(len(coarse))
    grid = coarse
    if fine_base:
        pieces = []
        for i in range(len(coarse) - 1):
            pieces.append(np.arange(coarse[i], coarse[i] + (coarse[i + 1] - coarse[i]), mdt))
            at[i + 1] = at[i] + len(pieces[-1])
        grid = np.concatenate(pieces + [coarse[-1:]])
    base = host_stepper.integrate_runge_kutta(f, grid, ic, 1, 1, b, c, a)            # every grid point recorded
    out = dict(traj=np.zeros((n, nd, nr)), vec=np.zeros((n, nd, nv, nr)), exp=np.zeros((n, nv, nr)),
               r=np.zeros((n, nv, nv, nr)), junction=None)
    q, r = np.linalg.qr(np.asarray(a0, dtype=np.float64))
    state = dict(q=q, r=r)

    def propagate(y_index, sub, direction):
        """q <- Q of QR(TL_sub(q)) along the trajectory started at the base state `y_index`; keeps that QR's R"""
        _, fm = host_stepper.integrate_runge_kutta_tgls(f, fjac, sub, base[:, :, at[y_index]], state['q'], direction, 0, b, c, a,
                                                        adjoint, inverse)
        state['q'], state['r'] = np.linalg.qr(fm[:, :, :, 0])

    def exponents(d):                                        # m_exp = log|diag r| / dt with the R of the step before (:612, :529)
        return np.log(np.abs(np.diagonal(state['r'], axis1=1, axis2=2))) / d

    def record(iw, y_index, m_exp):
        out['traj'][:, :, iw] = base[:, :, at[y_index]]
        out['vec'][:, :, :, iw] = state['q']
        out['exp'][:, :, iw] = m_exp
    n_pre = len(pretime)
    m_exp = np.zeros((n, nv))
    if not forward:
        for ti in range(len(pretime) - 1):
            propagate(ti, _sub(pretime[ti], pretime[ti + 1] - pretime[ti], mdt), 1)
        out['junction'] = base[:, :, at[n_pre - 1]].copy()
        iw = 0
        for ti in range(len(time) - 1):
            d = time[ti + 1] - time[ti]
            m_exp = exponents(d)
            recorded = write_steps > 0 and ti % write_steps == 0
            if recorded:
                record(iw, n_pre - 1 + ti, m_exp)
            propagate(n_pre - 1 + ti, _sub(time[ti], d, mdt), 1)
            if recorded:
                out['r'][:, :, :, iw] = state['r']
                iw += 1
        record(nr - 1, len(coarse) - 1, m_exp)              # (the exponents of the last interval's start, :629-631)
    else:
        tim, post = pretime, time                            # the reference's (time, posttime)
        rpost, rtim = reverse(post), reverse(tim)
        n_t = len(tim)

        def sub_back(tt, d):                                 # d < 0: the sub-steps of [tt + d, tt], integrated against time
            return np.concatenate((np.arange(tt + d, tt, mdt), np.full((1,), tt)))
        for ti in range(len(rpost) - 1):
            tt, d = rpost[ti], rpost[ti + 1] - rpost[ti]
            propagate(n_t - 1 + (len(post) - 1 - ti), sub_back(tt, d), -1)
        iw, y_idx = nr - 1, n_t - 1
        for ti in range(len(rtim) - 1):
            tt, d = rtim[ti], rtim[ti + 1] - rtim[ti]
            y_idx = n_t - 1 - ti
            m_exp = exponents(d)
            if write_steps > 0 and ti % write_steps == 0:
                record(iw, y_idx, m_exp)
                iw -= 1
            propagate(y_idx, sub_back(tt, d), -1)
        record(0, y_idx, m_exp)                              # (state of the last interval's start, :549-551)
    return out
