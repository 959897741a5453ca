"""Integrator classes (reference: qgs/integrators/integrator.py).

`RungeKuttaIntegrator` and `RungeKuttaTglsIntegrator` keep the reference's constructor, attributes and
methods (integrator.py:27-450, 515-1100).  What changes is the engine: the reference fans one task per
trajectory out to `num_threads` worker processes through pickling queues (integrator.py:388-395,
453-512); here the whole ensemble goes to the GPU in one call and every wavefront lane integrates one
member.  `start()` / `terminate()` are kept: they acquire / release the GPU handle instead of processes.

`num_threads` no longer sets a degree of parallelism, but it is kept because `initialize()` draws its
random initial conditions in `num_threads`-sized batches (integrator.py:257-291) and reproducing that
draw order is part of being a drop-in.
"""
import multiprocessing

import numpy as np

from qgs_amd.integrators import integrate as _fn


class _EnsembleIntegrator(object):
    """State shared by the trajectory and the tangent-linear integrators."""

    def __init__(self, num_threads=None, b=None, c=None, a=None, number_of_dimensions=None, device=None):
        self.num_threads = multiprocessing.cpu_count() if num_threads is None else num_threads
        # GPU(s) of the engine: an index, a list of indices (members are sharded over them, a device may repeat), 'all', or
        # None = the device the tendencies were created for; ensembles of >= 131 072 members then take every visible GPU
        # (integrate.resolve_device), as the reference takes every core by default (integrator.py:79-82)
        self.device = device
        self.b, self.c, self.a = _fn.resolve_tableau(b, c, a)
        self.ic = None
        self._time = None
        self._recorded_traj = None
        self.n_traj = 0
        self.n_dim = number_of_dimensions
        self.n_records = 0
        self._write_steps = 0
        self._time_direction = 1
        self.func = None
        self._model = None

    # -- engine life cycle ------------------------------------------------------------------------
    def terminate(self):
        """Release the integrator's hold on the GPU engine (reference: stop the worker processes)."""
        self._model = None

    def start(self):
        """(Re)acquire the GPU engine for the current `func` (reference: restart the worker processes)."""
        self.terminate()
        if self.func is not None and _fn.on_device(self.func):
            self._model = _fn.hip_model_of(self.func, device=_fn.resolve_device(self.device))

    def set_bca(self, b=None, c=None, a=None, ic_init=True):
        """Set the Butcher tableau; `ic_init` resets the stored initial conditions."""
        if a is not None:
            self.a = a
        if b is not None:
            self.b = b
        if c is not None:
            self.c = c
        if ic_init:
            self.ic = None
        self.start()

    # -- helpers ------------------------------------------------------------------------------------
    def _dimension(self):
        if self.n_dim is not None:
            return self.n_dim
        return _fn.dimension_of(self.func)

    def _prepare(self, t0, t, dt, ic, forward, write_steps):
        if ic is None:
            if self.ic is None:
                self.ic = np.zeros(self._dimension())
        else:
            self.ic = ic                       # by reference, like integrator.py:363
        if len(self.ic.shape) == 1:
            self.ic = self.ic.reshape((1, -1))
        self.n_traj = self.ic.shape[0]
        self.n_dim = self.ic.shape[1]
        self._time = _fn.time_grid(t0, t, dt)
        self._write_steps = write_steps
        self._time_direction = 1 if forward else -1
        if write_steps == 0:
            self.n_records = 1
        else:
            tot = self._time[::write_steps]
            self.n_records = len(tot) + (1 if tot[-1] != self._time[-1] else 0)

    def _record_times(self):
        return _fn.record_times(self._time, self._write_steps, self._time_direction == 1)

    def _initial_guess(self, number_of_trajectories, ic):
        """Random / user initial states of `initialize` (integrator.py:240-268); returns (tmp_ic, n, reconverge)."""
        reconverge = False
        if ic is None:
            i = self._dimension()
            if number_of_trajectories > self.num_threads:
                reconverge = True
                tmp_ic = np.zeros((number_of_trajectories, i))
                tmp_ic[:self.num_threads] = np.random.randn(self.num_threads, i)
            else:
                tmp_ic = np.random.randn(number_of_trajectories, i)
        else:
            tmp_ic = ic.copy()
            if len(tmp_ic.shape) > 1:
                number_of_trajectories = tmp_ic.shape[0]
        return tmp_ic, number_of_trajectories, reconverge

    def _final_states(self):
        raise NotImplementedError

    def initialize(self, convergence_time, dt, pert_size=0.01, reconvergence_time=None, forward=True,
                   number_of_trajectories=1, ic=None, reconverge=False):
        """Put `number_of_trajectories` initial conditions on the attractor (integrator.py:198-295): integrate
        random states over `convergence_time`; beyond `num_threads` trajectories, grow the set by perturbing
        converged states (`pert_size`) and re-converging them over `reconvergence_time`."""
        if reconverge is None:
            reconverge = False
        tmp_ic, number_of_trajectories, forced = self._initial_guess(number_of_trajectories, ic)
        reconverge = reconverge or forced

        if reconverge and reconvergence_time is not None:
            nt = self.num_threads
            self.integrate(0., convergence_time, dt, ic=tmp_ic[:nt], write_steps=0, forward=forward)
            x = self._final_states()
            tmp_ic[:nt] = x
            next_len = nt if number_of_trajectories - nt > nt else number_of_trajectories - nt
            index = nt
            while True:
                perturbation = pert_size * np.random.randn(next_len, x.shape[1])
                self.integrate(0., reconvergence_time, dt, ic=x[:next_len] + perturbation, write_steps=0, forward=forward)
                x = self._final_states()
                tmp_ic[index:index + next_len] = x
                index += next_len
                next_len = nt if number_of_trajectories - index > nt else number_of_trajectories - index
                if next_len <= 0:
                    break
            self.ic = tmp_ic
        else:
            self.integrate(0., convergence_time, dt, ic=tmp_ic, write_steps=0, forward=forward)
            self.ic = self._final_states()

    def get_ic(self):
        return self.ic

    def set_ic(self, ic):
        self.ic = ic


class RungeKuttaIntegrator(_EnsembleIntegrator):
    """Integrate dx/dt = f(t, x) for an ensemble of initial conditions with an explicit Runge-Kutta scheme.

    Parameters, attributes and methods as in the reference (integrator.py:27-450):
    ``RungeKuttaIntegrator(num_threads=None, b=None, c=None, a=None, number_of_dimensions=None)``; attributes
    ``num_threads, b, c, a, n_dim, n_traj, n_records, ic, func``.  One keyword beyond the reference: ``device`` -- a GPU
    index, a list of indices or ``'all'``: the members are split into contiguous shards, one per listed GPU, integrated
    concurrently in this process, and every GPU delivers its slice of the result array itself (`_lib.HipModelGroup`).  Left at
    None, ensembles of at least 131 072 members use every visible GPU, smaller ones the GPU the tendencies were created for.
    (Multi-process jobs, one rank per GPU with an RCCL gather: ``device=LOCAL_RANK``, see qgs_amd/parallel.py.)
    """

    def set_func(self, f, ic_init=True):
        """Set the tendencies `f` (from `create_tendencies`)."""
        self.func = f
        if ic_init:
            self.ic = None
        self.start()

    def integrate(self, t0, t, dt, ic=None, forward=True, write_steps=1):
        """Integrate from `t0` to `t` with step `dt`; `ic` (n_dim,) or (n_traj, n_dim); results via
        `get_trajectories()`.  Backward integration (`forward=False`) starts from `t`."""
        if self.func is None:
            print('No function to integrate defined!')
            return 0
        self._prepare(t0, t, dt, ic, forward, write_steps)
        # tensor tendencies: the fused HIP stepper; a user-written callable (integrator.py:1237-1256): the host stepper
        self._recorded_traj = _fn.run_rk(self.func, self._time, self.ic, self._time_direction, write_steps,
                                         self.b, self.c, self.a, device=self.device)

    def integrate_moments(self, t0, t, dt, ic=None, forward=True, write_steps=1, variance=True):
        """Integrate like `integrate`, but bring back only the ensemble mean and variance over the members:
        returns ``(time, mean, var)`` with mean / var of shape (n_dim, n_records) (``var`` is None when
        ``variance=False``); equal to ``np.mean(traj, axis=0)`` / ``np.var(traj, axis=0)`` of the (n_traj, n_dim,
        n_records) array that `integrate` + `get_trajectories` would return, which here stays on the device.
        (Device-side form of the member averages of qgs/integrators/statistics.py:55-63.)"""
        if self.func is None:
            print('No function to integrate defined!')
            return 0
        if not _fn.on_device(self.func):
            raise TypeError('integrate_moments reduces on the device: it needs tendencies from create_tendencies()')
        self._prepare(t0, t, dt, ic, forward, write_steps)
        model = _fn.hip_model_of(self.func, device=_fn.resolve_device(self.device, self.n_traj))
        mean, var, fin = model.rk_integrate_moments(self._time, self.ic, self._time_direction, write_steps,
                                                          self.b, self.c, self.a, variance=variance, final_states=True)
        self.last_final_states = fin                           # (n_traj, n_dim): e.g. the next window's initial conditions
        return self._record_times(), mean, var

    def get_trajectories(self):
        """``(time, traj)``: traj is ``np.squeeze`` of (n_traj, n_dim, n_records); time is a scalar when the last
        integration used ``write_steps=0`` (integrator.py:397-424)."""
        return self._record_times(), np.squeeze(self._recorded_traj)

    def _final_states(self):
        return self.get_trajectories()[1]


class RungeKuttaTglsIntegrator(_EnsembleIntegrator):
    """Integrate the trajectories together with their tangent linear / adjoint model (integrator.py:515-1100)."""

    def __init__(self, num_threads=None, b=None, c=None, a=None, number_of_dimensions=None, device=None):
        super(RungeKuttaTglsIntegrator, self).__init__(num_threads, b, c, a, number_of_dimensions, device)
        self.tg_ic = None
        self._recorded_fmatrix = None
        self.n_tg_traj = 0
        self._adjoint = False
        self._boundary = None
        self._inverse = 1.
        self.func_jac = None

    def set_func(self, f, fjac, ic_init=True):
        """Set the tendencies `f` and their Jacobian `fjac` (both from one `create_tendencies` call)."""
        self.func = f
        self.func_jac = fjac
        if ic_init:
            self.ic = None
        self.start()

    def start(self):
        super(RungeKuttaTglsIntegrator, self).start()
        if self._model is not None and self.func_jac is not None and _fn.on_device(self.func_jac):
            if _fn.hip_model_of(self.func_jac, 'fjac', device=_fn.resolve_device(self.device)) is not self._model:
                raise TypeError('f and fjac must come from the same create_tendencies() call')

    def integrate(self, t0, t, dt, ic=None, tg_ic=None, forward=True, adjoint=False, inverse=False, boundary=None,
                  write_steps=1):
        """As `RungeKuttaIntegrator.integrate`, plus `tg_ic` (None = identity; (n_dim,), (n_tg, n_dim),
        (n_traj, n_dim) or 3-D), `adjoint` (propagate with J^T), `inverse` (flip the sign of the tangent
        tendencies) and `boundary` (a callable ``boundary(t, x)`` adding an inhomogeneous term to the tangent tendencies,
        integrate.py:600-603; None = zero).  With ``boundary=None`` and tensor tendencies everything runs on the device; a
        boundary callable is evaluated on the host, stage by stage (host_stepper.py)."""
        if self.func is None or self.func_jac is None:
            print('No function to integrate defined!')
            return 0
        self._prepare(t0, t, dt, ic, forward, write_steps)
        tg_user = np.eye(self.n_dim) if tg_ic is None else np.asarray(tg_ic, dtype=np.float64)
        # NOTE the reference leaves `self.tg_ic` untouched for a 3-D tg_ic that is already (n_traj, n_dim, n_tg)
        # (integrator.py:955-958, a latent bug); here it is set in every case.
        self.tg_ic = _fn.normalise_tg_ic(tg_user, self.n_traj, self.n_dim)
        self.n_tg_traj = self.tg_ic.shape[1]          # sic: the reference stores shape[1] (integrator.py:960)
        self._adjoint = adjoint
        self._boundary = boundary
        self._inverse = -1. if inverse else 1.
        traj, fm = _fn.run_rk_tgls(self.func, self.func_jac, self._time, self.ic, self.tg_ic, self._time_direction, write_steps,
                                   self.b, self.c, self.a, adjoint, self._inverse, boundary, device=self.device)
        self._recorded_traj = traj
        self._recorded_fmatrix = _fn.restore_fmatrix_axes(fm, tg_user, self.n_dim)

    def get_trajectories(self):
        """``(time, traj, fmatrix)`` (integrator.py:1007-1049)."""
        return self._record_times(), np.squeeze(self._recorded_traj), np.squeeze(self._recorded_fmatrix)

    def _final_states(self):
        return self.get_trajectories()[1]

    def get_tg_ic(self):
        return self.tg_ic

    def set_tg_ic(self, tg_ic):
        self.tg_ic = tg_ic
