"""Ensemble statistics helper (reference: qgs/integrators/statistics.py:1-77).

`TrajectoriesStatistics` batches an ensemble of initial conditions through an integrator (`num` batches)
and averages user functions of the trajectories over the members.  The integrator is a
`qgs_amd.integrators.integrator.RungeKuttaIntegrator`, so each batch is one GPU call; the user functions
run on the host on the (n_traj, n_dim, n_records) array, as in the reference.
"""
import numpy as np


class TrajectoriesStatistics(object):
    def __init__(self):
        self.ic = None
        self.integrator = None
        self.func_list = list()
        self.mean_func = list()

    def set_integrator(self, integrator):
        self.integrator = integrator

    def set_func_list(self, func_list):
        self.func_list = func_list

    def set_ic(self, ic):
        self.ic = ic

    def get_ic(self):
        return self.ic

    def get_stats(self):
        return self.mean_func

    def initialize(self, convergence_time, dt, pert_size=0.01, reconvergence_time=None, number_of_trajectories=1, ic=None):
        """Put the ensemble on the attractor with the integrator's `initialize` (statistics.py:16-21)."""
        self.integrator.initialize(convergence_time, dt, pert_size=pert_size, reconvergence_time=reconvergence_time,
                                   number_of_trajectories=number_of_trajectories, ic=ic)
        self.ic = self.integrator.get_ic()

    def compute_stats(self, t0, t, dt, ic=None, forward=True, write_steps=1, num=1):
        """Member-average of every function of `func_list`, computed over `num` batches of the ensemble and then
        averaged over the batches (statistics.py:33-66; the last batch takes the remainder, and -- as in the
        reference -- batch 0 is integrated first and the last batch last, which coincide when num == 1)."""
        if ic is not None:
            self.set_ic(ic)
        sub = self.ic.shape[0] // num
        batches = [(0, slice(0, sub))] + [(i, slice(i * sub, (i + 1) * sub)) for i in range(1, num - 1)] \
            + [(num - 1, slice((num - 1) * sub, None))]
        realizations = None
        for i, sl in batches:
            self.integrator.integrate(t0, t, dt, ic=self.ic[sl], forward=forward, write_steps=write_steps)
            _, traj = self.integrator.get_trajectories()
            if realizations is None:
                realizations = np.zeros((len(self.func_list), num, traj.shape[1], traj.shape[2]))
            for j, f in enumerate(self.func_list):
                realizations[j, i] = np.mean(f(traj), axis=0)
        self.mean_func = np.mean(realizations, axis=1)

    def compute_moments(self, t0, t, dt, ic=None, forward=True, write_steps=1, num=1):
        """Ensemble mean and variance of the state at every record, reduced on the device: the same numbers as
        `compute_stats` with ``func_list = [lambda x: x, lambda x: x**2]`` (variance = second moment - mean**2), without
        moving the (n_traj, n_dim, n_records) trajectories to the host.  `num` batches as in `compute_stats`; the batch
        moments are combined with their member counts.  Returns ``(time, mean, var)``, both (n_dim, n_records)."""
        if ic is not None:
            self.set_ic(ic)
        n = self.ic.shape[0]
        sub = n // num
        bounds = [(i * sub, (i + 1) * sub if i < num - 1 else n) for i in range(num)]
        m1 = m2 = None
        times = None
        for lo, hi in bounds:
            if hi <= lo:
                continue
            times, mean, var = self.integrator.integrate_moments(t0, t, dt, ic=self.ic[lo:hi], forward=forward,
                                                                 write_steps=write_steps)
            w = (hi - lo) / n
            m1 = w * mean if m1 is None else m1 + w * mean
            m2 = w * (var + mean ** 2) if m2 is None else m2 + w * (var + mean ** 2)
        return times, m1, m2 - m1 ** 2
