"""qgs_amd -- MI355X (gfx950) implementation of the qgs ensemble tendencies + Runge-Kutta hot path.

Drop-in for that path of Climdyn/qgs: `QgParams` -> `create_tendencies` -> `RungeKuttaIntegrator` /
`RungeKuttaTglsIntegrator`, with the numba loops replaced by hand-written HIP kernels reached through
the C-ABI of include/qgs_hip.h (ctypes).  There is no CPU compute path in this package: every
numerical entry point raises if the HIP library or a GPU is missing.
"""

__version__ = "0.1.0"
