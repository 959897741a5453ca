"""A user-written system for the integrators: Lorenz-84 with its Jacobian and a boundary term for the tangent model, the
system of the reference's own usage example (qgs/integrators/integrator.py:1230-1256, 1285-1287).  Plain Python: no tensor."""
import numpy as np

a, F, G, b = 0.25, 16., 3., 6.


def fL84(t, x):
    xx = -x[1] ** 2 - x[2] ** 2 - a * x[0] + a * F
    yy = x[0] * x[1] - b * x[0] * x[2] - x[1] + G
    zz = b * x[0] * x[1] + x[0] * x[2] - x[2]
    return np.array([xx, yy, zz])


def DfL84(t, x):
    return np.array([[-a, -2. * x[1], -2. * x[2]],
                     [x[1] - b * x[2], -1. + x[0], -b * x[0]],
                     [b * x[1] + x[2], b * x[0], -1. + x[0]]])


def tboundary(t, x):
    return np.array([0., x[1], 0.])


def rp20_boundary(t, x):
    return 0.01 * x
