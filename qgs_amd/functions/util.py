"""Small helpers (reference: qgs/functions/util.py)."""
import numpy as np


def add_to_dict(dic, loc, value):
    """`dic[loc] += value`, creating the key when it is new (util.py:14-31)."""
    if loc in dic:
        dic[loc] += value
    else:
        dic[loc] = value
    return dic


def reverse(a):
    """Reversed copy of a 1-D array (util.py:34-53; used for backward time axes)."""
    return np.ascontiguousarray(np.asarray(a)[::-1])


def normalize_matrix_columns(a):
    """Columns scaled to unit 2-norm; returns ``(normalised, norms)`` (util.py:56-75).  Also takes a stack of matrices
    (..., rows, cols) -- the covariant Lyapunov estimator normalises all members at once."""
    a = np.asarray(a, dtype=np.float64)
    norm = np.sqrt(np.sum(a * a, axis=-2))
    return a / norm[..., None, :], norm


def solve_triangular_matrix(a, b):
    """x with a x = b for upper-triangular square `a`, `b`: column i from the leading i x i block with `np.linalg.solve`, as the
    reference does it (util.py:78-98).  Also takes stacks (..., n, n)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    x = np.zeros_like(b)
    for i in range(2, a.shape[-1] + 1):
        x[..., :i, i - 1] = np.linalg.solve(a[..., :i, :i], b[..., :i, i - 1][..., None])[..., 0]
    x[..., 0, 0] = b[..., 0, 0] / a[..., 0, 0]
    return x
