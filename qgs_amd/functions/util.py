"""Small helpers (reference: qgs/functions/util.py)."""
import numpy as np


def reverse(a):
    """Reversed copy of a 1-D array (util.py:34-53; used for backward time axes)."""
    return np.ascontiguousarray(np.asarray(a)[::-1])
