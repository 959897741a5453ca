"""TEST INFRASTRUCTURE ONLY -- dense-ndarray-backed stand-in for the subset of
pydata/sparse that the reference's *analytic* tensor assembly uses, so that
/root/reference can be imported in the build container (the real package is not
installed and there is no network).

Semantics kept: COO(coords, data) sums duplicates; `.coords` is lexicographic
(C-order of the linearised index, as pydata/sparse sorts it); `.data` follows.
Difference: explicit zeros are not stored (irrelevant to the value of f / Df).

Used only by tests/golden/make_golden.py.  Never imported by qgs_amd.
"""
import numpy as np


def _raw(x):
    return x._a if isinstance(x, _Base) else np.asarray(x)


def _box(r):
    r = np.asarray(r)
    if r.ndim == 0:
        return float(r)
    return COO(r)


class _Base(object):
    __array_priority__ = 100

    shape = property(lambda self: self._a.shape)
    ndim = property(lambda self: self._a.ndim)
    dtype = property(lambda self: self._a.dtype)
    nnz = property(lambda self: int(np.count_nonzero(self._a)))
    coords = property(lambda self: np.array(np.nonzero(self._a)))
    data = property(lambda self: self._a[np.nonzero(self._a)])
    T = property(lambda self: COO(self._a.T.copy()))

    def todense(self):
        return self._a.copy()

    def to_coo(self):
        return COO(self._a.copy())

    def copy(self):
        return COO(self._a.copy())

    def astype(self, t):
        return COO(self._a.astype(t))

    def swapaxes(self, i, j):
        return COO(np.swapaxes(self._a, i, j).copy())

    def __getitem__(self, key):
        return _box(self._a[key])

    def __matmul__(self, other):
        return _box(self._a @ _raw(other))

    def __rmatmul__(self, other):
        return _box(_raw(other) @ self._a)

    def __add__(self, other):
        return _box(self._a + _raw(other))

    __radd__ = __add__

    def __sub__(self, other):
        return _box(self._a - _raw(other))

    def __rsub__(self, other):
        return _box(_raw(other) - self._a)

    def __neg__(self):
        return COO(-self._a)

    def __mul__(self, other):
        return _box(self._a * _raw(other))

    __rmul__ = __mul__

    def __truediv__(self, other):
        return _box(self._a / _raw(other))


class COO(_Base):
    def __init__(self, coords, data=None, shape=None, prune=False, **kwargs):
        if data is None:
            self._a = np.array(_raw(coords), dtype=float)
            return
        coords = np.asarray(coords, dtype=int)
        data = np.asarray(data, dtype=float)
        if shape is None:
            shape = tuple(int(c.max()) + 1 for c in coords)
        dense = np.zeros(shape)
        np.add.at(dense, tuple(coords), data)
        self._a = dense


class DOK(_Base):
    def __init__(self, shape, dtype=float):
        self._a = np.zeros(shape, dtype=dtype)

    def __setitem__(self, key, value):
        self._a[key] = _raw(value) if isinstance(value, _Base) else value

    def __getitem__(self, key):
        r = self._a[key]
        return float(r) if np.ndim(r) == 0 else COO(r)


def zeros(shape, dtype=float, format='coo'):
    if format == 'dok':
        return DOK(shape, dtype)
    return COO(np.zeros(shape, dtype=dtype))


def tensordot(a, b, axes=2):
    return _box(np.tensordot(_raw(a), _raw(b), axes=axes))
