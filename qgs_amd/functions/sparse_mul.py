"""`sparse_mul3` / `sparse_mul2` (rank-3 tensors) and `sparse_mul5` / `sparse_mul4` (rank-5 tensors of the dynamic-T / T4
models) with the reference's call signatures, evaluated on the GPU.

The reference's users write their own tendencies by hand from ``aotensor.tensor.coords.T`` / ``aotensor.tensor.data``
(documentation user_guide.rst:437-458) and call these two functions (qgs/functions/sparse_mul.py:13-81).  Here the
tensor is shipped to the device on first use (a small cache keyed by the operands) and the contraction runs in the
same kernels as `f` / `Df`; there is no host implementation.  Supported: the forms the reference itself uses,
``sparse_mul3(coo, val, xx, xx)`` and ``sparse_mul2(jcoo, jval, xx)`` with ``xx[0] == 1`` (the constant slot).
For ensembles call `f` / `Df` from `create_tendencies` / `tendencies_from_tensor` with a 2-D state instead.
"""
import hashlib
from collections import OrderedDict

import numpy as np

from qgs_amd import _lib

_CACHE = OrderedDict()
_CACHE_MAX = 8


def _model(kind, ndim, coo, val):
    coo = np.ascontiguousarray(coo, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float64)
    key = (kind, ndim, hashlib.blake2b(coo.tobytes() + val.tobytes(), digest_size=16).digest())
    m = _CACHE.get(key)
    if m is None:
        if kind in ('mul3', 'mul5'):
            m = (_lib.HipModel(ndim, coo, val, None, None), None)
        else:
            # column 0 of the result, sum_k T_i0k v_k, is a tendencies evaluation of the entries with j == 0
            sel = coo[:, 1] == 0
            col0 = _lib.HipModel(ndim, coo[sel], val[sel], None, None) if sel.any() else None
            zero = np.zeros((0, coo.shape[1]), dtype=np.int32), np.zeros(0)
            m = (_lib.HipModel(ndim, zero[0], zero[1], coo, val), col0)
        _CACHE[key] = m
        while len(_CACHE) > _CACHE_MAX:
            for old in _CACHE.popitem(last=False)[1]:
                if old is not None:
                    old.close()
    else:
        _CACHE.move_to_end(key)
    return m


def _check_vec(v, name):
    v = np.asarray(v, dtype=np.float64)
    if v.ndim != 1 or v.shape[0] < 2:
        raise ValueError('%s must be a 1-D array of length ndim+1' % name)
    if v[0] != 1.0:
        raise NotImplementedError('%s[0] must be 1 (the constant slot of the qgs tensors)' % name)
    return v


def sparse_mul3(coo, val, a, b):
    """``res[i] = sum_n val[n] * a[j_n] * b[k_n]`` over the COO entries, ``res[0] = 1`` (sparse_mul.py:48-81).
    `coo` is (nnz, 3); `a` and `b` must be the same vector (the only form the reference uses)."""
    a = _check_vec(a, 'a')
    b = _check_vec(b, 'b')
    if a is not b and not np.array_equal(a, b):
        raise NotImplementedError('sparse_mul3 with a != b is not available on the device')
    n = a.shape[0] - 1
    res = np.empty(n + 1)
    res[0] = 1.
    res[1:] = _model('mul3', n, coo, val)[0].tendencies(a[1:])
    return res


def sparse_mul2(coo, val, vec):
    """``res[i, j] = sum_n val[n] * vec[k_n]`` -> (ndim+1, ndim+1) (sparse_mul.py:13-45); row 0 is zero apart from
    what the tensor holds there (the qgs tensors hold nothing)."""
    vec = _check_vec(vec, 'vec')
    n = vec.shape[0] - 1
    coo = np.asarray(coo)
    if (coo[:, 0] == 0).any():
        raise NotImplementedError('entries in row 0 are not supported')
    model, col0 = _model('mul2', n, coo, val)
    res = np.zeros((n + 1, n + 1))
    res[1:, 1:] = model.jacobian(vec[1:])
    if col0 is not None:
        res[1:, 0] = col0.tendencies(vec[1:])
    return res


def _same(vectors, what):
    first = vectors[0]
    for v in vectors[1:]:
        if v is not first and not np.array_equal(v, first):
            raise NotImplementedError('%s with different vectors is not available on the device' % what)
    return first


def sparse_mul5(coo, val, a, b, c, d):
    """``res[i] = sum_n val[n] * a[j_n] * b[k_n] * c[l_n] * d[m_n]``, ``res[0] = 1`` (sparse_mul.py:123-158).
    `coo` is (nnz, 5); the four vectors must be the same one (tendencies.py:100-103)."""
    x = _same([_check_vec(v, nm) for v, nm in ((a, 'a'), (b, 'b'), (c, 'c'), (d, 'd'))], 'sparse_mul5')
    n = x.shape[0] - 1
    res = np.empty(n + 1)
    res[0] = 1.
    res[1:] = _model('mul5', n, coo, val)[0].tendencies(x[1:])
    return res


def sparse_mul4(coo, val, a, b, c):
    """``res[i, j] = sum_n val[n] * a[k_n] * b[l_n] * c[m_n]`` -> (ndim+1, ndim+1) (sparse_mul.py:84-120), the three
    vectors being the same one (tendencies.py:105-109)."""
    x = _same([_check_vec(v, nm) for v, nm in ((a, 'a'), (b, 'b'), (c, 'c'))], 'sparse_mul4')
    n = x.shape[0] - 1
    coo = np.asarray(coo)
    if (coo[:, 0] == 0).any():
        raise NotImplementedError('entries in row 0 are not supported')
    model, col0 = _model('mul4', n, coo, val)
    res = np.zeros((n + 1, n + 1))
    res[1:, 1:] = model.jacobian(x[1:])
    if col0 is not None:
        res[1:, 0] = col0.tendencies(x[1:])
    return res
