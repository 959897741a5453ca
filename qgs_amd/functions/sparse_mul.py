"""`sparse_mul3` / `sparse_mul2` (rank-3 tensors) and `sparse_mul5` / `sparse_mul4` (rank-5 tensors of the dynamic-T / T4
models) with the reference's call signatures, evaluated on the GPU.

The reference's users write their own tendencies by hand from ``aotensor.tensor.coords.T`` / ``aotensor.tensor.data``
(documentation user_guide.rst:437-458) and call these two functions (qgs/functions/sparse_mul.py:13-81).  Here the
tensor is shipped to the device on first use (a small cache keyed by the operands); there is no host implementation.
The forms the reference itself uses, ``sparse_mul3(coo, val, xx, xx)`` and ``sparse_mul2(jcoo, jval, xx)`` with
``xx[0] == 1`` (the constant slot) and no entries in row 0, run in the same kernels as `f` / `Df`.  Anything else the
reference's functions accept -- different vectors, ``a[0] != 1``, entries in row 0 (sparse_mul.py:48-81, 123-158 take any
arrays) -- goes through the general contraction kernel (`qgs_contraction_*`: explicit vectors, the reference's operation
order, bitwise its loops).
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


def _contraction(n_slots, coo, val, matrix):
    coo = np.ascontiguousarray(coo, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float64)
    key = ('general', bool(matrix), n_slots, hashlib.blake2b(coo.tobytes() + val.tobytes(), digest_size=16).digest())
    m = _CACHE.get(key)
    if m is None:
        m = (_lib.Contraction(n_slots, coo, val, matrix=matrix), None)
        _CACHE[key] = m
        while len(_CACHE) > _CACHE_MAX:
            for old in _CACHE.popitem(last=False)[1]:
                if old is not None:
                    old.close()
    else:
        _CACHE.move_to_end(key)
    return m[0]


def _vectors(vs):
    vs = [np.asarray(v, dtype=np.float64) for v in vs]
    n = vs[0].shape[0] if vs[0].ndim == 1 else -1
    if n < 2 or any(v.ndim != 1 or v.shape[0] != n for v in vs):
        raise ValueError('the vectors must be 1-D arrays of one length ndim+1')
    return vs


def _fast_form(vs, coo):
    """The call the reference itself makes: one vector in every slot, constant slot 1, nothing in row 0 of the tensor."""
    first = vs[0]
    return (first[0] == 1.0 and all(v is first or np.array_equal(v, first) for v in vs[1:])
            and not (np.asarray(coo)[:, 0] == 0).any())


def sparse_mul3(coo, value, vec_a, vec_b):
    """``res[i] = sum_n value[n] * vec_a[j_n] * vec_b[k_n]`` over the COO entries, ``res[0] = 1`` (sparse_mul.py:48-81; same
    parameter names, so keyword calls of the reference carry over).  `coo` is (nnz, 3)."""
    a, b = _vectors((vec_a, vec_b))
    n = a.shape[0] - 1
    if _fast_form((a, b), coo):
        res = np.empty(n + 1)
        res[1:] = _model('mul3', n, coo, value)[0].tendencies(a[1:])
    else:
        res = _contraction(n + 1, coo, value, False).apply(a, b)
    res[0] = 1.
    return res


def sparse_mul2(coo, value, vec):
    """``res[i, j] = sum_n value[n] * vec[k_n]`` -> (ndim+1, ndim+1) (sparse_mul.py:13-45)."""
    vec, = _vectors((vec,))
    n = vec.shape[0] - 1
    if not _fast_form((vec,), coo):
        return _contraction(n + 1, coo, value, True).apply(vec)
    model, col0 = _model('mul2', n, coo, value)
    res = np.zeros((n + 1, n + 1))
    res[1:, 1:] = model.jacobian(vec[1:])
    if col0 is not None:
        res[1:, 0] = col0.tendencies(vec[1:])
    return res


def sparse_mul5(coo, value, vec_a, vec_b, vec_c, vec_d):
    """``res[i] = sum_n value[n] * vec_a[j_n] * vec_b[k_n] * vec_c[l_n] * vec_d[m_n]``, ``res[0] = 1`` (sparse_mul.py:122-158).
    `coo` is (nnz, 5)."""
    vs = _vectors((vec_a, vec_b, vec_c, vec_d))
    n = vs[0].shape[0] - 1
    if _fast_form(vs, coo):
        res = np.empty(n + 1)
        res[1:] = _model('mul5', n, coo, value)[0].tendencies(vs[0][1:])
    else:
        res = _contraction(n + 1, coo, value, False).apply(*vs)
    res[0] = 1.
    return res


def sparse_mul4(coo, value, vec_a, vec_b, vec_c):
    """``res[i, j] = sum_n value[n] * vec_a[k_n] * vec_b[l_n] * vec_c[m_n]`` -> (ndim+1, ndim+1) (sparse_mul.py:84-120)."""
    vs = _vectors((vec_a, vec_b, vec_c))
    n = vs[0].shape[0] - 1
    if not _fast_form(vs, coo):
        return _contraction(n + 1, coo, value, True).apply(*vs)
    model, col0 = _model('mul4', n, coo, value)
    res = np.zeros((n + 1, n + 1))
    res[1:, 1:] = model.jacobian(vs[0][1:])
    if col0 is not None:
        res[1:, 0] = col0.tendencies(vs[0][1:])
    return res
