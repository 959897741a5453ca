"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of oracle/libqgs_oracle.so (qgs_oracle.c).

Mirrors the *signatures* of the reference's jitted functions so parity tests read like the
reference: sparse_mul3 / sparse_mul2 (qgs/functions/sparse_mul.py), the f / Df closures
(qgs/functions/tendencies.py:111-121) and the two stepper loops
(qgs/integrators/integrate.py:182-223, :555-614).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_FAST = None

_i64 = ctypes.c_int64
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags='C_CONTIGUOUS')
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags='C_CONTIGUOUS')


def build(force=False):
    so = os.path.join(_HERE, 'libqgs_oracle.so')
    src = os.path.join(_HERE, 'qgs_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s'] + (['-B'] if force else []))
    return so


def build_fast():
    """Performance flavour of the same source for bench.py's `cpu_baseline` leg: -O3 -march=native, FMA contraction
    allowed.  Built on the host that runs it (the flags are host-specific), into a scratch directory."""
    import tempfile
    out = os.path.join(tempfile.mkdtemp(prefix='qgs_oracle_fast_'), 'libqgs_oracle_fast.so')
    subprocess.check_call([os.environ.get('CC', 'gcc'), '-O3', '-march=native', '-ffp-contract=fast', '-fopenmp', '-fPIC',
                           '-std=c11', '-shared', '-o', out, os.path.join(_HERE, 'qgs_oracle.c')])
    return out


def _bind(L):
    L.oracle_sparse_mul3.argtypes = [_i64, _i32p, _f64p, _i64, _f64p, _f64p, _f64p]
    L.oracle_sparse_mul2.argtypes = [_i64, _i32p, _f64p, _i64, _f64p, _f64p]
    L.oracle_sparse_mul5.argtypes = [_i64, _i32p, _f64p, _i64, _f64p, _f64p, _f64p, _f64p, _f64p]
    L.oracle_sparse_mul4.argtypes = [_i64, _i32p, _f64p, _i64, _f64p, _f64p, _f64p, _f64p]
    L.oracle_tendencies_r.argtypes = [ctypes.c_int, _i64, _i64, _i32p, _f64p, _i64, _f64p, _f64p]
    L.oracle_jacobian_r.argtypes = [ctypes.c_int, _i64, _i64, _i32p, _f64p, _i64, _f64p, _f64p]
    L.oracle_n_records.argtypes = [_f64p, _i64, _i64]
    L.oracle_n_records.restype = _i64
    L.oracle_rk_integrate_r.argtypes = [ctypes.c_int, _i64, _i64, _i32p, _f64p, _i64, _f64p, _f64p, _i64, ctypes.c_int, _i64,
                                        ctypes.c_int, _f64p, _f64p, _f64p, _i64, _f64p, ctypes.c_int]
    L.oracle_rk_tgls_integrate_r.argtypes = [ctypes.c_int, _i64, _i64, _i32p, _f64p, _i64, _i32p, _f64p, _i64, _i64, _f64p, _f64p,
                                           _f64p, _i64, ctypes.c_int, _i64, ctypes.c_int, _f64p, _f64p, _f64p,
                                           ctypes.c_int, ctypes.c_double, _i64, _f64p, _f64p, ctypes.c_int]
    L.oracle_max_threads.restype = ctypes.c_int
    for fn in (L.oracle_sparse_mul3, L.oracle_sparse_mul2, L.oracle_sparse_mul5, L.oracle_sparse_mul4,
               L.oracle_tendencies_r, L.oracle_jacobian_r, L.oracle_rk_integrate_r, L.oracle_rk_tgls_integrate_r):
        fn.restype = None
    return L


def lib(flavour='parity'):
    """'parity': oracle/libqgs_oracle.so (-O2 -ffp-contract=off, the checker).  'fast': see build_fast()."""
    global _LIB, _FAST
    if flavour == 'fast':
        if _FAST is None:
            _FAST = _bind(ctypes.CDLL(build_fast()))
        return _FAST
    if _LIB is None:
        _LIB = _bind(ctypes.CDLL(build()))
    return _LIB


def _c(x, dt=np.float64):
    return np.ascontiguousarray(x, dtype=dt)


def max_threads():
    return int(lib().oracle_max_threads())


def sparse_mul3(coo, value, vec_a, vec_b):
    coo, value, vec_a, vec_b = _c(coo, np.int32), _c(value), _c(vec_a), _c(vec_b)
    res = np.empty_like(vec_a)
    lib().oracle_sparse_mul3(len(value), coo, value, len(vec_a), vec_a, vec_b, res)
    return res


def sparse_mul2(coo, value, vec):
    coo, value, vec = _c(coo, np.int32), _c(value), _c(vec)
    res = np.empty((len(vec), len(vec)))
    lib().oracle_sparse_mul2(len(value), coo, value, len(vec), vec, res)
    return res


def sparse_mul5(coo, value, vec_a, vec_b, vec_c, vec_d):
    coo, value = _c(coo, np.int32), _c(value)
    vec_a, vec_b, vec_c, vec_d = _c(vec_a), _c(vec_b), _c(vec_c), _c(vec_d)
    res = np.empty_like(vec_a)
    lib().oracle_sparse_mul5(len(value), coo, value, len(vec_a), vec_a, vec_b, vec_c, vec_d, res)
    return res


def sparse_mul4(coo, value, vec_a, vec_b, vec_c):
    coo, value, vec_a, vec_b, vec_c = _c(coo, np.int32), _c(value), _c(vec_a), _c(vec_b), _c(vec_c)
    res = np.empty((len(vec_a), len(vec_a)))
    lib().oracle_sparse_mul4(len(value), coo, value, len(vec_a), vec_a, vec_b, vec_c, res)
    return res


class OracleModel(object):
    """Holds the COO operands the reference's `f`/`Df` closures capture (tendencies.py:92-96).  The rank of the
    tensor is the width of `coo`: 3 (QgsTensor: sparse_mul3 / sparse_mul2) or 5 (QgsTensorDynamicT / QgsTensorT4:
    sparse_mul5 / sparse_mul4, tendencies.py:98-109)."""

    def __init__(self, ndim, coo, val, jcoo=None, jval=None, flavour='parity'):
        self.ndim = int(ndim)
        self._flavour = flavour
        lib(flavour)
        self.coo, self.val = _c(coo, np.int32), _c(val)
        self.rank = int(self.coo.shape[1])
        assert self.rank in (3, 5)
        self.jcoo = _c(jcoo, np.int32) if jcoo is not None else None
        self.jval = _c(jval) if jval is not None else None

    # f(t, x) / Df(t, x) for a single state or a batch (n_traj, ndim)
    def f(self, t, x):
        x = _c(x)
        xb = x.reshape(-1, self.ndim)
        out = np.empty_like(xb)
        lib(self._flavour).oracle_tendencies_r(self.rank, self.ndim, len(self.val), self.coo, self.val, xb.shape[0], xb, out)
        return out.reshape(x.shape)

    def Df(self, t, x):
        x = _c(x)
        xb = x.reshape(-1, self.ndim)
        out = np.empty((xb.shape[0], self.ndim, self.ndim))
        lib(self._flavour).oracle_jacobian_r(self.rank, self.ndim, len(self.jval), self.jcoo, self.jval, xb.shape[0], xb, out)
        return out[0] if x.ndim == 1 else out

    @staticmethod
    def n_records(time, write_steps):
        time = _c(time)
        return int(lib().oracle_n_records(time, len(time), int(write_steps)))

    def integrate_runge_kutta_jit(self, time, ic, time_direction, write_steps, b, c, a, threads=1):
        """_integrate_runge_kutta_jit(f, time, ic, time_direction, write_steps, b, c, a)"""
        time, ic, b, c, a = _c(time), _c(ic), _c(b), _c(c), _c(a)
        n_traj = ic.shape[0]
        nrec = self.n_records(time, write_steps)
        rec = np.zeros((n_traj, self.ndim, nrec))
        lib(self._flavour).oracle_rk_integrate_r(self.rank, self.ndim, len(self.val), self.coo, self.val, n_traj, ic, time, len(time),
                                  int(time_direction), int(write_steps), len(b), b, c, a, nrec, rec, int(threads))
        return rec

    def integrate_runge_kutta_tgls_jit(self, time, ic, tg_ic, time_direction, write_steps, b, c, a,
                                       adjoint, inverse, threads=1):
        """_integrate_runge_kutta_tgls_jit(f, fjac, time, ic, tg_ic, ..., adjoint, inverse, boundary=zeros)"""
        time, ic, tg_ic, b, c, a = _c(time), _c(ic), _c(tg_ic), _c(b), _c(c), _c(a)
        n_traj, n_tg = ic.shape[0], tg_ic.shape[2]
        nrec = self.n_records(time, write_steps)
        rec = np.zeros((n_traj, self.ndim, nrec))
        recm = np.zeros((n_traj, self.ndim, n_tg, nrec))
        lib(self._flavour).oracle_rk_tgls_integrate_r(self.rank, self.ndim, len(self.val), self.coo, self.val, len(self.jval), self.jcoo,
                                       self.jval, n_traj, n_tg, ic, tg_ic, time, len(time), int(time_direction),
                                       int(write_steps), len(b), b, c, a, int(bool(adjoint)), float(inverse),
                                       nrec, rec, recm, int(threads))
        return rec, recm
