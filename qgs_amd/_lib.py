"""ctypes binding of libqgs_hip.so (the C-ABI of include/qgs_hip.h) and the `HipModel` handle.

This is the only place where the package crosses into native code.  PyTorch is *not* needed
here: the host-layout entry points take NumPy arrays; the device-layout entry points take raw
device pointers (ints), which callers obtain from torch tensors (`tensor.data_ptr()`).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libqgs_hip.so')

_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags='C_CONTIGUOUS')
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags='C_CONTIGUOUS')
_i64 = ctypes.c_int64
_int = ctypes.c_int
_vp = ctypes.c_void_p
_dbl = ctypes.c_double

#: every symbol include/qgs_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    'qgs_last_error': (ctypes.c_char_p, []),
    'qgs_backend_info': (_int, [ctypes.POINTER(_int), ctypes.c_char_p, _int]),
    'qgs_model_create': (_int, [_int, _int, _i64, _vp, _vp, _i64, _vp, _vp, ctypes.POINTER(_vp)]),
    'qgs_model_create_rank': (_int, [_int, _int, _int, _i64, _vp, _vp, _i64, _vp, _vp, ctypes.POINTER(_vp)]),
    'qgs_model_destroy': (_int, [_vp]),
    'qgs_model_info': (_i64, [_vp, _int]),
    'qgs_model_set_kernel': (_int, [_vp, _int]),
    'qgs_tendencies': (_int, [_vp, _i64, _f64p, _f64p]),
    'qgs_jacobian': (_int, [_vp, _i64, _f64p, _f64p]),
    'qgs_n_records': (_i64, [_f64p, _i64, _i64]),
    'qgs_record_window': (_int, [_i64, _i64, _i64, _int, _i64, _i64, ctypes.POINTER(_i64)]),
    'qgs_rk_integrate': (_int, [_vp, _i64, _f64p, _f64p, _i64, _int, _i64, _int, _f64p, _f64p, _f64p, _f64p]),
    'qgs_rk_tgls_integrate': (_int, [_vp, _i64, _i64, _f64p, _f64p, _f64p, _i64, _int, _i64, _int, _f64p, _f64p, _f64p,
                                     _int, _dbl, _f64p, _f64p]),
    'qgs_group_create': (_int, [_int, ctypes.POINTER(_int), _int, _int, _i64, _vp, _vp, _i64, _vp, _vp, ctypes.POINTER(_vp)]),
    'qgs_group_destroy': (_int, [_vp]),
    'qgs_group_size': (_int, [_vp]),
    'qgs_group_model': (_vp, [_vp, _int]),
    'qgs_group_shard': (_int, [_vp, _i64, _int, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    'qgs_group_set_kernel': (_int, [_vp, _int]),
    'qgs_group_tendencies': (_int, [_vp, _i64, _f64p, _f64p]),
    'qgs_group_jacobian': (_int, [_vp, _i64, _f64p, _f64p]),
    'qgs_group_rk_integrate': (_int, [_vp, _i64, _f64p, _f64p, _i64, _int, _i64, _int, _f64p, _f64p, _f64p, _f64p]),
    'qgs_group_rk_tgls_integrate': (_int, [_vp, _i64, _i64, _f64p, _f64p, _f64p, _i64, _int, _i64, _int, _f64p, _f64p, _f64p,
                                           _int, _dbl, _f64p, _f64p]),
    'qgs_rk_integrate_rows_device': (_int, [_vp, _i64, _vp, _f64p, _i64, _int, _i64, _int, _f64p, _f64p, _f64p, _vp]),
    'qgs_contraction_create': (_int, [_int, _int, _int, _int, _i64, _vp, _vp, ctypes.POINTER(_vp)]),
    'qgs_contraction_apply': (_int, [_vp, _f64p, _f64p]),
    'qgs_contraction_destroy': (_int, [_vp]),
    'qgs_host_alloc': (_int, [_i64, ctypes.POINTER(_vp)]),
    'qgs_host_free': (_int, [_vp]),
    'qgs_host_register': (_int, [_vp, _i64]),
    'qgs_memcpy_h2d': (_int, [_int, _vp, _vp, _i64, _vp]),
    'qgs_memcpy_d2h': (_int, [_int, _vp, _vp, _i64, _vp]),
    'qgs_host_unregister': (_int, [_vp]),
    'qgs_pack_states': (_int, [_vp, _i64, _i64, _vp, _vp, _vp]),
    'qgs_unpack_states': (_int, [_vp, _i64, _i64, _vp, _vp, _vp]),
    'qgs_pack_tangent': (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    'qgs_local_exponents_device': (_int, [_vp, _i64, _vp, ctypes.c_double, _vp, _vp]),
    'qgs_unpack_records': (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp]),
    'qgs_unpack_window': (_int, [_vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp]),
    'qgs_unpack_window_enqueue': (_int, [_vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp]),
    'qgs_drain_wait': (_int, [_vp]),
    'qgs_tendencies_device': (_int, [_vp, _i64, _i64, _vp, _vp, _vp]),
    'qgs_rk_integrate_device': (_int, [_vp, _i64, _i64, _vp, _f64p, _i64, _int, _i64, _int, _f64p, _f64p, _f64p, _vp, _vp]),
    'qgs_rk_tgls_integrate_device': (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _f64p, _i64, _int, _i64, _int, _f64p, _f64p,
                                            _f64p, _int, _dbl, _vp, _vp, _vp]),
    'qgs_batched_qr_device': (_int, [_vp, _i64, _i64, _int, _int, _vp, _vp, _vp]),
    'qgs_kernel_clock': (_int, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    'qgs_fp64_fma_rate': (_int, [_int, ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    'qgs_ensemble_moments_device': (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    'qgs_batched_matmul_device': (_int, [_vp, _i64, _i64, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp]),
    'qgs_clv_backstep_device': (_int, [_vp, _i64, _i64, _int, _vp, _vp, _vp, _vp, _vp, ctypes.c_double, _vp]),
    'qgs_rk_integrate_moments': (_int, [_vp, _i64, _f64p, _f64p, _i64, _int, _i64, _int, _f64p, _f64p, _f64p, _f64p, _vp, _vp]),
    'qgs_last_kernel_info': (_int, [_vp, ctypes.c_char_p, _int, ctypes.POINTER(_int), ctypes.POINTER(_int),
                                    ctypes.POINTER(_int), ctypes.POINTER(_int)]),
    'qgs_prebuild': (_int, [_int, _i64, _vp, _vp, _i64, _vp, _vp, _int, ctypes.POINTER(_int), ctypes.c_char_p]),
    'qgs_prebuild_rank': (_int, [_int, _int, _i64, _vp, _vp, _i64, _vp, _vp, _int, ctypes.POINTER(_int), ctypes.c_char_p]),
    'qgs_prebuild_qr': (_int, [_int, _int, ctypes.c_char_p]),
    'qgs_qr_kernel_source': (_i64, [_int, _int, ctypes.c_char_p, _i64]),
    'qgs_model_kernel_source': (_i64, [_vp, ctypes.c_char_p, _i64]),
}

_LIB = None


class QgsHipError(RuntimeError):
    pass


def build_library(force=False):
    """Compile qgs_amd/libqgs_hip.so with hipcc for gfx950 (works without a GPU)."""
    args = ['make', '-C', os.path.join(_HERE, 'csrc'), '-s']
    if force:
        args.append('-B')
    subprocess.check_call(args)
    return LIB_PATH


def lib():
    """Load libqgs_hip.so.  Raises QgsHipError if it has not been built -- there is no fallback."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise QgsHipError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C qgs_amd/csrc`.  qgs_amd has no CPU fallback." % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhiprtc.so.7 with the
        # same sonames as /opt/rocm's.  Whichever is mapped first serves both; torch does not find the GPU when
        # the system runtime was mapped first, so torch (the device-memory / stream plumbing of this stack) is
        # imported before libqgs_hip.so whenever it is installed.
        # (QGS_HIP_NO_TORCH_PRELOAD=1 skips this: the GPU-less kernel pre-build then uses the system ROCm's hiprtc,
        # which generates the better stepper code -- 282 instead of 324 registers.)
        if os.environ.get('QGS_HIP_NO_TORCH_PRELOAD') != '1':
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def last_error():
    return lib().qgs_last_error().decode(errors='replace')


def _check(rc):
    if rc != 0:
        raise QgsHipError(last_error())


def backend_info():
    n = _int(0)
    buf = ctypes.create_string_buffer(256)
    _check(lib().qgs_backend_info(ctypes.byref(n), buf, 256))
    return n.value, buf.value.decode()


def fp64_fma_rate(device=0, target_ms=30.0):
    """(TFLOP/s, milliseconds): what the device sustains on independent fp64 FMAs alone (qgs_fp64_fma_rate)."""
    tf, ms = ctypes.c_double(0.), ctypes.c_double(0.)
    _check(lib().qgs_fp64_fma_rate(int(device), float(target_ms), ctypes.byref(tf), ctypes.byref(ms)))
    return tf.value, ms.value


def n_records(time, write_steps):
    time = np.ascontiguousarray(time, dtype=np.float64)
    return int(lib().qgs_n_records(time, len(time), int(write_steps)))


def _c(x, dt=np.float64):
    return np.ascontiguousarray(x, dtype=dt)


def _ptr(a):
    return a.ctypes.data_as(_vp) if a is not None else None


def _advise_huge_pages(arr):
    """Ask for transparent huge pages behind a large fresh array (madvise MADV_HUGEPAGE on its whole 2 MiB blocks): the first
    touch of a 2 MiB page costs one fault instead of 512, which is what a result of tens of GB spends its first fill on."""
    try:
        addr, n = arr.ctypes.data, arr.nbytes
        lo = -(-addr // (2 << 20)) * (2 << 20)
        hi = (addr + n) // (2 << 20) * (2 << 20)
        if hi > lo:
            libc = ctypes.CDLL(None, use_errno=True)
            libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
            libc.madvise(ctypes.c_void_p(lo), ctypes.c_size_t(hi - lo), 14)            # MADV_HUGEPAGE
    except Exception:
        pass


class _Store(object):
    """One block of host memory of the result pool.

    A block the pool can keep (`pinned=True`) is page-locked while it lives: device-to-host copies into it run at the pinned PCIe
    rate, the kernels can store into it, and the page faults of a fresh allocation are taken once, not inside a copy.  It is
    allocated by the runtime (qgs_host_alloc = hipHostMalloc, portable + mapped), not carved out of the C library's heap and
    registered: registered heap memory is where every GPU write fault of round 4 happened.
    A block too large for the pool to keep is used once; page-locking it (0.04 s per GB: 8 s for 189 GB, 3 s of the 6 s the
    72 GB estimator record of profiles/r04_lyap_big.json took) buys nothing.  It is ordinary NumPy memory, huge pages
    requested, and is filled through the page-locked bounce ring of csrc/host_bridge.cpp while the next window is computed."""

    #: blocks above this size are never page-locked (QGS_HOST_PIN_MAX_BYTES, default 256 GiB)
    PIN_MAX_BYTES = int(os.environ.get('QGS_HOST_PIN_MAX_BYTES', str(256 << 30)))

    def __init__(self, n_doubles, pinned=True, huge=True):
        self.size = n_doubles
        self.nbytes = 8 * n_doubles
        self._ptr = None
        if pinned and self.nbytes <= self.PIN_MAX_BYTES:
            p = _vp()
            try:
                if lib().qgs_host_alloc(self.nbytes, ctypes.byref(p)) == 0 and p.value:
                    self._ptr = p.value
            except Exception:
                self._ptr = None
        if self._ptr is not None:
            self.array = np.frombuffer((ctypes.c_char * self.nbytes).from_address(self._ptr), dtype=np.float64)
        else:
            self.array = np.empty(n_doubles)
            # huge pages for a block that is filled in strided runs (a window of records at a time: every window reaches every
            # page); a block filled front to back by the host threads is better off without them -- a 2 MiB fault under two
            # threads' feet serialises them (72 GB estimator record: 1.8 - 2.1 s without, 2.0 - 3.1 s with, profiles/r05_lyap_big.md)
            if huge and self.nbytes >= (64 << 20):
                _advise_huge_pages(self.array)
        self._pinned = self._ptr is not None

    def __del__(self):
        try:
            if self._ptr is not None:
                self.array = None
                lib().qgs_host_free(_vp(self._ptr))
                self._ptr = None
        except Exception:
            pass


class _PooledBlock(object):
    """Owner of one result block: the ndarray handed to the caller has this object as its base, so the block returns
    to the pool when the caller's last reference (array or view) goes away."""

    def __init__(self, pool, store, shape):
        self._pool, self._store = pool, store
        self.__array_interface__ = {'shape': tuple(int(q) for q in shape), 'typestr': '<f8', 'version': 3,
                                    'data': (store.array.ctypes.data, False)}

    def __del__(self):
        try:
            self._pool._give_back(self._store)
        except Exception:
            pass


class _ResultPool(object):
    """Host memory of large results (ensemble trajectories, propagators), recycled and page-locked.

    A fresh 1.9 GB NumPy array costs ~110 ms of first-touch page faults inside the device-to-host copy (measured: 152 ms
    for the first copy into it, 41 ms for the following ones at 50 GB/s; 35.5 ms = 57 GB/s once the block is page-locked,
    which takes 83 ms once).  Results are still *fresh arrays owned by the caller* -- a block is only reused after every
    array and view on it has been garbage collected.  QGS_HOST_POOL_BYTES caps what the pool keeps (default 4 GiB, 0
    disables it)."""

    MIN_BYTES = 8 << 20

    def __init__(self):
        import threading
        self._free = {}
        self._held = 0
        self._cap = int(os.environ.get('QGS_HOST_POOL_BYTES', str(4 << 30)))
        self._lock = threading.Lock()          # the shards of a device group may ask for blocks from their own threads

    def empty(self, shape, pinned=True, huge=True):
        """`pinned=False`: a new block is not page-locked (a caller whose result is far larger than what the pool keeps, and reaches
        the host through the bounce ring anyway: page-locking its small side blocks costs more than it saves, 0.12 s per 1.9 GB)."""
        n = int(np.prod(shape))
        if n * 8 < self.MIN_BYTES or self._cap <= 0:
            return np.empty(shape)
        size = -(-n * 8 // (2 << 20)) * (2 << 20) // 8          # whole 2 MiB blocks, in doubles
        store = None
        with self._lock:
            lst = self._free.get(size)
            if lst:
                store = lst.pop()
                self._held -= store.nbytes
        if store is None:
            store = _Store(size, pinned=pinned and size * 8 <= self._cap, huge=huge)      # (a block the pool could never keep is not page-locked)
        return np.asarray(_PooledBlock(self, store, shape))

    @property
    def cap(self):
        return self._cap

    def _give_back(self, store):
        with self._lock:
            if store._pinned and self._held + store.nbytes <= self._cap:          # (only page-locked blocks are worth keeping)
                self._free.setdefault(store.size, []).append(store)
                self._held += store.nbytes


_RESULTS = _ResultPool()


def to_host(d_tensor):
    """A device tensor (torch) as a NumPy array.  The copy is made by the library (qgs_memcpy_d2h), which never shows the
    runtime a pageable pointer: small results arrive through its bounce blocks in ordinary NumPy memory, large ones (>= 512 MB)
    in a block of the result pool -- page-locked, one DMA copy at the PCIe rate (1.9 GB: 35 ms instead of ~0.3 s) -- or, beyond
    what the pool keeps, in huge-page NumPy memory through the bounce ring."""
    import torch
    if d_tensor.device.type != 'cuda':
        return d_tensor.detach().cpu().numpy()
    d_tensor = d_tensor.contiguous()
    nbytes = d_tensor.numel() * d_tensor.element_size()
    if d_tensor.dtype == torch.float64 and nbytes >= (512 << 20):
        out = _RESULTS.empty(tuple(d_tensor.shape))
    else:
        out = np.empty(tuple(d_tensor.shape), dtype=torch.empty(0, dtype=d_tensor.dtype).numpy().dtype)
    if nbytes:
        st = torch.cuda.current_stream(d_tensor.device).cuda_stream
        _check(lib().qgs_memcpy_d2h(_cuda_index(d_tensor.device), out.ctypes.data, d_tensor.data_ptr(), nbytes, st or None))
    return out


def _cuda_index(device):
    """The GPU a torch tensor actually lives on: an index-less `torch.device('cuda')` means the CURRENT device, not GPU 0."""
    import torch
    return torch.cuda.current_device() if device.index is None else device.index


def to_device(array, device):
    """A NumPy array as a new tensor on `device` (torch).  On a GPU the copy is made by the library (qgs_memcpy_h2d: through its
    page-locked bounce blocks, one pageable transfer at a time per device), not by handing the runtime the caller's pages."""
    import torch
    a = np.ascontiguousarray(array)
    device = torch.device(device)
    if device.type != 'cuda':
        return torch.from_numpy(a).to(device)
    t = torch.empty(a.shape, dtype=torch.from_numpy(np.empty(0, dtype=a.dtype)).dtype, device=device)
    if a.nbytes:
        st = torch.cuda.current_stream(device).cuda_stream
        _check(lib().qgs_memcpy_h2d(_cuda_index(t.device), t.data_ptr(), a.ctypes.data, a.nbytes, st or None))
    return t


def _tensor_rank(coo, jcoo=None):
    """3 for the (nnz, 3) coordinate lists of QgsTensor, 5 for the (nnz, 5) ones of QgsTensorDynamicT / QgsTensorT4."""
    rank = int(coo.shape[1]) if coo.ndim == 2 else 0
    if rank not in (3, 5) or (jcoo is not None and jcoo.ndim == 2 and jcoo.shape[0] and jcoo.shape[1] != rank):
        raise ValueError("tensor coordinates must be (nnz, 3) or (nnz, 5) and of the same rank for the Jacobian tensor")
    return rank


def prebuild(ndim, coo, val, jcoo, jval, stage_counts=(4,), arch=None):
    """Compile + cache the specialised kernels of a tensor without a GPU (see qgs_prebuild)."""
    coo, val = _c(coo, np.int32), _c(val)
    jcoo = _c(jcoo, np.int32) if jcoo is not None else None
    jval = _c(jval) if jval is not None else None
    sc = (_int * len(stage_counts))(*stage_counts)
    _check(lib().qgs_prebuild_rank(int(ndim), _tensor_rank(coo, jcoo), len(val), _ptr(coo), _ptr(val),
                                   0 if jval is None else len(jval), _ptr(jcoo), _ptr(jval), len(stage_counts), sc,
                                   arch.encode() if arch else None))


def prebuild_qr(n_rows, n_cols, arch=None):
    """Compile + cache the batched-QR kernel of one matrix shape without a GPU (see qgs_prebuild_qr)."""
    _check(lib().qgs_prebuild_qr(int(n_rows), int(n_cols), arch.encode() if arch else None))


def qr_kernel_source(n_rows, n_cols):
    """Generated source of that kernel, first line `// plan <signature>` (qgs_qr_kernel_source); no GPU needed."""
    n = lib().qgs_qr_kernel_source(int(n_rows), int(n_cols), None, 0)
    if n < 0:
        _check(-1)
    buf = ctypes.create_string_buffer(int(n) + 1)
    lib().qgs_qr_kernel_source(int(n_rows), int(n_cols), buf, int(n) + 1)
    return buf.value.decode()


class HipModel(object):
    """A model's tensors staged on one GPU (qgs_model handle).

    coo/val, jcoo/jval are the arrays the reference's closures capture
    (qgs/functions/tendencies.py:92-96): `tensor.coords.T`, `tensor.data`, and the same for
    `jacobian_tensor`.  (nnz, 3) coordinates: the rank-3 tensor contracted by sparse_mul3 / sparse_mul2;
    (nnz, 5): the rank-5 tensor of the dynamic-T / T4 models contracted by sparse_mul5 / sparse_mul4
    (tendencies.py:98-109).
    """

    KERNEL_AUTO, KERNEL_GENERIC, KERNEL_SPECIALISED = 0, 1, 2

    def __init__(self, ndim, coo, val, jcoo=None, jval=None, device=0):
        self.ndim = int(ndim)
        self.device = int(device)
        self.coo, self.val = _c(coo, np.int32), _c(val)
        self.jcoo = _c(jcoo, np.int32) if jcoo is not None else None
        self.jval = _c(jval) if jval is not None else None
        self.rank = _tensor_rank(self.coo, self.jcoo)
        h = _vp()
        _check(lib().qgs_model_create_rank(self.device, self.ndim, self.rank, len(self.val), _ptr(self.coo), _ptr(self.val),
                                           0 if self.jval is None else len(self.jval), _ptr(self.jcoo), _ptr(self.jval),
                                           ctypes.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, '_h', None):
            lib().qgs_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_kernel(self, kind):
        _check(lib().qgs_model_set_kernel(self._h, int(kind)))

    @property
    def specialised_available(self):
        return bool(lib().qgs_model_info(self._h, 4))

    @property
    def last_windows(self):
        """Record windows the last host-layout integration was cut into (1: the record fitted the device budget)."""
        return int(lib().qgs_model_info(self._h, 8))

    @property
    def last_groups(self):
        """Member groups the last `rk_integrate` into pageable memory ran in (1: all members in one pass)."""
        return int(lib().qgs_model_info(self._h, 9))

    @property
    def n_derived(self):
        """Derived monomials of the generated tendencies / Jacobian code (rank-5 tensors; (0, 0) for rank 3)."""
        return int(lib().qgs_model_info(self._h, 6)), int(lib().qgs_model_info(self._h, 7))

    @property
    def n_reduced_terms(self):
        """Terms of the tendencies / Jacobian polynomial in the bilinear form the generated code evaluates (qgs_model_info 10, 11)."""
        return int(lib().qgs_model_info(self._h, 10)), int(lib().qgs_model_info(self._h, 11))

    def last_kernel_info(self):
        name = ctypes.create_string_buffer(128)
        v, s, l, sc = _int(0), _int(0), _int(0), _int(0)
        _check(lib().qgs_last_kernel_info(self._h, name, 128, ctypes.byref(v), ctypes.byref(s), ctypes.byref(l),
                                          ctypes.byref(sc)))
        return dict(name=name.value.decode(), vgprs=v.value, sgprs=s.value, lds_bytes=l.value, scratch_bytes=sc.value)

    def kernel_clock(self):
        """(shader GHz, milliseconds) of the last generated kernel's clock probe (qgs_kernel_clock), or None when the last
        kernel carries none (the generic kernels)."""
        ghz, ms = ctypes.c_double(0.), ctypes.c_double(0.)
        if lib().qgs_kernel_clock(self._h, ctypes.byref(ghz), ctypes.byref(ms)) != 0:
            return None
        return ghz.value, ms.value

    def kernel_source(self):
        n = lib().qgs_model_kernel_source(self._h, None, 0)
        buf = ctypes.create_string_buffer(int(n) + 1)
        lib().qgs_model_kernel_source(self._h, buf, int(n) + 1)
        return buf.value.decode()

    def _check_ic(self, ic, tg_ic=None):
        """The C side copies n_traj * ndim doubles from the pointer it is given: refuse shapes that do not hold them."""
        if ic.ndim != 2 or ic.shape[1] != self.ndim or ic.shape[0] < 1:
            raise ValueError('initial conditions must have shape (n_traj, %d), got %r' % (self.ndim, ic.shape))
        if tg_ic is not None and (tg_ic.ndim != 3 or tg_ic.shape[:2] != (ic.shape[0], self.ndim) or tg_ic.shape[2] < 1):
            raise ValueError('tangent initial conditions must have shape (%d, %d, n_tg), got %r'
                             % (ic.shape[0], self.ndim, tg_ic.shape))

    # ---- host-layout calls (NumPy in, NumPy out) ----------------------------------------------
    def tendencies(self, x):
        x = _c(x)
        xb = x.reshape(-1, self.ndim)
        out = np.empty_like(xb)
        _check(lib().qgs_tendencies(self._h, xb.shape[0], xb, out))
        return out.reshape(x.shape)

    def jacobian(self, x):
        x = _c(x)
        xb = x.reshape(-1, self.ndim)
        out = _RESULTS.empty((xb.shape[0], self.ndim, self.ndim))
        _check(lib().qgs_jacobian(self._h, xb.shape[0], xb, out))
        return out[0] if x.ndim == 1 else out

    def rk_integrate(self, time, ic, time_direction, write_steps, b, c, a, out=None):
        """`out`: the caller's own (n_traj, ndim, n_records) C-contiguous float64 array instead of a block of the result pool."""
        time, ic, b, c, a = _c(time), _c(ic), _c(b), _c(c), _c(a)
        self._check_ic(ic)
        nrec = n_records(time, write_steps)
        if out is not None:
            if out.dtype != np.float64 or not out.flags.c_contiguous or out.shape != (ic.shape[0], self.ndim, nrec):
                raise ValueError('out must be a C-contiguous float64 array of shape (n_traj, ndim, n_records)')
            traj = out
        else:
            # (a record that will leave in member groups -- rk_member_groups of qgs_hip_api.hip: a large ensemble, beyond one device
            # window -- is filled front to back: no huge pages, see _Store)
            grouped = (ic.shape[0] >= 2048 and 3 * 8 * ic.shape[0] * self.ndim * nrec > (8 << 30)
                       and 'QGS_HIP_RECORD_WINDOW_MB' not in os.environ)
            traj = _RESULTS.empty((ic.shape[0], self.ndim, nrec), True, not grouped)
        _check(lib().qgs_rk_integrate(self._h, ic.shape[0], ic, time, len(time), int(time_direction), int(write_steps),
                                      len(b), b, c, a, traj))
        return traj

    def rk_integrate_moments(self, time, ic, time_direction, write_steps, b, c, a, variance=True, final_states=False):
        """Same run as `rk_integrate`; returns the ensemble mean and variance (n_dim, n_records) of every variable at every
        record (and optionally the final states) -- the trajectories never leave the device."""
        time, ic, b, c, a = _c(time), _c(ic), _c(b), _c(c), _c(a)
        self._check_ic(ic)
        nrec = n_records(time, write_steps)
        mean = np.empty((self.ndim, nrec))
        var = np.empty((self.ndim, nrec)) if variance else None
        fin = np.empty((ic.shape[0], self.ndim)) if final_states else None
        _check(lib().qgs_rk_integrate_moments(self._h, ic.shape[0], ic, time, len(time), int(time_direction), int(write_steps),
                                              len(b), b, c, a, mean, var.ctypes.data if variance else None,
                                              fin.ctypes.data if final_states else None))
        return mean, var, fin

    def ensemble_moments_device(self, n_traj, ld, n_rows, d_x, d_mean, d_var=0, stream=0):
        _check(lib().qgs_ensemble_moments_device(self._h, n_traj, ld, int(n_rows), d_x, d_mean, d_var or None, stream or None))

    def rk_tgls_integrate(self, time, ic, tg_ic, time_direction, write_steps, b, c, a, adjoint, inverse, out=None):
        """`out`: the caller's own (traj, fmatrix) pair of C-contiguous float64 arrays instead of blocks of the result pool."""
        time, ic, tg_ic, b, c, a = _c(time), _c(ic), _c(tg_ic), _c(b), _c(c), _c(a)
        self._check_ic(ic, tg_ic)
        nrec = n_records(time, write_steps)
        n_traj, n_tg = ic.shape[0], tg_ic.shape[2]
        if out is not None:
            traj, fm = out
            for arr, shape in ((traj, (n_traj, self.ndim, nrec)), (fm, (n_traj, self.ndim, n_tg, nrec))):
                if arr.dtype != np.float64 or not arr.flags.c_contiguous or arr.shape != shape:
                    raise ValueError('out must be C-contiguous float64 arrays of shapes (n_traj, ndim, n_records) and (n_traj, ndim, n_tg, n_records)')
        else:
            traj = _RESULTS.empty((n_traj, self.ndim, nrec))
            fm = _RESULTS.empty((n_traj, self.ndim, n_tg, nrec))
        _check(lib().qgs_rk_tgls_integrate(self._h, n_traj, n_tg, ic, tg_ic, time, len(time), int(time_direction),
                                           int(write_steps), len(b), b, c, a, int(bool(adjoint)), float(inverse),
                                           traj, fm))
        return traj, fm

    # ---- device-layout calls (raw device pointers as ints; enqueue on `stream`, no sync) ---------
    def pack_states(self, n_traj, ld, d_rows, d_modes, stream=0):
        _check(lib().qgs_pack_states(self._h, n_traj, ld, d_rows, d_modes, stream or None))

    def unpack_states(self, n_traj, ld, d_modes, d_rows, stream=0):
        _check(lib().qgs_unpack_states(self._h, n_traj, ld, d_modes, d_rows, stream or None))

    def pack_tangent(self, n_traj, ld, n_tg, d_rows, d_modes, stream=0):
        _check(lib().qgs_pack_tangent(self._h, n_traj, ld, int(n_tg), d_rows, d_modes, stream or None))

    def local_exponents_device(self, n, d_rdiag, dt, d_out, stream=0):
        """d_out[i] = log|d_rdiag[i]| / dt, i < n (qgs_local_exponents_device)."""
        _check(lib().qgs_local_exponents_device(self._h, int(n), d_rdiag, float(dt), d_out, stream or None))

    def unpack_records(self, n_traj, ld, n_inner, nrec, d_in, d_out, stream=0):
        _check(lib().qgs_unpack_records(self._h, n_traj, ld, n_inner, nrec, d_in, d_out, stream or None))

    def unpack_window(self, n_traj, ld, n_inner, n_window, nrec, first_record, d_window, dst, stream=0):
        """Records [first_record, first_record + n_window) of a (n_traj, n_inner, nrec) block `dst` (device pointer, or host
        pointer -- page-locked or pageable) from a mode-major window of records on the device."""
        _check(lib().qgs_unpack_window(self._h, n_traj, ld, int(n_inner), int(n_window), int(nrec), int(first_record), d_window, dst,
                                       stream or None))

    def unpack_window_enqueue(self, n_traj, ld, n_inner, n_window, nrec, first_record, d_window, dst, stream=0):
        """`unpack_window` that does not wait for a pageable `dst` to be filled: the window is handed to the device's drain thread
        (page-locked bounce blocks + host threads); `drain_wait()` blocks until every such window has arrived."""
        _check(lib().qgs_unpack_window_enqueue(self._h, n_traj, ld, int(n_inner), int(n_window), int(nrec), int(first_record), d_window,
                                               dst, stream or None))

    def drain_wait(self):
        _check(lib().qgs_drain_wait(self._h))

    def tendencies_device(self, n_traj, ld, d_x, d_dx, stream=0):
        _check(lib().qgs_tendencies_device(self._h, n_traj, ld, d_x, d_dx, stream or None))

    def rk_integrate_device(self, n_traj, ld, d_ic, time, time_direction, write_steps, b, c, a, d_rec, stream=0):
        time, b, c, a = _c(time), _c(b), _c(c), _c(a)
        _check(lib().qgs_rk_integrate_device(self._h, n_traj, ld, d_ic, time, len(time), int(time_direction),
                                             int(write_steps), len(b), b, c, a, d_rec, stream or None))

    def rk_integrate_rows_device(self, n_traj, d_ic_rows, time, time_direction, write_steps, b, c, a, d_traj_rows):
        """`rk_integrate` with both blocks in device memory in the reference's layouts ((n_traj, ndim) in, (n_traj, ndim,
        n_records) out); only a window of mode-major records is held as scratch.  Blocking."""
        time, b, c, a = _c(time), _c(b), _c(c), _c(a)
        _check(lib().qgs_rk_integrate_rows_device(self._h, n_traj, d_ic_rows, time, len(time), int(time_direction),
                                                  int(write_steps), len(b), b, c, a, d_traj_rows))

    def batched_qr_device(self, n_traj, ld, n_rows, n_cols, d_a, d_rdiag, stream=0):
        _check(lib().qgs_batched_qr_device(self._h, n_traj, ld, int(n_rows), int(n_cols), d_a, d_rdiag, stream or None))

    def batched_matmul_device(self, n_traj, ld, n_rows, n_inner, n_cols, d_a, d_b, d_c, trans_a=False, triangular=0, stream=0):
        """C[row][col][member] = A B (A^T B with trans_a) per member; triangular 1: upper triangle of C only, 2: B upper triangular."""
        _check(lib().qgs_batched_matmul_device(self._h, n_traj, ld, int(n_rows), int(n_inner), int(n_cols), int(bool(trans_a)),
                                               int(triangular), d_a, d_b, d_c, stream or None))

    def clv_backstep_device(self, n_traj, ld, n_vec, d_r, d_a_in, d_a_out, d_norm, d_noise=None, noise_pert=0.0, stream=0):
        """a_out = unit-norm columns of R^-1 a_in (+ noise * noise_pert on the diagonal); d_norm[col][member] = the norms."""
        _check(lib().qgs_clv_backstep_device(self._h, n_traj, ld, int(n_vec), d_r, d_a_in, d_a_out, d_norm, d_noise or None,
                                             float(noise_pert), stream or None))

    def rk_tgls_integrate_device(self, n_traj, ld, n_tg, d_ic, d_tg_ic, time, time_direction, write_steps, b, c, a,
                                 adjoint, inverse, d_rec, d_rec_fm, stream=0):
        time, b, c, a = _c(time), _c(b), _c(c), _c(a)
        _check(lib().qgs_rk_tgls_integrate_device(self._h, n_traj, ld, n_tg, d_ic, d_tg_ic, time, len(time),
                                                  int(time_direction), int(write_steps), len(b), b, c, a,
                                                  int(bool(adjoint)), float(inverse), d_rec, d_rec_fm, stream or None))


class Contraction(object):
    """A COO tensor staged on one GPU for the general contraction with explicit vectors (qgs_contraction handle): what the
    reference's sparse_mul3 / sparse_mul5 (`matrix=False`) and sparse_mul2 / sparse_mul4 (`matrix=True`) compute for ANY
    arguments (qgs/functions/sparse_mul.py:13-158)."""

    def __init__(self, n_slots, coo, val, matrix=False, device=0):
        self.n_slots, self.matrix = int(n_slots), bool(matrix)
        coo, val = _c(coo, np.int32), _c(val)
        rank = _tensor_rank(coo)
        self.n_fac = rank - (2 if matrix else 1)
        h = _vp()
        _check(lib().qgs_contraction_create(int(device), self.n_slots, rank, 2 if matrix else 1, len(val), _ptr(coo), _ptr(val),
                                            ctypes.byref(h)))
        self._h = h

    def apply(self, *vectors):
        if len(vectors) != self.n_fac:
            raise ValueError('%d vectors expected' % self.n_fac)
        vecs = np.ascontiguousarray(np.stack([np.asarray(v, dtype=np.float64) for v in vectors]))
        if vecs.shape != (self.n_fac, self.n_slots):
            raise ValueError('the vectors must have length %d' % self.n_slots)
        res = np.empty((self.n_slots, self.n_slots) if self.matrix else (self.n_slots,))
        _check(lib().qgs_contraction_apply(self._h, vecs, res))
        return res

    def close(self):
        if getattr(self, '_h', None):
            lib().qgs_contraction_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def visible_devices():
    """Indices of the GPUs this process can use (HIP's numbering, after ROCR_/HIP_VISIBLE_DEVICES)."""
    return list(range(backend_info()[0]))


def pool_moments(parts, variance=True):
    """Mean and population variance of a set from the (count, mean, variance) of its disjoint parts (Chan et al.'s pairwise
    update; arrays of any common shape)."""
    n_acc, mean, m2 = 0, None, None
    for n, mu, var in parts:
        if mean is None:
            n_acc, mean, m2 = n, np.array(mu, dtype=np.float64), (np.asarray(var) * n if variance else None)
            continue
        delta = mu - mean
        tot = n_acc + n
        if variance:
            m2 = m2 + np.asarray(var) * n + delta * delta * (n_acc * n / tot)
        mean = mean + delta * (n / tot)
        n_acc = tot
    return mean, (m2 / n_acc if variance else None)


class _ShardModel(HipModel):
    """A group's model of one shard: borrowed handle, destroyed with the group."""

    def __init__(self, handle, owner):                       # noqa: super().__init__ creates a handle; this one is borrowed
        self._h = handle
        self._owner = owner
        self.ndim = int(lib().qgs_model_info(handle, 0))
        self.device = int(lib().qgs_model_info(handle, 3))
        self.rank = int(lib().qgs_model_info(handle, 5))

    def close(self):
        self._h = None


class HipModelGroup(object):
    """A model's tensors staged on several GPUs (qgs_group handle): the host-layout calls of `HipModel`, with the members
    split into contiguous shards, one per listed device, each integrated and delivered by its own GPU.

    What the reference does with the cores of the machine (one task per trajectory over `num_threads` worker processes,
    qgs/integrators/integrator.py:79-82, 133-142, 386-395).  `devices` may list a device more than once.
    """

    KERNEL_AUTO, KERNEL_GENERIC, KERNEL_SPECIALISED = 0, 1, 2

    def __init__(self, ndim, coo, val, jcoo=None, jval=None, devices=None):
        self.ndim = int(ndim)
        self.devices = [int(d) for d in (visible_devices() if devices is None else devices)]
        if not self.devices:
            raise ValueError('a device group needs at least one device')
        self.coo, self.val = _c(coo, np.int32), _c(val)
        self.jcoo = _c(jcoo, np.int32) if jcoo is not None else None
        self.jval = _c(jval) if jval is not None else None
        self.rank = _tensor_rank(self.coo, self.jcoo)
        h = _vp()
        devs = (_int * len(self.devices))(*self.devices)
        _check(lib().qgs_group_create(len(self.devices), devs, self.ndim, self.rank, len(self.val), _ptr(self.coo), _ptr(self.val),
                                      0 if self.jval is None else len(self.jval), _ptr(self.jcoo), _ptr(self.jval),
                                      ctypes.byref(h)))
        self._h = h
        self.models = [_ShardModel(lib().qgs_group_model(h, i), self) for i in range(len(self.devices))]

    def close(self):
        if getattr(self, '_h', None):
            for m in self.models:
                m.close()
            lib().qgs_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return len(self.devices)

    def shard(self, n_traj, i):
        """(start, count) of shard i of an ensemble of n_traj members."""
        a, n = _i64(0), _i64(0)
        _check(lib().qgs_group_shard(self._h, int(n_traj), int(i), ctypes.byref(a), ctypes.byref(n)))
        return a.value, n.value

    def set_kernel(self, kind):
        _check(lib().qgs_group_set_kernel(self._h, int(kind)))

    @property
    def specialised_available(self):
        return self.models[0].specialised_available

    def last_kernel_info(self):
        return self.models[0].last_kernel_info()

    _check_ic = HipModel._check_ic

    def tendencies(self, x):
        x = _c(x)
        xb = x.reshape(-1, self.ndim)
        out = np.empty_like(xb)
        _check(lib().qgs_group_tendencies(self._h, xb.shape[0], xb, out))
        return out.reshape(x.shape)

    def jacobian(self, x):
        x = _c(x)
        xb = x.reshape(-1, self.ndim)
        out = _RESULTS.empty((xb.shape[0], self.ndim, self.ndim))
        _check(lib().qgs_group_jacobian(self._h, xb.shape[0], xb, out))
        return out[0] if x.ndim == 1 else out

    def rk_integrate(self, time, ic, time_direction, write_steps, b, c, a):
        time, ic, b, c, a = _c(time), _c(ic), _c(b), _c(c), _c(a)
        self._check_ic(ic)
        nrec = n_records(time, write_steps)
        traj = _RESULTS.empty((ic.shape[0], self.ndim, nrec))
        _check(lib().qgs_group_rk_integrate(self._h, ic.shape[0], ic, time, len(time), int(time_direction), int(write_steps),
                                            len(b), b, c, a, traj))
        return traj

    def rk_tgls_integrate(self, time, ic, tg_ic, time_direction, write_steps, b, c, a, adjoint, inverse):
        time, ic, tg_ic, b, c, a = _c(time), _c(ic), _c(tg_ic), _c(b), _c(c), _c(a)
        self._check_ic(ic, tg_ic)
        nrec = n_records(time, write_steps)
        n_traj, n_tg = ic.shape[0], tg_ic.shape[2]
        traj = _RESULTS.empty((n_traj, self.ndim, nrec))
        fm = _RESULTS.empty((n_traj, self.ndim, n_tg, nrec))
        _check(lib().qgs_group_rk_tgls_integrate(self._h, n_traj, n_tg, ic, tg_ic, time, len(time), int(time_direction),
                                                 int(write_steps), len(b), b, c, a, int(bool(adjoint)), float(inverse),
                                                 traj, fm))
        return traj, fm

    def rk_integrate_moments(self, time, ic, time_direction, write_steps, b, c, a, variance=True, final_states=False):
        """Ensemble mean / variance of every variable at every record: each shard reduces on its own GPU, the per-shard
        moments are pooled on the host (Chan et al.'s pairwise update)."""
        ic = _c(ic)
        self._check_ic(ic)
        n_total = ic.shape[0]
        import threading
        parts = [None] * len(self.models)

        errors = []

        def run(i):
            try:
                a0, n = self.shard(n_total, i)
                if n > 0:
                    parts[i] = (n,) + self.models[i].rk_integrate_moments(time, ic[a0:a0 + n], time_direction, write_steps, b, c, a,
                                                                           variance=variance, final_states=final_states)
            except Exception as e:                               # re-raised on the calling thread below
                errors.append(e)
        threads = [threading.Thread(target=run, args=(i,)) for i in range(len(self.models))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        parts = [q for q in parts if q is not None]
        mean, var = pool_moments([(n, mu, v) for n, mu, v, _ in parts], variance)
        fin = np.concatenate([q[3] for q in parts], axis=0) if final_states else None
        return mean, var, fin
