"""CPU: the C-ABI library builds, loads and exports every symbol include/qgs_hip.h declares, and the
ctypes table in qgs_amd/_lib.py names exactly those symbols.  No compute call is made (no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO


def _header_symbols():
    src = open(os.path.join(REPO, 'include', 'qgs_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(qgs_[a-z0-9_]+)\s*\(', src)))


def test_header_declares_something():
    syms = _header_symbols()
    assert 'qgs_rk_integrate' in syms and 'qgs_tendencies' in syms and len(syms) >= 15


def test_library_exports_every_header_symbol():
    from qgs_amd import _lib
    _lib.build_library()
    L = ctypes.CDLL(_lib.LIB_PATH)
    for s in _header_symbols():
        assert hasattr(L, s), 'libqgs_hip.so does not export ' + s


def test_ctypes_table_matches_header():
    from qgs_amd import _lib
    assert sorted(_lib.SIGNATURES) == _header_symbols()


def test_n_records_host_helper():
    from qgs_amd import _lib
    t = np.concatenate((np.arange(0., 1., 0.1), [1.]))
    assert _lib.n_records(t, 0) == 1
    assert _lib.n_records(t, 1) == 11
    assert _lib.n_records(t, 3) == 5
    assert _lib.n_records(t, 5) == 3
    assert _lib.n_records(t[:1], 1) == 1


def test_fails_loudly_without_gpu():
    """On a GPU-less host every numerical entry point must raise, never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from qgs_amd import _lib
    with pytest.raises(_lib.QgsHipError):
        _lib.backend_info()
    with pytest.raises(_lib.QgsHipError):
        _lib.HipModel(2, np.array([[1, 0, 1]], dtype=np.int32), np.array([1.0]))


def test_prebuild_generates_code_objects(tmp_path, monkeypatch):
    """hiprtc cross-compiles the tensor-specialised kernels for gfx950 without a GPU."""
    from qgs_amd import _lib
    monkeypatch.setenv('QGS_HIP_CACHE_DIR', str(tmp_path))
    coo = np.array([[1, 0, 1], [1, 1, 2], [2, 0, 0], [2, 1, 1]], dtype=np.int32)
    val = np.array([-0.5, 2.0, 0.25, -1.0])
    jcoo = np.array([[1, 1, 0], [1, 1, 2], [1, 2, 1], [2, 1, 1]], dtype=np.int32)
    jval = np.array([-0.5, 2.0, 2.0, -2.0])
    _lib.prebuild(2, coo, val, jcoo, jval, stage_counts=(2,))
    objs = [f for f in os.listdir(tmp_path) if f.endswith('.hsaco')]
    # one code object per kernel: f, Df, rk_s2, rkr_s2 (write_steps = 1), rkstages_s2, tgl_s2, rkstagesp_s2 + tglp_s2 (stage record in
    # mode pairs), tglx4_s2 (2 rows: no split)
    assert len(objs) == 9
    assert all(os.path.getsize(os.path.join(tmp_path, f)) > 1000 for f in objs)


def test_prebuild_shards_cover_every_code_object(tmp_path):
    """build() pre-builds in parallel: with QGS_HIP_PREBUILD_SHARD=i/n a process compiles every n-th code object of the
    list; the shards are disjoint and together produce all of them."""
    import subprocess
    import sys
    code = ("import sys, numpy as np\n"
            "sys.path.insert(0, %r)\n"
            "from qgs_amd import _lib\n"
            "coo = np.array([[1, 0, 1], [1, 1, 2], [2, 0, 0], [2, 1, 1]], dtype=np.int32)\n"
            "val = np.array([-0.5, 2.0, 0.25, -1.0])\n"
            "jcoo = np.array([[1, 1, 0], [1, 1, 2], [1, 2, 1], [2, 1, 1]], dtype=np.int32)\n"
            "jval = np.array([-0.5, 2.0, 2.0, -2.0])\n"
            "_lib.prebuild(2, coo, val, jcoo, jval, stage_counts=(2,))\n" % REPO)
    seen = []
    for i in range(2):
        d = tmp_path / ('shard%d' % i)
        d.mkdir()
        env = dict(os.environ, QGS_HIP_CACHE_DIR=str(d), QGS_HIP_PREBUILD_SHARD='%d/2' % i, QGS_HIP_NO_TORCH_PRELOAD='1')
        subprocess.check_call([sys.executable, '-c', code], env=env)
        seen.append(sorted(f for f in os.listdir(d) if f.endswith('.hsaco')))
    assert len(seen[0]) + len(seen[1]) == 9 and abs(len(seen[0]) - len(seen[1])) == 1
    assert not set(seen[0]) & set(seen[1])


def test_package_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under qgs_amd/ may import, load or reference it."""
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, 'qgs_amd')):
        for fn in files:
            if fn.endswith(('.py', '.cpp', '.hip', '.h')):
                txt = open(os.path.join(root, fn), errors='replace').read()
                if re.search(r'(^|\W)(import\s+oracle|from\s+oracle|libqgs_oracle|oracle/)', txt):
                    bad.append(os.path.join(root, fn))
    assert not bad, bad


def test_result_pool_hands_out_fresh_arrays_and_recycles_released_ones():
    """Large results come from recycled host blocks (no first-touch page faults in the device-to-host copy), but a block
    is reused only after the caller dropped every array and view on it: results stay caller-owned, fresh arrays."""
    import gc
    import numpy as np
    from qgs_amd import _lib
    pool = _lib._ResultPool()
    small = pool.empty((4, 36, 3))
    assert small.base is None                                   # below the threshold: a plain NumPy array
    a = pool.empty((2048, 36, 101))
    a[:] = 1.
    assert isinstance(a, np.ndarray) and a.flags['C_CONTIGUOUS'] and a.flags['WRITEABLE'] and a.shape == (2048, 36, 101)
    addr = a.ctypes.data
    b = pool.empty((2048, 36, 101))
    assert b.ctypes.data != addr                                # a is still held by the caller
    view = np.squeeze(a[:1])
    del a
    gc.collect()
    assert pool.empty((2048, 36, 101)).ctypes.data != addr      # a view keeps the block out of the pool
    del view
    gc.collect()
    c = pool.empty((2048, 36, 101))
    assert c.ctypes.data == addr                                # released: recycled
    assert _lib._f64p.from_param(c) is not None                 # accepted by the ctypes signatures
