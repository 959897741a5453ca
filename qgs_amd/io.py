"""Trajectory output and a tensor cache ("on-disk tensor cache; trajectory output format").

The reference's scripts end with ``np.savetxt(filename, traj)`` on rows ``[time, x_1 .. x_n]`` (qgs_rp.py:114-131,
qgs_maooam.py:123-140): fine for one trajectory, hopeless for an ensemble record (65 536 members x 36 variables x 101
records = 1.9 GB of doubles would become ~5 GB of text written at a few MB/s).  `save_trajectories` writes the arrays of
`get_trajectories()` as two ``.npy`` files that can be memory-mapped back; `save_trajectory_txt` keeps the scripts' text
layout for a single trajectory.

`cached_tendencies(params, cache_dir)` is `create_tendencies` with the tensors kept on disk: the inner products and the
tensor assembly (5 s for MAOOAM 6x6 here, 94 s in the reference) run once per parameter set.
"""
import hashlib
import os

import numpy as np


def save_trajectories(filename, time, traj):
    """Write ``(time, traj)`` as returned by `get_trajectories()` to ``<filename>.time.npy`` / ``<filename>.traj.npy``.

    `traj` may have any of the shapes the integrators return ((n_dim,), (n_dim, n_records), (n_traj, n_dim, n_records), ...).
    The big array is streamed through a memory map in blocks along its first axis, so no second copy of it is made."""
    time = np.asarray(time, dtype=np.float64)
    traj = np.asarray(traj)
    np.save(filename + '.time.npy', time)
    out = np.lib.format.open_memmap(filename + '.traj.npy', mode='w+', dtype=traj.dtype, shape=traj.shape)
    if traj.ndim == 0 or traj.nbytes <= (64 << 20):
        out[...] = traj
    else:
        rows = max(1, (64 << 20) // max(1, traj[0].nbytes))
        for a in range(0, traj.shape[0], rows):
            out[a:a + rows] = traj[a:a + rows]
    out.flush()
    del out
    return filename + '.time.npy', filename + '.traj.npy'


def load_trajectories(filename, mmap_mode='r'):
    """``(time, traj)`` written by `save_trajectories`; `traj` is a read-only memory map unless ``mmap_mode=None``."""
    return np.load(filename + '.time.npy'), np.load(filename + '.traj.npy', mmap_mode=mmap_mode)


def save_trajectory_txt(filename, time, traj, **savetxt_kwargs):
    """One trajectory in the text layout of the reference's scripts (qgs_rp.py:114-131): a row per record, the time in the
    first column, then the variables.  `traj` is (n_dim, n_records) (or (n_dim,) for a single record)."""
    time = np.atleast_1d(np.asarray(time, dtype=np.float64))
    traj = np.asarray(traj, dtype=np.float64)
    if traj.ndim == 1:
        traj = traj[:, np.newaxis]
    if traj.ndim != 2 or traj.shape[1] != time.shape[0]:
        raise ValueError('expected one trajectory (n_dim, n_records) with %d records, got %r' % (time.shape[0], traj.shape))
    np.savetxt(filename, np.column_stack((time, traj.T)), **savetxt_kwargs)


def _assembly_version():
    """Hash of the source files that turn a parameter set into the cached arrays: a fix to the inner products, the tensor
    assembly, the derived parameters or the Jacobian operands must not keep serving tensors cached by the old code.  Every
    module of the package takes part except the GPU glue (bindings, integrators, toolbox), which never touches the arrays."""
    root = os.path.dirname(os.path.abspath(__file__))
    h = hashlib.sha256(b'qgs_amd tensor cache v3')
    for sub in ('basis', 'functions', 'inner_products', 'params', 'tensors'):
        d = os.path.join(root, sub)
        for fn in sorted(os.listdir(d)):
            if fn.endswith('.py'):
                h.update(fn.encode())
                with open(os.path.join(d, fn), 'rb') as f:
                    h.update(f.read())
    return h.hexdigest()[:12]


def _canonical(obj, depth=0):
    """Parameter objects as nested plain data with sorted keys (pickle bytes are not canonical: memoisation and attribute
    order let equal parameter sets hash differently)."""
    if depth > 12:
        return repr(obj)
    if isinstance(obj, (bool, int, str, bytes, type(None))):
        return obj
    if isinstance(obj, float):
        return float(obj).hex()
    if isinstance(obj, np.ndarray):
        return ('ndarray', obj.dtype.str, obj.shape, hashlib.sha256(np.ascontiguousarray(obj).tobytes()).hexdigest())
    if isinstance(obj, np.generic):
        return _canonical(obj.item(), depth + 1)
    if isinstance(obj, (list, tuple)):
        return [_canonical(q, depth + 1) for q in obj]
    if isinstance(obj, dict):
        return sorted((repr(k), _canonical(v, depth + 1)) for k, v in obj.items())
    state = getattr(obj, '__dict__', None)
    if state is not None:
        return (type(obj).__name__, _canonical(state, depth + 1))
    return repr(obj)


def params_key(params):
    """Hash of a parameter set (canonical form of the `QgParams` object: every physical parameter and the mode selection) and
    of the tensor-assembly code: a change to either gives another key."""
    text = repr((_assembly_version(), int(params.ndim), _canonical(params)))
    return hashlib.sha256(text.encode()).hexdigest()[:24]


def cached_tendencies(params, cache_dir, device=0):
    """``[f, Df]`` of `create_tendencies(params)`, with the tensor operands read from / written to
    ``<cache_dir>/qgs_tensor_<key>.npz`` (key = `params_key(params)`); `f` and `Df` are bound to GPU `device` whether the
    tensors came from the cache or were just assembled."""
    from qgs_amd.functions.tendencies import create_tendencies, tendencies_from_tensor
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, 'qgs_tensor_%s.npz' % params_key(params))
    if os.path.exists(path):
        z = np.load(path)
        f, Df = tendencies_from_tensor(int(z['ndim']), z['coo'], z['val'], z['jcoo'], z['jval'], device=device)
        return [f, Df]
    f0, Df0 = create_tendencies(params)
    tmp = path + '.tmp%d.npz' % os.getpid()
    np.savez_compressed(tmp, ndim=np.int64(f0.ndim), coo=f0.coo, val=f0.val, jcoo=Df0.coo, jval=Df0.val)
    os.replace(tmp, path)
    f, Df = tendencies_from_tensor(f0.ndim, f0.coo, f0.val, Df0.coo, Df0.val, device=device)
    return [f, Df]
