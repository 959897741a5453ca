"""CPU: the host-side logic of the multi-GPU routes that needs no GPU -- which devices an integrator takes by default, how the
per-shard moments are pooled, how members are dealt to shards."""
import numpy as np


def test_default_device_is_every_gpu_for_large_ensembles(monkeypatch):
    from qgs_amd.integrators import integrate as fn
    from qgs_amd import _lib
    assert fn.resolve_device(None, 1000) is None and fn.resolve_device(0, 10 ** 7) == 0
    monkeypatch.setattr(_lib, 'visible_devices', lambda: [0, 1, 2, 3, 4, 5, 6, 7])
    assert fn.resolve_device(None, 2 * 65536) == 'all' and fn.resolve_device(None, 2 * 65536 - 1) is None
    assert fn.resolve_device([0, 1], 10 ** 7) == [0, 1]
    monkeypatch.setattr(_lib, 'visible_devices', lambda: [0])
    assert fn.resolve_device(None, 10 ** 7) is None



def test_pooled_moments_equal_the_moments_of_the_whole():
    from qgs_amd._lib import pool_moments
    rng = np.random.RandomState(0)
    x = rng.randn(1000, 5, 7) * 3.0 + 10.0
    cuts = [0, 1, 338, 339, 900, 1000]                       # shards of 1, 337, 1, 561 and 100 members
    parts = [(b - a, x[a:b].mean(axis=0), x[a:b].var(axis=0)) for a, b in zip(cuts[:-1], cuts[1:])]
    mean, var = pool_moments(parts)
    assert np.abs(mean - x.mean(axis=0)).max() < 1e-13 and np.abs(var - x.var(axis=0)).max() < 1e-12
    mean, var = pool_moments(parts, variance=False)
    assert var is None and np.abs(mean - x.mean(axis=0)).max() < 1e-13


def test_shard_bounds_match_the_group_rule():
    """parallel.shard_bounds (ranks) and qgs_group_shard (devices of one process) deal members the same way: contiguous blocks,
    remainder to the first shards.  (qgs_group_shard itself needs a group, i.e. a GPU: tests/test_gpu_windows_group.py.)"""
    from qgs_amd.parallel import shard_bounds
    assert shard_bounds(1048576, 8) == [(r * 131072, (r + 1) * 131072) for r in range(8)]
    assert shard_bounds(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_bounds(3, 5) == [(0, 1), (1, 2), (2, 3), (3, 3), (3, 3)]
