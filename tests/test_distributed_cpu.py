"""CPU, world_size 2 and 8, gloo: the multi-GPU path (member sharding + final gather) with the per-rank engine
replaced by an oracle-backed stand-in (the HIP engine needs a GPU; sharding / gathering does not)."""
import os
import socket

import numpy as np
import pytest

from conftest import RK4, load_golden


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


class _OracleIntegrator(object):
    """RungeKuttaIntegrator interface on top of the CPU oracle (test seam of integrate_ensemble)."""

    def __init__(self, b=None, c=None, a=None):
        self.b, self.c, self.a = (RK4['b'], RK4['c'], RK4['a']) if b is None else (b, c, a)
        self._recorded_traj = None

    def set_func(self, f):
        from oracle.oracle import OracleModel
        self._m = OracleModel(f.ndim, f.coo, f.val)

    def integrate(self, t0, t, dt, ic=None, forward=True, write_steps=1):
        from qgs_amd.integrators.integrate import time_grid
        self._time, self._ws, self._fw = time_grid(t0, t, dt), write_steps, forward
        self._recorded_traj = self._m.integrate_runge_kutta_jit(self._time, ic, 1 if forward else -1, write_steps,
                                                                self.b, self.c, self.a)

    def get_trajectories(self):
        from qgs_amd.integrators.integrate import record_times
        return record_times(self._time, self._ws, self._fw), np.squeeze(self._recorded_traj)

    def terminate(self):
        pass


def _worker(rank, world, port, n_traj, write_steps, out_dir, gather_bytes=None):
    import torch.distributed as dist
    if gather_bytes:                                     # forces the record-chunked gather (parallel.gather_to_host)
        os.environ['QGS_GATHER_BYTES'] = str(gather_bytes)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from qgs_amd.functions.tendencies import tendencies_from_tensor
    from qgs_amd.parallel import integrate_ensemble, ShardedEnsemble
    g = load_golden('a36')
    f, _ = tendencies_from_tensor(g.ndim, g['coo'], g['val'])
    ic = np.random.RandomState(0).rand(n_traj, g.ndim) * 0.01
    ens = ShardedEnsemble(n_traj)
    assert ens.rank == rank and sum(ens.counts) == n_traj
    time, traj = integrate_ensemble(f, 0., 1., 0.1, ic, write_steps=write_steps, integrator_factory=_OracleIntegrator)
    np.savez(os.path.join(out_dir, 'r%d.npz' % rank), time=np.asarray(time), traj=traj)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_traj,write_steps,gather_bytes', [(8, 0, None), (7, 3, None), (1, 1, None), (7, 1, 5000), (8, 2, 3000)])
def test_sharded_ensemble_gloo_world2(tmp_path, n_traj, write_steps, gather_bytes):
    import torch.multiprocessing as mp
    from oracle.oracle import OracleModel
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_traj, write_steps, str(tmp_path), gather_bytes), nprocs=2, join=True)
    g = load_golden('a36')
    ic = np.random.RandomState(0).rand(n_traj, g.ndim) * 0.01
    from qgs_amd.integrators.integrate import time_grid
    ref = OracleModel(g.ndim, g['coo'], g['val']).integrate_runge_kutta_jit(time_grid(0., 1., 0.1), ic, 1, write_steps,
                                                                            RK4['b'], RK4['c'], RK4['a'])
    for r in range(2):
        z = np.load(os.path.join(str(tmp_path), 'r%d.npz' % r))
        assert z['traj'].shape == ref.shape
        assert np.array_equal(z['traj'], ref)              # every rank holds the full, ordered ensemble


@pytest.mark.parametrize('n_traj,write_steps,gather_bytes', [(19, 2, None), (5, 1, None), (21, 1, 9000)])
def test_sharded_ensemble_gloo_world8(tmp_path, n_traj, write_steps, gather_bytes):
    """The rank count of BASELINE configs[4] (8 ranks, here on gloo with the oracle as the per-rank engine): ragged shards (19
    members: 3, 3, 3, 2, ...), EMPTY shards (5 members over 8 ranks: ranks 5-7 integrate nothing and still take part in the
    gather), and the record-chunked gather (`gather_to_host` under a byte budget smaller than one rank's block)."""
    import torch.multiprocessing as mp
    from oracle.oracle import OracleModel
    from qgs_amd.parallel import shard_bounds
    world = 8
    counts = [b - a for a, b in shard_bounds(n_traj, world)]
    assert sum(counts) == n_traj and (n_traj >= world or counts.count(0) == world - n_traj)
    mp.spawn(_worker, args=(world, _free_port(), n_traj, write_steps, str(tmp_path), gather_bytes), nprocs=world, join=True)
    g = load_golden('a36')
    ic = np.random.RandomState(0).rand(n_traj, g.ndim) * 0.01
    from qgs_amd.integrators.integrate import time_grid
    ref = OracleModel(g.ndim, g['coo'], g['val']).integrate_runge_kutta_jit(time_grid(0., 1., 0.1), ic, 1, write_steps,
                                                                            RK4['b'], RK4['c'], RK4['a'])
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), 'r%d.npz' % r))
        assert z['traj'].shape == ref.shape and np.array_equal(z['traj'], ref)      # every rank: the full, ordered ensemble


def _root_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from qgs_amd.parallel import ShardedEnsemble, RootGather
    ens = ShardedEnsemble(world * 5)
    root = RootGather(ens, dst=0)
    bufs = [torch.zeros((5, 3), dtype=torch.float64) for _ in range(2)]
    pending = [None, None]
    last = None
    for k in range(4):                                     # double-buffered asynchronous gathers, as bench.py does
        if pending[k % 2] is not None:
            pending[k % 2].wait()
        bufs[k % 2].fill_(100.0 * k + rank)
        pending[k % 2], parts = root.start(bufs[k % 2], async_op=True)
        last = parts
    for w in pending:
        if w is not None:
            w.wait()
    if rank == 0:
        np.save(os.path.join(out_dir, 'root.npy'), torch.cat(last).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_root_gather_async_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_root_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), 'root.npy'))
    assert got.shape == (10, 3)
    assert np.all(got[:5] == 300.0) and np.all(got[5:] == 301.0)


def test_shard_bounds():
    from qgs_amd.parallel import shard_bounds
    assert shard_bounds(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_bounds(1048576, 8) == [(i * 131072, (i + 1) * 131072) for i in range(8)]
    assert shard_bounds(1, 2) == [(0, 1), (1, 1)]
