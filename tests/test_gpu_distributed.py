"""GPU: the multi-GPU plumbing on real hardware at the world size the box offers -- RCCL (backend "nccl") process group,
gathers on device tensors, the device-resident `integrate_ensemble`, and `bench.py` as the driver types it."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


_RANK_SCRIPT = r'''
import json, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(repo)r)
sys.path.insert(0, os.path.join(%(repo)r, 'tests'))
rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
torch.cuda.set_device(local)
dev = torch.device('cuda', local)
dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
from qgs_amd import _lib
from qgs_amd.functions.tendencies import tendencies_from_tensor
from qgs_amd.parallel import integrate_ensemble, ShardedEnsemble, RootGather
g = np.load(os.path.join(%(repo)r, 'tests', 'golden', 'a36.npz'))
ndim = int(g['ndim'])
f, Df = tendencies_from_tensor(ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
out = {}
# 1. gathers on device tensors through RCCL
n_total = 6 * world
ens = ShardedEnsemble(n_total)
assert ens.distributed and ens.world_size == world and ens.rank == rank
local_block = torch.arange(ens.n_local * 3, dtype=torch.float64, device=dev).reshape(ens.n_local, 3) + 1000 * rank
full = ens.gather(local_block)
assert full.is_cuda and full.shape == (n_total, 3)
expect = torch.cat([torch.arange(c * 3, dtype=torch.float64).reshape(c, 3) + 1000 * r for r, c in enumerate(ens.counts)])
out['all_gather_ok'] = bool(torch.equal(full.cpu(), expect))
root = RootGather(ens, dst=0)
work, parts = root.start(local_block, async_op=True)
if work is not None:
    work.wait()
torch.cuda.synchronize()
if rank == 0:
    out['root_gather_ok'] = bool(torch.equal(torch.cat(parts).cpu(), expect))
# 2. the device-resident ensemble integration: every rank on ITS device, one collective, one D2H
ic = np.random.RandomState(5).rand(130 * world + 3, ndim) * 0.01
time, traj = integrate_ensemble(f, 0., 1., 0.1, ic, write_steps=3)
model = f.hip_model(device=local)
out['model_device'] = int(_lib.lib().qgs_model_info(model._h, 3))
out['local_rank'] = local
c = np.array([0., .5, .5, 1.]); b = np.array([1 / 6, 1 / 3, 1 / 3, 1 / 6]); a = np.zeros((4, 4)); a[1, 0] = .5; a[2, 1] = .5; a[3, 2] = 1.
from qgs_amd.integrators.integrate import time_grid
ref = model.rk_integrate(time_grid(0., 1., 0.1), ic, 1, 3, b, c, a)
out['traj_shape'] = list(traj.shape)
out['max_abs_diff_vs_single_call'] = float(np.abs(traj - ref).max())
out['time'] = [float(x) for x in np.atleast_1d(time)]
# backward, no records, ragged shards
time_b, traj_b = integrate_ensemble(f, 0., 1., 0.1, ic[:world * 2 + 1], forward=False, write_steps=0)
ref_b = model.rk_integrate(time_grid(0., 1., 0.1), ic[:world * 2 + 1], -1, 0, b, c, a)
out['backward_diff'] = float(np.abs(traj_b - ref_b).max())
if rank == 0:
    print('RESULT ' + json.dumps(out), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def _run_ranks(world):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, '-c', _RANK_SCRIPT % {'repo': REPO}], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se.decode()[-4000:]
    line = [ln for ln in outs[0][0].decode().splitlines() if ln.startswith('RESULT ')]
    assert line, outs[0][0].decode()
    return json.loads(line[0][7:])


def _device_count():
    import torch
    return int(torch.cuda.device_count())


def test_rccl_gathers_and_device_resident_ensemble():
    """`init_process_group('nccl')` on the device(s) of this box (world size = min(visible GPUs, 2)): RootGather /
    ShardedEnsemble.gather on device tensors, integrate_ensemble through the device route, model on the rank's GPU."""
    world = min(2, max(1, _device_count()))
    r = _run_ranks(world)
    assert r['all_gather_ok'] and r['root_gather_ok']
    assert r['model_device'] == r['local_rank']
    assert r['traj_shape'] == [130 * world + 3, 36, 5]
    assert r['max_abs_diff_vs_single_call'] == 0.0          # same kernel, same member -> same lanes' arithmetic
    assert r['backward_diff'] == 0.0
    assert r['time'] == pytest.approx([0., .3, .6, .9, 1.])


def _bench(*args, timeout=900):
    p = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py')] + list(args), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout, cwd=REPO)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def test_bench_with_a_process_group_at_world_size_one():
    rc, out, err = _bench('--gpus', '1', '--force-dist', '--steps', '4', '--warmup', '1', '--no-cpu-baseline', '--no-extra-configs',
                          '--no-cold-start')
    assert rc == 0, err[-3000:]
    lines = [ln for ln in out.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r['n_gpus'] == 1 and r['steps'] == 4 and r['value'] > 1e9
    assert r['gather_ms'] is not None and r['gather_ms'] > 0.0
    # the entries of an N > 1 line: the gather-free per-GPU reference (the denominator of the scaling efficiency) and the total
    ref = r['single_gpu_reference']
    assert ref['members_per_gpu'] == 65536 and ref['passes'] == 4 and ref['value'] > 1e9
    assert 0.5 < ref['value_over_n_times_reference'] < 1.5 and r['config']['members_total'] == 65536
    # and the timed kernel's own output was checked against the oracle
    assert r['parity_check']['ok'] and r['parity_check']['kernel'] == r['roofline']['kernel'] and 'parity_failures' not in r
    assert r['roofline']['bound'] == 'fp64_valu' and 0.3 < r['roofline']['frac'] <= 1.0
    assert r['roofline']['kernel'].startswith('qgs_spec_rk')


def test_bench_typed_with_more_gpus_than_visible_says_so():
    n = _device_count()
    rc, out, err = _bench('--gpus', str(n + 7), '--steps', '1', '--warmup', '0')
    assert rc != 0 and rc != 2
    assert '%d GPUs requested, %d visible' % (n + 7, n) in err


@pytest.mark.skipif('_device_count() < 2')
def test_bench_self_launches_two_ranks():
    rc, out, err = _bench('--gpus', '2', '--steps', '2', '--warmup', '1')
    assert rc == 0, err[-3000:]
    r = json.loads([ln for ln in out.splitlines() if ln.startswith('{')][0])
    assert r['n_gpus'] == 2 and r['gather_ms'] > 0.0
