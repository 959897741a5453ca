"""CPU, world size 8, gloo: bench.py's rank body -- the control flow every rank of `bench.py --gpus 8` runs between its
barriers (warm-up, gather-free reference passes, timed passes with the double-buffered asynchronous gather onto rank 0,
drain, max-over-ranks timing, the separately timed gathers) -- with the HIP engine replaced by a host stand-in.

No 8-GPU box has run this path yet; ordering or deadlock mistakes in the overlap scheme cannot show at world size 1 (RCCL's
gather degenerates) and the world-size-2 tests cover `RootGather`, not the bench's own loop."""
import json
import os
import socket
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


class _Work(object):
    """A gather in flight on buffer q: the engine learns when the rank body waits for it."""

    def __init__(self, engine, q, work):
        self.engine, self.q, self.work = engine, q, work

    def wait(self):
        if self.work is not None:
            self.work.wait()
        self.engine.in_flight[self.q] = False


class HostEngine(object):
    """The engine interface of bench.rank_body on host tensors: `compute` stamps buffer q with (rank, pass number)."""

    def __init__(self, torch, dist, rank, world, n=6, ndim=3):
        sys.path.insert(0, REPO)
        from qgs_amd.parallel import ShardedEnsemble, RootGather
        self.torch, self.dist, self.rank, self.world, self.n = torch, dist, rank, world, n
        self.out = [torch.zeros((n, ndim), dtype=torch.float64) for _ in range(2)]
        ens = ShardedEnsemble(world * n)
        self.root = RootGather(ens, dst=0)
        self.all = torch.full((world * n, ndim), -1.0, dtype=torch.float64) if rank == 0 else None
        self.passes = 0
        self.in_flight = [False, False]
        self.log = []

    def compute(self, q, record):
        assert not self.in_flight[q], 'buffer %d overwritten while its gather is in flight' % q
        self.passes += 1
        time.sleep(0.0005 * ((self.rank * 7 + self.passes * 3) % 5))       # ranks drift apart, as GPUs do
        self.out[q].fill_(1000.0 * self.rank + self.passes)
        self.log.append((q, self.passes, bool(record)))

    def start_gather(self, q):
        self.in_flight[q] = True
        work, _ = self.root.start(self.out[q], out=self.all, async_op=True)
        return _Work(self, q, work)

    def timed_gather(self, q):
        t0 = time.perf_counter()
        self.start_gather(q).wait()
        return (time.perf_counter() - t0) * 1e3

    def synchronize(self):
        pass

    def max_over_ranks(self, x):
        t = self.torch.tensor([x], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def _worker(rank, world, port, steps, warmup, ref_passes, out_dir):
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, REPO)
    import bench
    eng = HostEngine(torch, dist, rank, world)
    elapsed, ref, gather_ms = bench.rank_body(eng, dist, True, steps, warmup, reference_passes=ref_passes)
    res = {'elapsed': elapsed, 'ref': ref, 'gather_ms': gather_ms, 'passes': eng.passes, 'log': eng.log,
           'in_flight': eng.in_flight}
    if rank == 0:
        res['all'] = eng.all.numpy().tolist()
    with open(os.path.join(out_dir, 'r%d.json' % rank), 'w') as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


def test_rank_body_world8_gloo(tmp_path):
    import torch.multiprocessing as mp
    world, steps, warmup, ref_passes = 8, 7, 3, 2
    mp.spawn(_worker, args=(world, _free_port(), steps, warmup, ref_passes, str(tmp_path)), nprocs=world, join=True)
    res = [json.load(open(os.path.join(str(tmp_path), 'r%d.json' % r))) for r in range(world)]
    # every rank did warm-up + reference + timed passes, alternating the two buffers from 0 in each phase, events only in the timed one
    for r in res:
        assert r['passes'] == warmup + ref_passes + steps
        want = [(k % 2, False) for k in range(warmup)] + [(k % 2, False) for k in range(ref_passes)] + [(k % 2, True) for k in range(steps)]
        assert [(q, rec) for q, _, rec in r['log']] == want
        assert r['in_flight'] == [False, False]                       # nothing left in flight after the body
    # the timings are the maximum over ranks: identical everywhere, positive
    assert len({r['elapsed'] for r in res}) == 1 and res[0]['elapsed'] > 0
    assert len({r['ref'] for r in res}) == 1 and res[0]['ref'] > 0
    assert all(r['gather_ms'] is not None and r['gather_ms'] >= 0 for r in res)
    # rank 0 holds, in member order, what every rank last computed into buffer 0 (the separately timed gathers send buffer 0;
    # its last writer is the last even-numbered timed pass)
    last_even_timed = warmup + ref_passes + (steps if (steps - 1) % 2 == 0 else steps - 1)
    got = np.array(res[0]['all'])
    n = got.shape[0] // world
    for r in range(world):
        assert np.all(got[r * n:(r + 1) * n] == 1000.0 * r + last_even_timed), (r, got[r * n])


def test_rank_body_without_a_process_group():
    """World size 1 without --force-dist: no gather, no collective, no reference passes."""
    sys.path.insert(0, REPO)
    import bench

    class Solo(object):
        def __init__(self):
            self.passes, self.gathers = 0, 0

        def compute(self, q, record):
            self.passes += 1

        def start_gather(self, q):
            self.gathers += 1

        def synchronize(self):
            pass
    eng = Solo()
    elapsed, ref, gather_ms = bench.rank_body(eng, None, False, 4, 2)
    assert eng.passes == 6 and eng.gathers == 0 and ref is None and gather_ms is None and elapsed > 0
