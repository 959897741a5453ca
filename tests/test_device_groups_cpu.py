"""CPU: the host-side logic of the multi-GPU routes that needs no GPU -- which devices an integrator takes by default, how the
per-shard moments are pooled, how members are dealt to shards."""
import numpy as np


def test_default_device_is_every_gpu_for_large_ensembles_only_on_request(monkeypatch):
    """Spreading a large ensemble over every visible GPU without being asked to is opt-in (QGS_HIP_AUTO_ALL_DEVICES=1): the
    single-process device group has never run on two physical devices."""
    from qgs_amd.integrators import integrate as fn
    from qgs_amd import _lib
    assert fn.resolve_device(None, 1000) is None and fn.resolve_device(0, 10 ** 7) == 0
    monkeypatch.setattr(_lib, 'visible_devices', lambda: [0, 1, 2, 3, 4, 5, 6, 7])
    for var in ('LOCAL_RANK', 'RANK', 'WORLD_SIZE'):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.delenv('QGS_HIP_AUTO_ALL_DEVICES', raising=False)
    assert fn.resolve_device(None, 10 ** 7) is None and fn.resolve_device('all', 10 ** 7) == 'all'      # the default: one device
    monkeypatch.setenv('QGS_HIP_AUTO_ALL_DEVICES', '1')
    assert fn.resolve_device(None, 2 * 65536) == 'all' and fn.resolve_device(None, 2 * 65536 - 1) is None
    assert fn.resolve_device([0, 1], 10 ** 7) == [0, 1]
    monkeypatch.setattr(_lib, 'visible_devices', lambda: [0])
    assert fn.resolve_device(None, 10 ** 7) is None


def test_a_rank_of_a_multi_process_job_never_spreads_by_itself(monkeypatch):
    """One rank per GPU (torchrun, bench.py --gpus N, parallel.integrate_ensemble): the other GPUs belong to the other ranks,
    so `device=None` keeps meaning the tendencies' own device whatever the ensemble size; 'all' must be asked for."""
    from qgs_amd.integrators import integrate as fn
    from qgs_amd import _lib
    monkeypatch.setattr(_lib, 'visible_devices', lambda: [0, 1, 2, 3, 4, 5, 6, 7])
    for var in ('LOCAL_RANK', 'RANK', 'WORLD_SIZE'):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv('QGS_HIP_AUTO_ALL_DEVICES', '1')
    assert fn.resolve_device(None, 10 ** 7) == 'all'
    for var, val in (('LOCAL_RANK', '3'), ('RANK', '0'), ('WORLD_SIZE', '8')):
        monkeypatch.setenv(var, val)
        assert fn.in_multi_process_job() and fn.resolve_device(None, 10 ** 7) is None
        assert fn.resolve_device('all', 10 ** 7) == 'all'
        monkeypatch.delenv(var)
    monkeypatch.setenv('WORLD_SIZE', '1')
    assert not fn.in_multi_process_job() and fn.resolve_device(None, 10 ** 7) == 'all'



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


def test_record_window_plan_covers_every_record_once():
    """The window plan of the host-layout integrations (qgs_record_window: host arithmetic, no GPU): for any run length,
    cadence, direction and window size the windows tile the directed records and the steps without gaps or overlaps, every
    record written at the top of a step lies in the window that runs that step, the final record in the last window only, and
    the stored index ranges tile the record axis (mirrored for backward runs)."""
    import ctypes
    from qgs_amd import _lib
    L = _lib.lib()
    rng = np.random.RandomState(3)
    cases = [(0, 0, 1), (0, 1, 1), (9, 3, 3), (10, 3, 4), (100, 1, 13), (100, 0, 5), (7, 10, 1), (50, 7, 2)]
    cases += [(int(rng.randint(0, 300)), int(rng.randint(0, 12)), int(rng.randint(1, 40))) for _ in range(200)]
    for n_steps, ws, W in cases:
        time = np.arange(n_steps + 1, dtype=np.float64)
        n_rec = _lib.n_records(time, ws)
        n_top = 0 if ws == 0 else -(-n_steps // ws)                      # records written at the top of a step: steps 0, ws, 2 ws, ... < n_steps
        assert n_rec == n_top + 1
        for backward in (0, 1):
            out = (ctypes.c_int64 * 6)()
            n_win = L.qgs_record_window(n_rec, n_steps, ws, backward, W, 0, out)
            assert n_win == -(-n_rec // min(W, n_rec))
            next_lo, next_step, stored = 0, 0, []
            for k in range(n_win):
                assert L.qgs_record_window(n_rec, n_steps, ws, backward, W, k, out) == n_win
                lo, hi, sb, se, wf, lo_s = (int(q) for q in out)
                assert lo == next_lo and lo < hi <= n_rec and hi - lo <= W
                assert sb == next_step and sb <= se <= n_steps
                assert wf == (1 if k == n_win - 1 else 0) and (not wf or (hi == n_rec and se == n_steps))
                tops = [st // ws for st in range(sb, se) if ws > 0 and st % ws == 0]
                assert all(lo <= iw < hi for iw in tops), (n_steps, ws, W, k, tops, lo, hi)
                assert sorted(tops + ([n_rec - 1] if wf else [])) == list(range(lo, hi)), (n_steps, ws, W, k)
                assert lo_s == (n_rec - hi if backward else lo)
                stored += list(range(lo_s, lo_s + hi - lo))
                next_lo, next_step = hi, se
            assert next_lo == n_rec and next_step == n_steps and sorted(stored) == list(range(n_rec))
    assert L.qgs_record_window(0, 1, 1, 0, 1, 0, out) < 0 and _lib.last_error()


def test_host_memory_guard_reads_the_container_limits(tmp_path):
    """What a result may take of the host is MemAvailable cut down to the process' control-group limits (v2 `memory.max`, v1
    `memory.limit_in_bytes`, on the group itself or any ancestor): /proc/meminfo describes the machine, not the container."""
    from qgs_amd.toolbox import lyapunov
    proc = tmp_path / 'cgroup'
    root = tmp_path / 'sys'
    # cgroup v2, limit on the parent group
    (root / 'a' / 'b').mkdir(parents=True)
    proc.write_text('0::/a/b\n')
    (root / 'memory.max').write_text('max\n')
    (root / 'a' / 'memory.max').write_text('%d\n' % (64 << 30))
    (root / 'a' / 'memory.current').write_text('%d\n' % (4 << 30))
    (root / 'a' / 'b' / 'memory.max').write_text('max\n')
    assert lyapunov._cgroup_memory_room(str(proc), str(root)) == 60 << 30
    # the tighter of two limits on the way
    (root / 'a' / 'b' / 'memory.max').write_text('%d\n' % (16 << 30))
    (root / 'a' / 'b' / 'memory.current').write_text('%d\n' % (1 << 30))
    assert lyapunov._cgroup_memory_room(str(proc), str(root)) == 15 << 30
    # usage counts the group's page cache, which the kernel gives back: 10 GB "used" of which 9 GB are file pages leave 15 GB of 16
    (root / 'a' / 'b' / 'memory.current').write_text('%d\n' % (10 << 30))
    (root / 'a' / 'b' / 'memory.stat').write_text('anon %d\ninactive_file %d\nactive_file %d\nshmem 0\n' % (1 << 30, 6 << 30, 3 << 30))
    assert lyapunov._cgroup_memory_room(str(proc), str(root)) == 15 << 30
    (root / 'a' / 'b' / 'memory.stat').unlink()
    (root / 'a' / 'b' / 'memory.current').write_text('%d\n' % (1 << 30))
    # cgroup v1 memory controller; 2^63-ish = no limit
    proc.write_text('4:memory:/jobs/x\n0::/\n')
    (root / 'memory' / 'jobs' / 'x').mkdir(parents=True)
    (root / 'memory' / 'memory.limit_in_bytes').write_text('9223372036854771712\n')
    (root / 'memory' / 'jobs' / 'x' / 'memory.limit_in_bytes').write_text('%d\n' % (32 << 30))
    (root / 'memory' / 'jobs' / 'x' / 'memory.usage_in_bytes').write_text('%d\n' % (2 << 30))
    (root / 'a' / 'memory.max').write_text('max\n')
    (root / 'a' / 'b' / 'memory.max').write_text('max\n')
    assert lyapunov._cgroup_memory_room(str(proc), str(root)) == 30 << 30
    (root / 'memory' / 'jobs' / 'x' / 'memory.usage_in_bytes').write_text('%d\n' % (12 << 30))
    (root / 'memory' / 'jobs' / 'x' / 'memory.stat').write_text('cache 0\ninactive_file 1\ntotal_inactive_file %d\ntotal_active_file %d\n' % (7 << 30, 3 << 30))
    assert lyapunov._cgroup_memory_room(str(proc), str(root)) == 30 << 30
    # no limit anywhere
    (root / 'memory' / 'jobs' / 'x' / 'memory.limit_in_bytes').write_text('9223372036854771712\n')
    assert lyapunov._cgroup_memory_room(str(proc), str(root)) is None
    assert lyapunov._host_memory_available() > 0


def test_clv_host_helpers_match_the_plain_loops():
    """The batched column normalisation and triangular solve of the covariant estimator against one-matrix-at-a-time loops
    (what qgs/functions/util.py:55-98 does per trajectory)."""
    from qgs_amd.functions.util import add_to_dict, normalize_matrix_columns as _normalize_columns, solve_triangular_matrix as _solve_triangular
    assert add_to_dict(add_to_dict({}, 'k', 2.0), 'k', 1.5) == {'k': 3.5}
    rng = np.random.RandomState(5)
    r = np.triu(rng.randn(4, 7, 7)) + 3.0 * np.eye(7)
    b = np.triu(rng.randn(4, 7, 7))
    x = _solve_triangular(r, b)
    an, norm = _normalize_columns(b + np.eye(7))
    for n in range(4):
        want = np.zeros((7, 7))
        for i in range(2, 8):
            want[:i, i - 1] = np.linalg.solve(r[n, :i, :i], b[n, :i, i - 1])
        want[0, 0] = b[n, 0, 0] / r[n, 0, 0]
        assert np.array_equal(x[n], want)
        assert np.abs(r[n] @ x[n] - b[n]).max() < 1e-13
        for i in range(7):
            col = (b[n] + np.eye(7))[:, i]
            assert abs(norm[n, i] - np.linalg.norm(col, 2)) < 1e-15 and np.abs(an[n][:, i] - col / np.linalg.norm(col, 2)).max() < 1e-15


def test_subspace_intersection_matches_the_plain_loops_on_any_number_of_threads():
    """Method 1 of the covariant estimator on the host: batched over members and records, one thread or several, against the
    loop the reference runs per trajectory, record and vector (qgs/toolbox/lyapunov.py:1313-1317)."""
    from qgs_amd.toolbox.lyapunov import _intersect_subspaces
    rng = np.random.RandomState(12)
    nt, nd, nr = 5, 7, 3
    bvec, fvec = np.empty((nt, nd, nd, nr)), np.empty((nt, nd, nd, nr))
    for i in range(nt):
        for r in range(nr):
            bvec[i, :, :, r] = np.linalg.qr(rng.randn(nd, nd))[0]
            fvec[i, :, :, r] = np.linalg.qr(rng.randn(nd, nd))[0]
    one = _intersect_subspaces(bvec, fvec, 1)
    assert np.array_equal(one, _intersect_subspaces(bvec, fvec, 3)) and np.array_equal(one, _intersect_subspaces(bvec, fvec, 64))
    for i in range(nt):
        for r in range(nr):
            for j in range(nd):
                u = np.linalg.svd(bvec[i, :, :j + 1, r].T @ fvec[i, :, :nd - j, r])[0]
                assert np.abs(one[i, r, :, j] - (bvec[i, :, :j + 1, r] @ u)[:, 0]).max() < 1e-14


def test_estimator_member_groups(monkeypatch):
    """Records that do not fit one device window leave in member groups when the ensemble is large (lyapunov.py
    `_member_groups`): a group's record is one contiguous piece of the member-major result blocks.  The rule, on a fake device."""
    from qgs_amd.toolbox.lyapunov import LyapunovsEstimator

    class FakeTorch(object):
        class cuda(object):
            mem_get_info = staticmethod(lambda dev: (280 << 30, 288 << 30))
        device = staticmethod(lambda *a: None)

    class FakeModel(object):
        device = 0
    for k in ('QGS_HIP_RECORD_WINDOW_MB', 'QGS_HIP_RECORD_GROUP_MEMBERS'):
        monkeypatch.delenv(k, raising=False)
    est = LyapunovsEstimator(num_threads=1)
    est.n_dim, est.n_vec, est.n_records = 36, 36, 401
    per_member = 8 * 401 * (36 * 36 + 36 + 36)
    g, budget = est._member_groups(FakeTorch, FakeModel, 16384, False)
    assert g == 2048 and budget >= 2 * per_member * g                     # 72 GB: eight groups, each one window of all records
    est.n_records = 1001
    g, budget = est._member_groups(FakeTorch, FakeModel, 16384, False)
    assert g == 2048 and 4 * (budget // 2) <= (280 << 30) // 3            # 180 GB: window + staging of two groups within a third
    g, _ = est._member_groups(FakeTorch, FakeModel, 16384, True)          # with the matrices before the QR: twice the record
    assert g % 64 == 0 and 1024 <= g < 2048
    est.n_records = 11
    assert est._member_groups(FakeTorch, FakeModel, 16384, False) == (16384, None)     # fits one window: one pass
    est.n_records = 100001
    assert est._member_groups(FakeTorch, FakeModel, 16384, False) == (16384, None)     # groups would be < 1 024 members: windows of records
    est.n_records = 401
    assert est._member_groups(FakeTorch, FakeModel, 1500, False)[0] == 1500            # few members: windows of records
    monkeypatch.setenv('QGS_HIP_RECORD_WINDOW_MB', '512')
    assert est._member_groups(FakeTorch, FakeModel, 16384, False) == (16384, None)     # a budget set by hand keeps the windows
    monkeypatch.setenv('QGS_HIP_RECORD_GROUP_MEMBERS', '100')
    assert est._member_groups(FakeTorch, FakeModel, 16384, False)[0] == 128            # by hand: rounded up to whole wavefronts
