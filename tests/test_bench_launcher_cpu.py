"""CPU: bench.py's launcher refuses cleanly when fewer GPUs are visible than requested (no GPU in this container: 0), and
the kernel-compile helper identifies its compiler."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_when_gpus_are_missing():
    import torch
    n = torch.cuda.device_count()
    p = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', str(n + 2), '--steps', '1'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode not in (0, 2)
    assert ('%d GPUs requested, %d visible' % (n + 2, n)) in p.stderr.decode()
    assert p.stdout.decode().strip() == ''


def test_compile_helper_reports_the_system_hiprtc():
    helper = os.path.join(REPO, 'qgs_amd', 'qgs_kcompile')
    assert os.access(helper, os.X_OK), 'build it: make -C qgs_amd/csrc'
    out = subprocess.run([helper, '--version'], stdout=subprocess.PIPE, timeout=60, check=True).stdout.decode().strip()
    assert out.startswith('hiprtc') and 'libhiprtc.so.' in out
    ldd = subprocess.run(['ldd', helper], stdout=subprocess.PIPE, timeout=60, check=True).stdout.decode()
    assert '/opt/rocm' in [ln for ln in ldd.splitlines() if 'libhiprtc' in ln][0]


def test_cache_key_separates_compilers(tmp_path):
    """The same source compiled by the helper and by the in-process hiprtc lands in two different cache entries."""
    code = ("import sys, numpy as np\n"
            "sys.path.insert(0, %r)\n"
            "from qgs_amd import _lib\n"
            "g = np.load(%r)\n"
            "_lib.prebuild(int(g['ndim']), g['coo'], g['val'], None, None, stage_counts=(2,))\n"
            % (REPO, os.path.join(REPO, 'tests', 'golden', 'rp20.npz')))
    counts = []
    for inproc in ('0', '1'):
        subprocess.run([sys.executable, '-c', code], check=True, timeout=900,
                       env=dict(os.environ, QGS_HIP_CACHE_DIR=str(tmp_path), QGS_HIP_INPROC_RTC=inproc))
        counts.append(len([f for f in os.listdir(str(tmp_path)) if f.endswith('.hsaco')]))
    assert counts[0] > 0 and counts[1] == 2 * counts[0]


def test_launcher_starts_one_child_per_rank_and_reports_their_failure():
    """`launch_ranks` with the GPU count faked to 2 (none here): both children start as ranks (RANK / WORLD_SIZE set, so they do
    not try to launch again), find no GPU, exit 1 -- and the parent says which ranks failed instead of hanging or exiting 2."""
    import pytest
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip('the premise of this test is a host without GPUs (the build container)')
    code = ("import sys\n"
            "sys.path.insert(0, %r)\n"
            "import bench\n"
            "bench.visible_gpus = lambda: 2\n"
            "sys.exit(bench.launch_ranks(2, ['--gpus', '2', '--steps', '1', '--warmup', '0']))\n" % REPO)
    p = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    err = p.stderr.decode()
    assert p.returncode == 1, err[-2000:]
    assert 'no GPU visible' in err and 'ranks failed (rank, exit code): [(0, 1), (1, 1)]' in err
    assert p.stdout.decode().strip() == ''


_PREBUILD = ("import sys, numpy as np\n"
             "sys.path.insert(0, %r)\n"
             "from qgs_amd import _lib\n"
             "g = np.load(%r)\n"
             "_lib.prebuild(int(g['ndim']), g['coo'], g['val'], None, None, stage_counts=(2,))\n"
             "print('PREBUILT')\n" % (REPO, os.path.join(REPO, 'tests', 'golden', 'rp20.npz')))


def _hsaco(d):
    return sorted(f for f in os.listdir(str(d)) if f.endswith('.hsaco'))


def test_unwritable_kernel_cache_still_compiles(tmp_path):
    """A cache directory that cannot be written (read-only shared install; here: a path below a regular file, which not even
    root can create): every miss compiles through temp files under $TMPDIR and the publish is skipped -- no failure, nothing
    left behind."""
    blocker, scratch = tmp_path / 'blocker', tmp_path / 'scratch'
    blocker.write_text('not a directory')
    scratch.mkdir()
    p = subprocess.run([sys.executable, '-c', _PREBUILD], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900,
                       env=dict(os.environ, QGS_HIP_CACHE_DIR=str(blocker / 'kcache'), TMPDIR=str(scratch)))
    assert p.returncode == 0 and b'PREBUILT' in p.stdout, p.stderr.decode()[-2000:]
    assert os.listdir(str(scratch)) == []


def test_helper_that_cannot_run_falls_back_to_the_in_process_compiler(tmp_path):
    """`--version` failing (e.g. the helper's RPATH does not resolve on this host) selects the in-process hiprtc AND its cache
    identity; a helper that answers `--version` but dies while compiling is dropped for the rest of the process, and what the
    in-process compiler produced is filed under ITS identity, never under the helper's."""
    inproc = tmp_path / 'inproc'
    subprocess.run([sys.executable, '-c', _PREBUILD], check=True, timeout=900,
                   env=dict(os.environ, QGS_HIP_CACHE_DIR=str(inproc), QGS_HIP_INPROC_RTC='1'))
    want = _hsaco(inproc)
    assert want
    # (a) no usable helper at all
    a = tmp_path / 'a'
    p = subprocess.run([sys.executable, '-c', _PREBUILD], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900,
                       env=dict(os.environ, QGS_HIP_CACHE_DIR=str(a), QGS_HIP_HELPER='/bin/false'))
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert b'--version failed' in p.stderr and _hsaco(a) == want
    # (b) a helper that identifies itself and then fails for a non-compile reason (exit code 3)
    fake = tmp_path / 'fake_helper.sh'
    fake.write_text('#!/bin/sh\nif [ "$1" = "--version" ]; then echo hiprtc9.9-fake; exit 0; fi\nexit 3\n')
    os.chmod(str(fake), 0o755)
    b = tmp_path / 'b'
    p = subprocess.run([sys.executable, '-c', _PREBUILD], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900,
                       env=dict(os.environ, QGS_HIP_CACHE_DIR=str(b), QGS_HIP_HELPER=str(fake)))
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert b'no longer usable' in p.stderr and _hsaco(b) == want
    # (c) a real compile error (exit code 1) is reported, not papered over by a second compiler
    fake.write_text('#!/bin/sh\nif [ "$1" = "--version" ]; then echo hiprtc9.9-fake; exit 0; fi\necho "error: nope" >&2\nexit 1\n')
    c = tmp_path / 'c'
    p = subprocess.run([sys.executable, '-c', _PREBUILD], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900,
                       env=dict(os.environ, QGS_HIP_CACHE_DIR=str(c), QGS_HIP_HELPER=str(fake)))
    assert p.returncode != 0 and b'error: nope' in p.stderr and not os.path.isdir(str(c)) or _hsaco(c) == []


def test_bench_default_workloads_follow_the_baseline_configs():
    """`bench.py --gpus 8` runs BASELINE configs[4] (131 072 members per GPU = 1 048 576), every other N configs[1] per GPU; the
    launcher counts GPUs without loading HIP (none in this container)."""
    sys.path.insert(0, REPO)
    import bench
    assert [bench.default_members(n) for n in (1, 2, 4, 8)] == [65536, 65536, 65536, 131072]
    assert 8 * bench.default_members(8) == 1048576
    assert bench.visible_gpus() == 0


def _fake_kfd(tmp_path, simd_counts):
    for i, c in enumerate(simd_counts):
        d = tmp_path / 'nodes' / str(i)
        d.mkdir(parents=True)
        (d / 'properties').write_text('cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n' % (0 if c else 64, c))
    return str(tmp_path / 'nodes')


def test_gpu_count_from_a_kfd_tree_and_the_visibility_masks(tmp_path):
    """An 8-GPU node as the launcher parent sees it (two CPU nodes + eight GPU nodes in the KFD topology), under the masks an
    operator may have set: indices, uuids, an empty value, an invalid entry that ends the list, ROCr's mask and HIP's together."""
    sys.path.insert(0, REPO)
    import bench
    nodes = _fake_kfd(tmp_path, [0, 0] + [1024] * 8)
    assert bench.visible_gpus(nodes, env={}) == 8
    assert bench.visible_gpus(nodes, env={'ROCR_VISIBLE_DEVICES': '0,1,2,3'}) == 4
    assert bench.visible_gpus(nodes, env={'ROCR_VISIBLE_DEVICES': 'GPU-1fa2,GPU-77c0'}) == 2
    assert bench.visible_gpus(nodes, env={'HIP_VISIBLE_DEVICES': ''}) == 0
    assert bench.visible_gpus(nodes, env={'ROCR_VISIBLE_DEVICES': '0,1,-1,2'}) == 2          # the runtimes stop at the first invalid entry
    assert bench.visible_gpus(nodes, env={'ROCR_VISIBLE_DEVICES': '0,1,9'}) == 2            # (an index beyond the node)
    assert bench.visible_gpus(nodes, env={'ROCR_VISIBLE_DEVICES': '0,0,1'}) == 1            # (a repeated entry)
    assert bench.visible_gpus(nodes, env={'ROCR_VISIBLE_DEVICES': '0,1,2,3', 'HIP_VISIBLE_DEVICES': '0,1'}) == 2
    assert bench.visible_gpus(nodes, env={'ROCR_VISIBLE_DEVICES': '4,5', 'CUDA_VISIBLE_DEVICES': '0,1,2'}) == 2
    assert bench.visible_gpus(_fake_kfd(tmp_path / 'cpu_only', [0, 0]), env={}) == 0


_STUB_RANK = """import os, sys, time
rank = int(os.environ['RANK'])
assert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1' and os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
mode = sys.argv[1]
if mode == 'ok':
    if rank == 0:
        print('{"metric": "stub", "rank0": true}')
    sys.exit(0)
if mode == 'die_before_rendezvous':
    if rank == 1:
        sys.exit(7)                     # e.g. its GPU is missing: it never reaches init_process_group
    time.sleep(600)                     # rank 0 waits in the rendezvous for a peer that will never come
"""


def _run_launcher(tmp_path, mode, grace=0.5):
    stub = tmp_path / 'stub_rank.py'
    stub.write_text(_STUB_RANK)
    code = ("import sys, time\n"
            "sys.path.insert(0, %r)\n"
            "import bench\n"
            "bench.visible_gpus = lambda *a, **k: 2\n"
            "t0 = time.time()\n"
            "rc = bench.launch_ranks(2, [%r], program=%r, grace=%r)\n"
            "print('ELAPSED %%.1f' %% (time.time() - t0), file=sys.stderr)\n"
            "sys.exit(rc)\n" % (REPO, mode, str(stub), grace))
    return subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)


def test_launcher_relays_rank_zero_and_exits_zero(tmp_path):
    p = _run_launcher(tmp_path, 'ok')
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert p.stdout.decode().strip() == '{"metric": "stub", "rank0": true}'


def test_a_rank_that_dies_before_the_rendezvous_ends_the_job(tmp_path):
    """Rank 1 exits before `init_process_group`; rank 0 would wait for it for ever.  The parent notices, gives the peers their grace
    period, kills exactly the processes it started and exits 1 within seconds -- naming the rank that failed."""
    p = _run_launcher(tmp_path, 'die_before_rendezvous')
    err = p.stderr.decode()
    assert p.returncode == 1, err[-2000:]
    assert '(1, 7)' in err and 'ranks failed' in err
    elapsed = float([ln for ln in err.splitlines() if ln.startswith('ELAPSED')][0].split()[1])
    assert elapsed < 30.0, err
    assert p.stdout.decode().strip() == ''


def test_traffic_lookup_never_mixes_grid_sizes(tmp_path):
    """`roofline.traffic` is a committed PMC figure looked up by (kernel, work-items of the launch, duration): two grid sizes of one
    kernel name are two entries; a grid the table does not hold gives None, never another grid's bytes (the headline
    carried the 1 048 576-member launch's 604 MB)."""
    sys.path.insert(0, REPO)
    import bench
    table = {'k@grid65536': {'hbm_bytes_per_launch': 11, 'mean_ms': 0.5, 'grid': 65536},
             'k@grid65536/class1': {'hbm_bytes_per_launch': 38, 'mean_ms': 4.4, 'grid': 65536},
             'k@grid1048576': {'hbm_bytes_per_launch': 604, 'mean_ms': 69.0, 'grid': 1048576},
             'k': {'hbm_bytes_per_launch': 604, 'mean_ms': 69.0, 'grid': 1048576}}                     # (a bare key of an older table: ignored)
    get = lambda grid, ms: (bench.traffic_entry('k', grid, ms, table) or {}).get('hbm_bytes_per_launch')
    assert get(65536, 4.3) == 38 and get(65536, 0.48) == 11 and get(1048576, 68.7) == 604
    assert get(131072, 8.6) is None and get(65536, 40.0) is None
    assert len({get(65536, 4.3), get(1048576, 68.7)}) == 2
    # the committed table: every entry is keyed by kernel AND grid, and no two grids of a kernel share a value
    real = bench._traffic_table()
    assert real and all('@grid' in k for k in real), sorted(real)
    by_kernel = {}
    for k, e in real.items():
        by_kernel.setdefault(k.split('@')[0], {}).setdefault(e['grid'], set()).add(e['hbm_bytes_per_launch'])
    for kern, grids in by_kernel.items():
        vals = [frozenset(v) for v in grids.values()]
        assert len(vals) == len(set(vals)), kern
    assert bench.launch_threads('qgs_spec_rk_s4', 65536) == 65536 and bench.launch_threads('qgs_spec_rk_s4', 1048576) == 1048576
    assert bench.launch_threads('qgs_spec_rkldsa8', 65536) == 1024 * 512 and bench.launch_threads('qgs_spec_rklds16', 65536) == 1024 * 1024
    assert bench.launch_threads('qgs_spec_qr_36x36', 16384) == 262144 and bench.launch_threads('qgs_spec_tglp_s4', 16384, 36) == 589824
