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
