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
