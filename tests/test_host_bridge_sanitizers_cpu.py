"""CPU: the host side of the result path -- qgs_amd/csrc/host_bridge.cpp: bounce rings, the per-device drain thread, the pool of
gather / scatter threads -- under ThreadSanitizer and under AddressSanitizer + UBSan, behind a test double of the HIP runtime
(tests/stub_hip: device memory = heap memory, a stream = a worker thread that runs its operations in order and asynchronously to
its caller).  The driver (tests/host_bridge_driver.cpp) replays the library's traffic patterns: blocking copies around the block and
task-grain sizes, record windows with two device buffers and `ready` events, rows larger than a bounce block, 8-byte runs a page
apart, member groups (hundreds of short jobs), a failing transfer, and all of it from several shard threads on several devices at
once plus a second thread on device 0.  GPU-side sanitizers do not exist on the pool (and are not needed for host arithmetic)."""
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, 'qgs_amd', 'csrc')


def _build(tmp_path, sanitizer):
    out = str(tmp_path / ('host_bridge_' + sanitizer.split(',')[0]))
    subprocess.run(['g++', '-O1', '-g', '-std=c++17', '-fsanitize=' + sanitizer, '-fno-omit-frame-pointer',
                    '-I', os.path.join(REPO, 'tests', 'stub_hip'), '-I', CSRC, '-o', out,
                    os.path.join(REPO, 'tests', 'host_bridge_driver.cpp'), os.path.join(CSRC, 'host_bridge.cpp'), '-lpthread'],
                   check=True, timeout=600)
    return out


@pytest.mark.parametrize('sanitizer, env', [
    ('thread', {'TSAN_OPTIONS': 'halt_on_error=1 second_deadlock_stack=1'}),
    ('address,undefined', {'ASAN_OPTIONS': 'detect_leaks=0', 'UBSAN_OPTIONS': 'halt_on_error=1'}),   # (the driver keeps its stub streams / events)
])
def test_host_bridge_under_sanitizers(tmp_path, sanitizer, env):
    exe = _build(tmp_path, sanitizer)
    p = subprocess.run([exe, '3'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900,
                       env=dict(os.environ, QGS_HIP_HOST_THREADS='4', **env))
    out, err = p.stdout.decode(), p.stderr.decode()
    assert p.returncode == 0, (out[-2000:], err[-4000:])
    assert out.startswith('OK shards=3'), out
    assert 'WARNING: ThreadSanitizer' not in err and 'ERROR: AddressSanitizer' not in err and 'runtime error' not in err, err[-4000:]
