"""CPU: the kernel generator (a 1.9 k-line string builder, qgs_amd/csrc/codegen.cpp) under AddressSanitizer + UBSan on every golden
tensor, every emitter (register-resident, general-tableau, LDS-resident, tangent / adjoint, batched QR).  GPU sanitizers are not
available on the pool; the generator is host code and is checked here."""
import glob
import os
import subprocess

import numpy as np
import pytest

from conftest import CONFIGS, REPO, load_golden

CSRC = os.path.join(REPO, 'qgs_amd', 'csrc')


@pytest.fixture(scope='module')
def dump_binary(tmp_path_factory):
    out = str(tmp_path_factory.mktemp('san') / 'codegen_dump_san')
    subprocess.run(['g++', '-O1', '-g', '-std=c++17', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined',
                    '-fno-omit-frame-pointer', '-o', out, os.path.join(CSRC, 'codegen_dump.cpp')] + [f for f in sorted(glob.glob(os.path.join(CSRC, 'codegen*.cpp'))) if not f.endswith('codegen_dump.cpp')],
                   check=True, timeout=600)
    return out


def _tensor_text(g, path):
    rank = g['coo'].shape[1]
    tag = ('T', 'J') if rank == 3 else ('T5', 'J5')
    with open(path, 'w') as f:
        for kind, coo, val in ((tag[0], g['coo'], g['val']), (tag[1], g['jcoo'], g['jval'])):
            for c, v in zip(coo, val):
                f.write('%s %s %s\n' % (kind, ' '.join(str(int(q)) for q in c), float(v).hex()))


@pytest.mark.parametrize('name', CONFIGS)
def test_generator_is_clean_under_asan_ubsan(dump_binary, tmp_path, name):
    g = load_golden(name)
    txt = str(tmp_path / (name + '.txt'))
    _tensor_text(g, txt)
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    for extra in (['all'], ['stages=2', 'split=2']):
        p = subprocess.run([dump_binary, str(g.ndim), txt] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
        err = p.stderr.decode()
        assert p.returncode == 0, err[-3000:]
        assert 'runtime error' not in err and 'AddressSanitizer' not in err and 'LeakSanitizer' not in err, err[-3000:]
        src = p.stdout.decode()
        assert ('qgs_spec_rk_s' in src) == (g.ndim <= 64)
        if extra == ['all']:
            assert 'qgs_spec_rklds' in src and 'qgs_spec_qr_' in src
