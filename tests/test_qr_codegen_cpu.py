"""Build host (no GPU): the generated batched-QR kernels of the row and grid designs multiply with a DPP operand
(`v_fmac_f64_dpp ... row_newbcast`, inline assembly -- qgs_amd/csrc/codegen.cpp emit_dpp_fmacs).  A VGPR written by a VALU
instruction must not be read as a DPP operand within the next two wait states, and the compiler does not look inside inline
assembly for that: a register-allocator copy (v_accvgpr_read_b32, v_mov) placed right in front of such a statement is read
stale.  Round 5 found it as wrong factors at 44 x 40 (operands parked in accumulation registers).  Checked here on the
compiled instruction stream of shapes of every design, and on the generated source.

The factors themselves are checked against LAPACK on the GPU (tests/test_gpu_lyapunov.py), the algorithm being that of
np.linalg.qr as the reference uses it (qgs/toolbox/lyapunov.py:600-610)."""
import os
import re
import shutil
import subprocess

import pytest

from qgs_amd import _lib

HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'

# (rows, cols, design as the plan signature shows it)
SHAPES = [(36, 36, 'row'), (44, 40, 'row1'), (64, 20, 'row1'), (48, 48, 'row1'), (56, 20, 'row'), (20, 20, 'row'),
          (100, 16, 'row'),
          (228, 40, 'grid'), (64, 10, 'grid'), (64, 40, 'grid'), (20, 5, 'tile'), (64, 64, 'tile'), (36, 12, 'tile')]


def _design(sig):
    if 'g' in sig.split('r')[-1]:
        return 'grid'
    if sig.startswith('m4p'):
        return 'row1' if sig.endswith('o1') else 'row'
    return 'tile'


def _vregs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]$', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def dpp_hazards(asm_text):
    """(number of DPP instructions, list of (writer, reader) pairs less than two wait states apart)."""
    ins = []
    for line in asm_text.splitlines():
        line = line.split(';')[0].strip()
        if not line or line.startswith(('.', '//')) or line.endswith(':'):
            continue
        ins.append(line.replace(',', ' ').split())
    n_dpp, bad = 0, []
    for n, p in enumerate(ins):
        if not any(x.startswith('row_newbcast') for x in p):
            continue
        n_dpp += 1
        src = _vregs(p[2])
        waited, k = 0, n - 1
        while k >= 0 and waited < 2:
            q = ins[k]
            if q[0] == 's_nop':
                waited += int(q[1], 0) + 1
            else:
                if q[0].startswith('v_') and len(q) > 1 and (_vregs(q[1]) & src):
                    bad.append((' '.join(q), ' '.join(p)))
                    break
                waited += 1
            k -= 1
    return n_dpp, bad


def test_scanner_sees_the_hazard():
    asm = '''
        v_accvgpr_read_b32 v30, a18
        v_fmac_f64_dpp v[46:47], v[30:31], v[242:243] row_newbcast:0 row_mask:0xf bank_mask:0xf
        v_accvgpr_read_b32 v31, a19
        s_nop 1
        v_fmac_f64_dpp v[46:47], v[30:31], v[242:243] row_newbcast:0 row_mask:0xf bank_mask:0xf
        v_mov_b32_e32 v10, v3
        v_mov_b32_e32 v5, v4
        v_fmac_f64_dpp v[46:47], v[10:11], v[242:243] row_newbcast:0 row_mask:0xf bank_mask:0xf
        v_mov_b32_e32 v10, v3
        v_mov_b32_e32 v5, v4
        v_mov_b32_e32 v6, v4
        v_fmac_f64_dpp v[46:47], v[10:11], v[242:243] row_newbcast:0 row_mask:0xf bank_mask:0xf
    '''
    n, bad = dpp_hazards(asm)
    assert n == 4 and len(bad) == 2 and bad[0][0].startswith('v_accvgpr_read_b32 v30') and bad[1][0].startswith('v_mov_b32_e32 v10')


@pytest.mark.parametrize('rows,cols,design', SHAPES)
def test_generated_source_and_plan(rows, cols, design):
    src = _lib.qr_kernel_source(rows, cols)
    sig = src.splitlines()[0].split()[-1]
    assert _design(sig) == design, sig
    assert 'qgs_spec_qr_%dx%d' % (rows, cols) in src
    stmts = [l for l in src.splitlines() if 'v_fmac_f64_dpp' in l and 'asm volatile' in l]
    if design == 'tile':
        assert not stmts                                       # reflector through LDS: no DPP
        return
    assert stmts
    for l in stmts:
        # every statement waits before its first DPP instruction, holds at most eight of them, and nothing but those
        body = l.split('asm volatile("')[1].split('" :')[0]
        parts = body.split('\\n\\t')
        assert parts[0] == 's_nop 1' and 1 <= sum(p.startswith('v_fmac_f64_dpp') for p in parts) <= 8
        assert all(p == 's_nop 1' or p.startswith('v_fmac_f64_dpp') for p in parts)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='needs hipcc')
@pytest.mark.parametrize('rows,cols,design', [s for s in SHAPES if s[2] != 'tile'])
def test_compiled_kernel_has_no_dpp_hazard(rows, cols, design, tmp_path):
    src = tmp_path / 'qr.hip'
    src.write_text(_lib.qr_kernel_source(rows, cols))
    out = tmp_path / 'qr.s'
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-o', str(out), str(src)],
                   check=True, capture_output=True, timeout=600)
    asm = out.read_text()
    n, bad = dpp_hazards(asm)
    assert n > 100 and not bad, bad[:5]
    scratch = int(re.search(r'\.private_segment_fixed_size:\s*(\d+)', asm).group(1))
    vgpr = int(re.search(r'\.vgpr_count:\s*(\d+)', asm).group(1))
    if (rows, cols) in ((36, 36), (20, 20), (44, 40), (48, 48), (64, 20)):
        assert scratch == 0, (scratch, vgpr)                   # the shapes the bench and the profiles quote
    assert vgpr <= (512 if design == 'row1' else 256)
