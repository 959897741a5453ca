"""Build host (no GPU): the generated kernels hold inline assembly, and the compiler inserts the wait states an instruction
needs after another only between instructions it knows -- not around what is inside an `asm` statement.  The three cases the
generated code has (qgs_amd/csrc/codegen.cpp: emit_dpp_fmacs, qgs_store_row, qgs_store_row2), each closed inside its own
statement, are checked here on the COMPILED instruction stream:

* a VGPR written by a VALU instruction must not be read as a DPP operand within two wait states (nor a DPP instruction follow a
  VALU write of EXEC within five): the batched-QR kernels of the row and grid designs multiply with `v_fmac_f64_dpp ...
  row_newbcast`, and a register-allocator copy (v_accvgpr_read_b32, v_mov) right in front of such a statement was read stale --
  found in round 5 as wrong factors at 44 x 40 (operands parked in accumulation registers);
* an SGPR written by a VALU instruction (the reload of a spilled SGPR) must not address a vector-memory instruction within five
  wait states: the scalar-base stores of the steppers;
* the data registers of a store of more than 64 bits must not be written by a VALU instruction within two wait states: the
  128-bit stage-record stores of the tangent model's stepper.

The factors themselves are checked against LAPACK on the GPU (tests/test_gpu_lyapunov.py), the algorithm being that of
np.linalg.qr as the reference uses it (qgs/toolbox/lyapunov.py:600-610)."""
import glob
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


def _regs(tok, kind):
    m = re.match(r'%s\[(\d+):(\d+)\]$' % kind, tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'%s(\d+)$' % kind, tok)
    return {int(m.group(1))} if m else set()


def _instructions(asm_text):
    ins = []
    for line in asm_text.splitlines():
        line = line.split(';')[0].strip()
        if not line or line.startswith(('.', '//')) or line.endswith(':'):
            continue
        ins.append(line.replace(',', ' ').split())
    return ins


def _near(ins, n, step, need, hit):
    """Walk from instruction n in direction `step` while fewer than `need` wait states have passed; the first instruction for
    which hit(q) holds is returned."""
    waited, k = 0, n + step
    while 0 <= k < len(ins) and waited < need:
        q = ins[k]
        if q[0] == 's_nop':
            waited += int(q[1], 0) + 1
        else:
            if hit(q):
                return q
            waited += 1
        k += step
    return None


def _valu_writes(q, regs, kind='v'):
    return q[0].startswith('v_') and len(q) > 1 and bool(_regs(q[1], kind) & regs)


def hazards(asm_text):
    """Counts of the instructions concerned and the list of (kind, first instruction, second instruction) too close together."""
    ins = _instructions(asm_text)
    count = {'dpp': 0, 'wide_store': 0, 'vmem_sgpr': 0}
    bad = []
    for n, p in enumerate(ins):
        if any(x.startswith(('row_newbcast', 'row_shr', 'row_shl', 'row_bcast', 'quad_perm', 'row_ror', 'wave_')) for x in p):
            count['dpp'] += 1
            src = _regs(p[2], 'v')
            q = _near(ins, n, -1, 2, lambda q: _valu_writes(q, src))
            if q:
                bad.append(('dpp operand', ' '.join(q), ' '.join(p)))
            q = _near(ins, n, -1, 5, lambda q: q[0].startswith('v_') and len(q) > 1 and q[1] == 'exec')
            if q:
                bad.append(('dpp after exec', ' '.join(q), ' '.join(p)))
        if p[0].startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
            sg = set()
            for t in p[1:]:
                sg |= _regs(t, 's')
            if sg:
                count['vmem_sgpr'] += 1
                for r in sorted(sg):
                    # the LAST writer of the register counts: a scalar instruction that overwrote it in between has waited for
                    # the VALU write itself
                    q = _near(ins, n, -1, 5, lambda q: len(q) > 1 and q[0].startswith(('v_', 's_')) and r in _regs(q[1], 's'))
                    if q and q[0].startswith('v_'):
                        bad.append(('vmem address', ' '.join(q), ' '.join(p)))
                        break
            if re.search(r'store_(dwordx[34]|b96|b128)', p[0]):
                count['wide_store'] += 1
                data = _regs(p[2], 'v')
                q = _near(ins, n, +1, 2, lambda q: _valu_writes(q, data))
                if q:
                    bad.append(('store data', ' '.join(p), ' '.join(q)))
    return count, bad


def dpp_hazards(asm_text):
    count, bad = hazards(asm_text)
    return count['dpp'], [b[1:] for b in bad if b[0].startswith('dpp')]


def test_scanner_sees_the_hazards():
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
    asm = '''
        v_readlane_b32 s4, v40, 3
        v_readlane_b32 s5, v40, 4
        s_add_u32 s8, s4, s10
        s_addc_u32 s9, s5, s11
        global_store_dwordx2 v1, v[2:3], s[8:9]
        v_readlane_b32 s6, v40, 5
        v_mov_b32_e32 v9, 0
        global_store_dwordx2 v1, v[2:3], s[6:7]
        global_store_dwordx4 v1, v[4:7], s[8:9]
        v_mov_b32_e32 v6, 0
        global_store_dwordx4 v1, v[4:7], s[8:9]
        s_nop 1
        v_mov_b32_e32 v6, 0
        v_cmpx_gt_u32_e32 exec, v1, v2
        s_nop 2
        v_fmac_f64_dpp v[46:47], v[30:31], v[242:243] row_newbcast:0 row_mask:0xf bank_mask:0xf
    '''
    count, bad = hazards(asm)
    assert count == {'dpp': 1, 'wide_store': 2, 'vmem_sgpr': 4}
    assert [b[0] for b in bad] == ['vmem address', 'store data', 'dpp after exec']


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
    count, bad = hazards(asm)
    assert count['dpp'] > 100 and not bad, bad[:5]
    scratch = int(re.search(r'\.private_segment_fixed_size:\s*(\d+)', asm).group(1))
    vgpr = int(re.search(r'\.vgpr_count:\s*(\d+)', asm).group(1))
    if (rows, cols) in ((36, 36), (20, 20), (44, 40), (48, 48), (64, 20)):
        assert scratch == 0, (scratch, vgpr)                   # the shapes the bench and the profiles quote
    assert vgpr <= (512 if design == 'row1' else 256)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='needs hipcc')
def test_compiled_model_kernels_have_no_inline_asm_hazard(tmp_path):
    """Every register-resident kernel of MAOOAM-36 (steppers with and without records and stage records, tangent and adjoint
    kernels, general tableaus) and the LDS-resident ones, from the generator as the library runs it."""
    import numpy as np
    from conftest import GOLDEN_DIR
    csrc = os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), 'csrc')
    dump = str(tmp_path / 'codegen_dump')
    subprocess.run(['g++', '-O1', '-std=c++17', '-o', dump, os.path.join(csrc, 'codegen_dump.cpp')] + [f for f in sorted(glob.glob(os.path.join(csrc, 'codegen*.cpp'))) if not f.endswith('codegen_dump.cpp')],
                   check=True, timeout=600)
    g = np.load(os.path.join(GOLDEN_DIR, 'm36.npz'))
    txt = tmp_path / 'm36.txt'
    with open(txt, 'w') as f:
        for kind, coo, v in (('T', g['coo'], g['val']), ('J', g['jcoo'], g['jval'])):
            for c, x in zip(coo, v):
                f.write('%s %s %s\n' % (kind, ' '.join(str(int(q)) for q in c), float(x).hex()))
    src = tmp_path / 'm36.hip'
    with open(src, 'w') as f:
        subprocess.run([dump, str(int(g['ndim'])), str(txt), 'all'], check=True, stdout=f, stderr=subprocess.DEVNULL, timeout=600)
    out = tmp_path / 'm36.s'
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-o', str(out), str(src)],
                   check=True, capture_output=True, timeout=1200)
    asm = out.read_text()
    count, bad = hazards(asm)
    assert count['wide_store'] >= 72 and count['vmem_sgpr'] >= 200 and not bad, bad[:5]
    # the statements as the generator writes them: the scalar move in front of every scalar-base store, the wait behind the wide ones
    ins = _instructions(asm)
    for n, p in enumerate(ins):
        if p[0].startswith('global_store') and _regs(p[-1], 's') and not p[-1].startswith('off'):
            if ins[n - 1][0] == 's_mov_b64':
                assert _regs(ins[n - 1][1], 's') == _regs(p[-1], 's')
            if p[0] == 'global_store_dwordx4':
                assert ins[n - 1][0] == 's_mov_b64' and ins[n + 1][:2] == ['s_nop', '1']


# ---- the hand-scheduled kernels (round 6): their bodies are assembly the generator writes, so the rules are checked on that text ----
def _asm_statements(source):
    """The instruction lists of the multi-line `asm volatile(` statements of a generated source (one list per statement)."""
    out, cur = [], None
    for line in source.splitlines():
        t = line.strip()
        if t == 'asm volatile(':
            cur = []
        elif cur is not None:
            if t.startswith('"'):
                txt = t.strip('"').replace('\\n', '')
                if txt and not txt.startswith('.'):
                    cur.append(txt.replace(',', ' ').split())
            else:
                out.append(cur)
                cur = None
    return out


def _strict_dpp_and_lane_read_hazards(ins):
    """(DPP instructions, lane reads, violations): the compiler's DPP rule -- no VGPR a VALU instruction wrote within two wait states
    may be read by a DPP instruction, accumulator and second operand included -- and the rule found in round 6: a lane read
    (v_readfirstlane_b32) must not follow the VALU instruction that produces its source within two wait states (a stale low word was
    measured behind v_mul_f64)."""
    n_dpp = n_lane = 0
    bad = []
    for n, p in enumerate(ins):
        if any(x.startswith('row_newbcast') for x in p):
            n_dpp += 1
            regs = set()
            for t in p[1:4]:
                regs |= _regs(t.lstrip('-'), 'v')
            q = _near(ins, n, -1, 2, lambda q: _valu_writes(q, regs))
            if q:
                bad.append(('dpp', ' '.join(q), ' '.join(p)))
        if p[0] == 'v_readfirstlane_b32' and not p[2].startswith('%'):
            n_lane += 1
            src = _regs(p[2], 'v')
            q = _near(ins, n, -1, 2, lambda q: q[0] != 'v_readfirstlane_b32' and _valu_writes(q, src))
            if q:
                bad.append(('lane read', ' '.join(q), ' '.join(p)))
    return n_dpp, n_lane, bad


def test_strict_scanner_sees_both_rules():
    ok = [['v_mul_f64', 'v[4:5]', 'v[0:1]', 'v[2:3]'], ['s_nop', '1'], ['v_readfirstlane_b32', 's3', 'v4'],
          ['v_fma_f64', 'v[8:9]', 'v[0:1]', 'v[2:3]', 'v[8:9]'], ['v_mov_b64', 'v[20:21]', '0'], ['v_mov_b64', 'v[22:23]', '0'],
          ['v_fmac_f64_dpp', 'v[10:11]', 'v[12:13]', 'v[8:9]', 'row_newbcast:3', 'row_mask:0xf', 'bank_mask:0xf']]
    assert _strict_dpp_and_lane_read_hazards(ok)[2] == []
    stale = [['v_mul_f64', 'v[4:5]', 'v[0:1]', 'v[2:3]'], ['v_readfirstlane_b32', 's3', 'v4']]
    assert _strict_dpp_and_lane_read_hazards(stale)[2][0][0] == 'lane read'
    acc = [['v_fmac_f64_dpp', 'v[10:11]', 'v[12:13]', 'v[8:9]', 'row_newbcast:3'], ['v_mov_b64', 'v[20:21]', '0'],
           ['v_fmac_f64_dpp', 'v[10:11]', 'v[14:15]', 'v[8:9]', 'row_newbcast:4']]
    assert _strict_dpp_and_lane_read_hazards(acc)[2][0][0] == 'dpp'


@pytest.mark.parametrize('tensor, knobs, kernel, min_dpp', [('t228', [], 'qgs_spec_rkldsa8', 15000), ('a36', ['tglasm=1'], 'qgs_spec_tglpa_s4', 1000),
                                                            ('rp20', ['tglasm=1', 'stages=2'], 'qgs_spec_tglpa_s2', 200),
                                                            ('t228', ['ldstglasm=1', 'all'], 'qgs_spec_tglldsa8', 30000),
                                                            ('t228', ['ldstglasm=1', 'all'], 'qgs_spec_adjldsa8', 30000)])
def test_hand_scheduled_bodies_keep_the_dpp_and_lane_read_rules(tmp_path, tensor, knobs, kernel, min_dpp):
    """The stage body of the LDS-resident stepper (MAOOAM 6x6) and the hand-scheduled tangent kernel, as the generator writes them:
    every DPP instruction two wait states behind the last VALU write of anything it reads, every lane read two wait states behind
    the instruction that produced its source."""
    import numpy as np
    from conftest import GOLDEN_DIR
    csrc = os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), 'csrc')
    dump = str(tmp_path / 'codegen_dump')
    subprocess.run(['g++', '-O1', '-std=c++17', '-o', dump, os.path.join(csrc, 'codegen_dump.cpp')] + [f for f in sorted(glob.glob(os.path.join(csrc, 'codegen*.cpp'))) if not f.endswith('codegen_dump.cpp')],
                   check=True, timeout=600)
    g = np.load(os.path.join(GOLDEN_DIR, tensor + '.npz'))
    txt = tmp_path / 'tensor.txt'
    with open(txt, 'w') as f:
        for kind, coo, v in (('T', g['coo'], g['val']), ('J', g['jcoo'], g['jval'])):
            for c, x in zip(coo, v):
                f.write('%s %s %s\n' % (kind, ' '.join(str(int(q)) for q in c), float(x).hex()))
    src = subprocess.run([dump, str(int(g['ndim'])), str(txt), 'tables'] + knobs, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                         timeout=600).stdout.decode()
    chunks = ['#ifndef QGS_SPEC_PRELUDE' + c for c in src.split('#ifndef QGS_SPEC_PRELUDE')[1:]]
    mine = [c for c in chunks if re.search(r'\b' + kernel + r'\(', c)]
    assert mine, sorted(set(re.findall(r'(qgs_spec_\w+)\(', src)))
    n_dpp = n_lane = 0
    for ins in _asm_statements(mine[0]):
        if len(ins) < 50:
            continue                        # (the one-line statements of the prelude)
        d, l, bad = _strict_dpp_and_lane_read_hazards(ins)
        assert not bad, bad[:5]
        n_dpp += d
        n_lane += l
    assert n_dpp >= min_dpp and (n_lane > 0 or 'lds' in kernel), (n_dpp, n_lane)
