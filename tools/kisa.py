"""Build host: ISA statistics of one generated kernel, compiled by the same helper (system hiprtc) the library uses.

    python tools/kisa.py <bench|m36|t228|...> <kernel name, e.g. qgs_spec_tglp_s4> [KEY=VALUE generator knobs ...] [--keep DIR]

Generates the kernel for the tensor (qgs_amd/csrc/codegen_dump, built on first use), compiles it with qgs_amd/qgs_kcompile,
disassembles it with llvm-objdump and prints VGPR / scratch use and an instruction histogram (fp64 arithmetic, moves,
accumulation-register moves, lane moves, LDS, vector memory, scalar memory, waits), whole kernel and hottest loop."""
import collections
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
CSRC = os.path.join(REPO, 'qgs_amd', 'csrc')
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'


def tensors(name):
    if name == 'bench':
        from bench import load_model_tensors
        ndim, coo, val, jcoo, jval, _ = load_model_tensors()
        return ndim, coo, val, jcoo, jval
    g = np.load(os.path.join(REPO, 'tests', 'golden', name + '.npz'))
    return int(g['ndim']), g['coo'], g['val'], g['jcoo'], g['jval']


def classify(mn):
    if re.match(r'v_(fma|fmac|mul|add|pk_fma|pk_mul|pk_add)_f64', mn):
        return 'fp64'
    if mn.startswith('v_accvgpr'):
        return 'accvgpr'
    if mn.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')):
        return 'lane'
    if mn.startswith('v_mov') or mn.startswith('v_pk_mov'):
        return 'v_mov'
    if mn.startswith('ds_'):
        return 'lds'
    if mn.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'scratch' if mn.startswith('scratch_') else 'vmem'
    if mn.startswith('s_load') or mn.startswith('s_buffer_load'):
        return 'smem'
    if mn.startswith('s_waitcnt'):
        return 's_waitcnt'
    if mn.startswith('s_barrier'):
        return 's_barrier'
    if mn.startswith('v_'):
        return 'valu_other'
    if mn.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    args = [a for a in sys.argv[1:]]
    keep = json_path = None
    if '--json' in args:
        i = args.index('--json')
        json_path = args[i + 1]
        del args[i:i + 2]
    if '--keep' in args:
        i = args.index('--keep')
        keep = args[i + 1]
        del args[i:i + 2]
    name, kernel = args[0], args[1]
    extra = args[2:]
    ndim, coo, val, jcoo, jval = tensors(name)
    dump = os.path.join(CSRC, 'codegen_dump')
    subprocess.check_call(['make', '-C', CSRC, '-s', 'codegen_dump'])
    work = keep or tempfile.mkdtemp(prefix='kisa_')
    os.makedirs(work, exist_ok=True)
    txt = os.path.join(work, 'tensor.txt')
    rank = coo.shape[1]
    tag = ('T', 'J') if rank == 3 else ('T5', 'J5')
    with open(txt, 'w') as f:
        for kind, c_, v_ in ((tag[0], coo, val), (tag[1], jcoo, jval)):
            for c, v in zip(c_, v_):
                f.write('%s %s %s\n' % (kind, ' '.join(str(int(q)) for q in c), float(v).hex()))
    src_all = subprocess.run([dump, str(ndim), txt, 'all'] + extra, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
    chunks = ['#ifndef QGS_SPEC_PRELUDE' + c for c in src_all.split('#ifndef QGS_SPEC_PRELUDE')[1:]]
    mine = [c for c in chunks if re.search(r'void\s+__launch_bounds__\([^)]*\)\s+' + re.escape(kernel) + r'\(', c)]
    if not mine:
        names = sorted(set(re.findall(r'(qgs_spec_\w+)\(', src_all)))
        sys.exit('kernel %s not generated; available: %s' % (kernel, ', '.join(names)))
    src = os.path.join(work, kernel + '.hip')
    with open(src, 'w') as f:
        f.write(mine[0])
    obj = os.path.join(work, kernel + '.hsaco')
    flags = []
    m = re.search(r'// qgs-compile-flags:(.*)', mine[0])
    if m:
        flags = m.group(1).split()
    subprocess.check_call([os.path.join(REPO, 'qgs_amd', 'qgs_kcompile'), 'gfx950', src, obj] + flags)
    notes = subprocess.run([READELF, '--notes', obj], stdout=subprocess.PIPE).stdout.decode()
    for key in ('.vgpr_count', '.agpr_count', '.sgpr_count', '.private_segment_fixed_size', '.group_segment_fixed_size', '.vgpr_spill_count', '.sgpr_spill_count'):
        mm = re.search(re.escape(key) + r':\s*(\d+)', notes)
        if mm:
            print('%-30s %s' % (key, mm.group(1)))
    dis = subprocess.run([OBJDUMP, '-d', obj], stdout=subprocess.PIPE).stdout.decode()
    with open(os.path.join(work, kernel + '.s'), 'w') as f:
        f.write(dis)
    insts = []                                               # (address, mnemonic, operands)
    for ln in dis.splitlines():
        mm = re.match(r'\s+(\S+)\s+(.*?)\s*//\s*([0-9A-Fa-f]+):', ln)
        if mm:
            insts.append((int(mm.group(3), 16), mm.group(1), mm.group(2)))
    hist = collections.Counter(classify(mn) for _, mn, _ in insts)
    print('whole kernel: %d instructions' % len(insts), dict(hist.most_common()))
    # hottest loop: the SMALLEST backward-branch span that still holds more than half of the kernel's fp64 instructions
    addr_index = {a: i for i, (a, _, _) in enumerate(insts)}
    is_fp64 = [classify(mn) == 'fp64' for _, mn, _ in insts]
    prefix = [0]
    for q in is_fp64:
        prefix.append(prefix[-1] + int(q))
    best = None
    for i, (a, mn, ops) in enumerate(insts):
        if mn.startswith('s_cbranch') or mn == 's_branch':
            mm = re.search(r'(-?\d+)\s*$', ops)
            if mm:
                off = int(mm.group(1))
                if off >= 32768:
                    off -= 65536
                tgt = a + 4 + 4 * off
                if tgt in addr_index and addr_index[tgt] < i:
                    j = addr_index[tgt]
                    if 2 * (prefix[i + 1] - prefix[j]) > prefix[-1]:
                        span = i - j
                        if best is None or span < best[0]:
                            best = (span, j, i)
    if best:
        loop = insts[best[1]:best[2] + 1]
        h2 = collections.Counter(classify(mn) for _, mn, _ in loop)
        print('hot loop: %d instructions' % len(loop), dict(h2.most_common()))
        valu = sum(v for k, v in h2.items() if k in ('fp64', 'accvgpr', 'lane', 'v_mov', 'valu_other'))
        print('    VALU %d of which fp64 %d (%.1f %%)' % (valu, h2['fp64'], 100.0 * h2['fp64'] / max(1, valu)))
    print('files in', work)
    if json_path:
        import json
        data = {}
        if os.path.exists(json_path):
            with open(json_path) as f:
                data = json.load(f)
        regs = {k.strip('.'): int(re.search(re.escape(k) + r':\s*(\d+)', notes).group(1)) for k in
                ('.vgpr_count', '.agpr_count', '.private_segment_fixed_size', '.group_segment_fixed_size') if re.search(re.escape(k) + r':\s*(\d+)', notes)}
        data['%s:%s' % (name, kernel)] = {'tensor': name, 'kernel': kernel, 'knobs': extra, 'registers': regs,
                                         'whole_kernel': dict(hist), 'hot_loop': dict(h2) if best else None,
                                         'hot_loop_instructions': len(loop) if best else None}
        with open(json_path, 'w') as f:
            json.dump(data, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
