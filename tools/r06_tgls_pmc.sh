#!/bin/bash
# GPU box, developer build: PMC passes over config 4 (tools/tgls_ab.py: 16 384 members x 36 columns x 10 sub-steps, 100 calls per sample)
# with the compiler-scheduled pair kernel and the hand-scheduled one.  Output: gpurun_out/r06_tgls_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_tgls_pmc
rm -rf $O; mkdir -p $O
export RK_AB_LIB=$R/qgs_amd/libqgs_hip_dev.so QGS_HIP_CACHE_DIR=/tmp/kc_tgl; mkdir -p $QGS_HIP_CACHE_DIR
for tag in before after; do
  if [ $tag = before ]; then V="QGS_HIP_TGL_ASM=0"; else V="QGS_HIP_TGL_ASM=1${TGL_EXTRA:+,$TGL_EXTRA}"; fi
  python3 $R/tools/tgls_ab.py $V > $O/warm_$tag.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/${tag}_sq1 -- python3 $R/tools/tgls_ab.py $V > $O/${tag}_sq1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $O/${tag}_sq2 -- python3 $R/tools/tgls_ab.py $V > $O/${tag}_sq2.log 2>&1
done
python3 - $O <<'PY'
import csv, glob, collections, sys
O = sys.argv[1]
out = {}
for tag in ('before', 'after'):
    e = {}
    for d in ('sq1', 'sq2'):
        fs = glob.glob('%s/%s_%s/**/*counter_collection.csv' % (O, tag, d), recursive=True)
        if not fs:
            continue
        acc, dur = collections.defaultdict(list), []
        for r in csv.DictReader(open(fs[0])):
            if 'qgs_spec_tglp' in r['Kernel_Name']:
                e['kernel'] = r['Kernel_Name']
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
                dur.append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-6)
        for k, v in acc.items():
            e[k] = sum(v) / len(v)
        if dur:
            e.setdefault('ms_profiled', sorted(dur)[len(dur) // 2])
    if e.get('SQ_WAVE_CYCLES'):
        e['wait_any_frac'] = e.get('SQ_WAIT_ANY', 0) / e['SQ_WAVE_CYCLES']
    if e.get('GRBM_GUI_ACTIVE') and e.get('ms_profiled'):
        e['grbm_clock_ghz'] = e['GRBM_GUI_ACTIVE'] / 8.0 / (e['ms_profiled'] * 1e6)
    if e.get('SQ_INSTS_VALU') and e.get('SQ_WAVES'):
        e['valu_per_wave_step'] = e['SQ_INSTS_VALU'] / e['SQ_WAVES'] / 10.0
    if e.get('grbm_clock_ghz') and e.get('SQ_INSTS_VALU'):
        e['valu_issue_occupancy'] = e['SQ_INSTS_VALU'] * 4.0 / (e['ms_profiled'] * 1e-3 * e['grbm_clock_ghz'] * 1e9 * 1024)
    out[tag] = e
keys = ['kernel', 'ms_profiled', 'grbm_clock_ghz', 'SQ_WAVES', 'SQ_INSTS_VALU', 'valu_per_wave_step', 'valu_issue_occupancy', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'wait_any_frac',
        'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_INSTS_LDS', 'SQ_LDS_IDX_ACTIVE', 'SQ_LDS_BANK_CONFLICT', 'SQ_INSTS_SMEM', 'SQ_INSTS_SALU', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR']
with open(O + '/../r06_tgls_pmc.txt', 'w') as f:
    f.write('%-24s %24s %24s\n' % ('per launch (tangent pass)', 'compiler-scheduled', 'hand-scheduled'))
    for k in keys:
        f.write('%-24s %24s %24s\n' % (k, *[('%.6g' % out[t][k]) if isinstance(out[t].get(k), float) else str(out[t].get(k, '-')) for t in ('before', 'after')]))
print(open(O + '/../r06_tgls_pmc.txt').read())
PY
