#!/bin/bash
# GPU box: PMC passes over the LDS-resident tangent kernel of MAOOAM 6x6 (1 024 members x 228 vectors x 10 sub-steps, 3 launches per
# pass), compiler-scheduled (QGS_HIP_LDS_TGL_ASM=0, "before") and hand-scheduled ("after"): waits, VALU, LDS, scalar cache, clock and
# HBM bytes (FETCH_SIZE / WRITE_SIZE in their own passes, corrected as guides/MI355X_MICROARCH.md prescribes).
# Output: gpurun_out/r06_tgllds_pmc.json + .txt   (the f-row tangent kernel)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_tgllds_pmc
rm -rf $O; mkdir -p $O
export R06_PROF=1 QGS_HIP_CACHE_DIR=/tmp/kc_pmc; mkdir -p $QGS_HIP_CACHE_DIR
P="python3 $R/tools/r06_tgllds_ab.py --child pmc"
for tag in before after; do
  if [ $tag = before ]; then export QGS_HIP_LDS_TGL_ASM=0; else export QGS_HIP_LDS_TGL_ASM=1; fi
  $P > $O/warm_$tag.log 2>&1       # (compilation outside the profiled runs)
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/${tag}_sq1 -- $P > $O/${tag}_sq1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/${tag}_sq2 -- $P > $O/${tag}_sq2.log 2>&1
  rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/${tag}_sq3 -- $P > $O/${tag}_sq3.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${tag}_fetch -- $P > $O/${tag}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${tag}_write -- $P > $O/${tag}_write.log 2>&1
done
python3 - $O <<'PY'
import csv, glob, collections, json, sys
O = sys.argv[1]
out = {}
for tag in ('before', 'after'):
    e = {}
    for d in ('sq1', 'sq2', 'sq3', 'fetch', 'write'):
        fs = glob.glob('%s/%s_%s/**/*counter_collection.csv' % (O, tag, d), recursive=True)
        if not fs:
            continue
        acc, dur = collections.defaultdict(list), []
        for r in csv.DictReader(open(fs[0])):
            if 'tgllds' in r['Kernel_Name']:
                e['kernel'] = r['Kernel_Name']
                e['grid'] = int(r['Grid_Size'])
                e['vgprs'] = r.get('VGPR_Count') or r.get('Arch_VGPR_Count')
                e['scratch'] = r.get('Scratch_Size') or r.get('Private_Segment_Size')
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
                dur.append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-6)
        for k, v in acc.items():
            v2 = v[1:] if len(v) > 1 else v                      # (first launch: cold)
            e[k] = sum(v2) / len(v2)
        if dur:
            e.setdefault('ms_profiled', sorted(dur)[len(dur) // 2])
    if 'FETCH_SIZE' in e and 'WRITE_SIZE' in e:
        e['hbm_bytes_per_launch'] = int((2 * e['FETCH_SIZE'] + e['WRITE_SIZE']) * 1024)          # FETCH doubled: the gfx950 correction of the guide
        e['traffic_over_algorithmic'] = e['hbm_bytes_per_launch'] / (2.0 * 8 * (228 + 228 * 228) * 1024 * 10)
    if 'GRBM_GUI_ACTIVE' in e and e.get('ms_profiled'):
        e['grbm_clock_ghz'] = e['GRBM_GUI_ACTIVE'] / 8.0 / (e['ms_profiled'] * 1e6)
    if e.get('SQ_WAVE_CYCLES'):
        e['wait_any_frac'] = e.get('SQ_WAIT_ANY', 0) / e['SQ_WAVE_CYCLES']
        e['wait_inst_any_frac'] = e.get('SQ_WAIT_INST_ANY', 0) / e['SQ_WAVE_CYCLES']
    if e.get('SQC_DCACHE_REQ'):
        e['scalar_dcache_miss_frac'] = e.get('SQC_DCACHE_MISSES', 0) / e['SQC_DCACHE_REQ']
    if e.get('grbm_clock_ghz') and e.get('SQ_INSTS_VALU'):
        e['valu_issue_occupancy'] = e['SQ_INSTS_VALU'] * 4.0 / (e['ms_profiled'] * 1e-3 * e['grbm_clock_ghz'] * 1e9 * 1024)
    out[tag] = e
json.dump(out, open(O + '/../r06_tgllds_pmc.json', 'w'), indent=1)
keys = ['kernel', 'grid', 'ms_profiled', 'grbm_clock_ghz', 'SQ_WAVES', 'SQ_INSTS_VALU', 'valu_issue_occupancy', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'wait_any_frac', 'SQ_WAIT_INST_ANY',
        'wait_inst_any_frac', 'SQ_WAIT_INST_LDS', 'SQ_INSTS_LDS', 'SQ_LDS_IDX_ACTIVE', 'SQ_LDS_BANK_CONFLICT', 'SQ_INSTS_SMEM', 'SQ_INSTS_SALU', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR',
        'SQC_DCACHE_REQ', 'SQC_DCACHE_MISSES', 'scalar_dcache_miss_frac', 'FETCH_SIZE', 'WRITE_SIZE', 'hbm_bytes_per_launch', 'traffic_over_algorithmic']
with open(O + '/../r06_tgllds_pmc.txt', 'w') as f:
    f.write('%-28s %22s %22s\n' % ('per launch (1 024 x 228 x 10)', 'before (QGS_HIP_LDS_TGL_ASM=0)', 'after (hand-scheduled)'))
    for k in keys:
        f.write('%-28s %22s %22s\n' % (k, *[('%.6g' % out[t][k]) if isinstance(out[t].get(k), float) else str(out[t].get(k, '-')) for t in ('before', 'after')]))
print(open(O + '/../r06_tgllds_pmc.txt').read())
PY
