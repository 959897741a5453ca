#!/bin/bash
# GPU box: counters of the batched QR kernel (tools/qr_prof.py): HBM traffic (FETCH_SIZE / WRITE_SIZE, one pass each, as
# guides/MI355X_MICROARCH.md prescribes), SQ issue / wait, LDS, instruction cache.  RK_AB_LIB=<library> profiles another build.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05qr${1:+_$1}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="python3 $R/tools/qr_prof.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $P > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- $P > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- $P > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/sq1 -o p -- $P > $O/sq1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/sq2 -o p -- $P > $O/sq2.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d $O/sq3 -o p -- $P > $O/sq3.log 2>&1
python3 - $O <<'PY'
import csv, glob, collections, json, sys
O = sys.argv[1]
out = collections.defaultdict(dict)
for d in ('fetch', 'write', 'sq1', 'sq2', 'sq3'):
    fs = glob.glob(O + '/' + d + '/**/*counter_collection.csv', recursive=True)
    if not fs:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r['Kernel_Name'].startswith('qgs_spec_qr') or 'batched_qr' in r['Kernel_Name']:
            acc[(r['Kernel_Name'], r['Counter_Name'])].append((float(r['Counter_Value']), (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-6, int(r['Grid_Size']), int(r['Workgroup_Size'])))
    for (k, c), v in acc.items():
        v = v[2:] if len(v) > 4 else v                      # (the first launches: cold caches, clocks)
        out[k][c] = {'mean_per_launch': sum(x[0] for x in v) / len(v), 'launches': len(v), 'mean_ms_in_this_pass': sum(x[1] for x in v) / len(v),
                     'grid': v[0][2], 'workgroup': v[0][3]}
for k, cs in out.items():
    if 'FETCH_SIZE' in cs and 'WRITE_SIZE' in cs:
        f, w = cs['FETCH_SIZE']['mean_per_launch'], cs['WRITE_SIZE']['mean_per_launch']
        cs['hbm'] = {'fetch_bytes (FETCH_SIZE KiB x 2 x 1024, the gfx950 correction of guides/MI355X_MICROARCH.md)': 2 * f * 1024, 'write_bytes (WRITE_SIZE KiB x 1024)': w * 1024,
                     'hbm_bytes_per_launch': (2 * f + w) * 1024, 'algorithmic_bytes_one_way': 16384 * 36 * 36 * 8}
json.dump(out, open(O + '/qr_pmc_summary.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
f=$(find $O/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/qr_kernel_stats.csv
grep -h "error\|Error" $O/*.log | head -5
