#!/bin/bash
# GPU box: round-2 evidence for profiles/ -- rocprofv3 kernel stats of the bench command, HBM traffic of the stepper (separate
# FETCH_SIZE / WRITE_SIZE passes), SQ passes on qgs_spec_rk_s4 / qgs_spec_tgl_s4 / qgs_spec_rklds16.  PMC passes use --kernel-trace only.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02p
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
# headline kernel alone (the 100-step launches of the `configs` entries would mix into the qgs_spec_rk_s4 average) ...
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B --no-extra-configs > $O/stats.log 2>&1
# ... and the whole bench line with its `configs` entries
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_configs -o bench -- $B > $O/stats_configs.log 2>&1
S="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- $S > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- $S > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/sq_bench -o p -- $B > $O/sq_bench.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/sq_bench2 -o p -- $B > $O/sq_bench2.log 2>&1
python3 - $O <<'PY'
import csv, glob, collections, json, sys
O = sys.argv[1]
def rows(d):
    fs = glob.glob(O + '/' + d + '/**/*counter_collection.csv', recursive=True)
    return list(csv.DictReader(open(fs[0]))) if fs else []
out = {}
for d in ('fetch', 'write', 'sq_bench', 'sq_bench2'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows(d):
        k = r['Kernel_Name']
        if k.startswith('qgs_'):
            acc[k][r['Counter_Name']].append((float(r['Counter_Value']), int(r['Grid_Size']), (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-6))
    for k, cs in acc.items():
        for c, v in cs.items():
            out.setdefault(k, {})[c] = {'mean': sum(x[0] for x in v) / len(v), 'n': len(v), 'grid': sorted(set(x[1] for x in v)), 'mean_ms': sum(x[2] for x in v) / len(v)}
json.dump(out, open(O + '/pmc_summary.json', 'w'), indent=1)
for k in sorted(out):
    print(k, {c: ('%.4g' % v['mean'], v['n']) for c, v in out[k].items()})
PY
cd $R; find $O -name "*kernel_stats.csv" | head -3
