#!/bin/bash
# GPU box: round-3 evidence for profiles/.  rocprofv3 kernel traces of the bench command in steady state: 30 timed + 10 warm-up
# passes, the summary drops the first 10 dispatches of every kernel (cold clocks) and keeps per-dispatch rows, so that the
# averages can be checked against the driver-timed ms_per_step and the bench line's own HIP-event time.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03p
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o bench -- python3 $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extra-configs > $O/bench_headline.json 2> $O/trace.log
rocprofv3 --kernel-trace --output-format csv -d $O/trace_configs -o bench -- python3 $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $O/bench_configs.json 2> $O/trace_configs.log
python3 $R/tools/r03_trace_summary.py $O
# PMC passes (each its own run, --kernel-trace only): HBM traffic of the headline stepper, SQ / scalar-cache counters of the stepper,
# the ndim-228 stepper and the tangent kernel
S="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs"
B="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
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
json.dump(out, open(O + '/r03_pmc_summary.json', 'w'), indent=1)
for k in sorted(out):
    print(k, {c: ('%.4g' % v['mean'], v['n']) for c, v in out[k].items()})
PY
