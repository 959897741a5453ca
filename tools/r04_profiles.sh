#!/bin/bash
# GPU box: round-4 evidence for profiles/.  Same method as round 3 (tools/r03_profiles.sh): rocprofv3 kernel traces of the bench
# command in steady state (30 timed + 10 warm-up passes; the summary drops the first 10 dispatches of every kernel and keeps
# per-dispatch rows), then the two HBM PMC passes (FETCH_SIZE / WRITE_SIZE, each its own run, --kernel-trace only) over the whole
# bench command -> per-launch traffic of every kernel of the line, and one SQ / scalar-cache pass pair.
# The kernels are the value-free ones of round 4 (coefficient tables filled at module load).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04p
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o bench -- python3 $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extra-configs --no-cold-start > $O/bench_headline.json 2> $O/trace.log
rocprofv3 --kernel-trace --output-format csv -d $O/trace_configs -o bench -- python3 $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-cold-start > $O/bench_configs.json 2> $O/trace_configs.log
python3 $R/tools/r03_trace_summary.py $O r04 > $O/trace_summary.txt 2>&1
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cold-start"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- $B > $O/write.log 2>&1
B5="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cold-start"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/sq_bench -o p -- $B5 > $O/sq_bench.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/sq_bench2 -o p -- $B5 > $O/sq_bench2.log 2>&1
python3 - $O <<'PY'
import csv, glob, collections, json, sys
O = sys.argv[1]
def rows(d):
    fs = glob.glob(O + '/' + d + '/**/*counter_collection.csv', recursive=True)
    return list(csv.DictReader(open(fs[0]))) if fs else []
# ---- HBM traffic per launch (FETCH doubled as guides/MI355X_MICROARCH.md prescribes for gfx950; counters in KiB) ----
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('fetch', 'write'):
    for r in rows(d):
        k = r['Kernel_Name']
        if k.startswith('qgs_'):
            dur = (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-6
            acc[k][r['Counter_Name']].append((float(r['Counter_Value']), dur, int(r['Grid_Size'])))
traffic = {}
for k, cs in acc.items():
    f, w = cs.get('FETCH_SIZE', []), cs.get('WRITE_SIZE', [])
    # one kernel name may serve launches of different sizes: classes by grid size, and within a grid by duration (1.6x jumps)
    grids = sorted(set(x[2] for x in f))
    for g in grids:
        fg, wg = [x for x in f if x[2] == g], [x for x in w if x[2] == g]
        if not fg or not wg: continue
        mx = max(x[1] for x in fg)
        fg = [x for x in fg if x[1] > mx / 1.6]
        mxw = max(x[1] for x in wg)
        wg = [x for x in wg if x[1] > mxw / 1.6]
        fk, wk = sum(x[0] for x in fg) / len(fg), sum(x[0] for x in wg) / len(wg)
        name = k if len(grids) == 1 else '%s@grid%d' % (k, g)
        traffic[name] = {'fetch_size_kib_raw': fk, 'write_size_kib_raw': wk, 'hbm_bytes_per_launch': int((2 * fk + wk) * 1024),
                         'launches': [len(fg), len(wg)], 'mean_ms': sum(x[1] for x in fg) / len(fg), 'grid': g, 'round': 4}
json.dump(traffic, open(O + '/r04_hbm_traffic.json', 'w'), indent=1)
# ---- SQ / SQC summary ----
out = {}
for d in ('sq_bench', 'sq_bench2'):
    a2 = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows(d):
        k = r['Kernel_Name']
        if k.startswith('qgs_'):
            a2[k][r['Counter_Name']].append((float(r['Counter_Value']), int(r['Grid_Size']), (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-6))
    for k, cs in a2.items():
        for c, v in cs.items():
            out.setdefault(k, {})[c] = {'mean': sum(x[0] for x in v) / len(v), 'n': len(v), 'grid': sorted(set(x[1] for x in v)), 'mean_ms': sum(x[2] for x in v) / len(v)}
json.dump(out, open(O + '/r04_pmc_summary.json', 'w'), indent=1)
for k in sorted(traffic): print(k, traffic[k])
PY
# rows of the stepper kernel of the two HBM passes, as in earlier rounds
for c in fetch write; do f=$(find $O/$c -name '*counter_collection.csv' | head -1); [ -n "$f" ] && (head -1 $f; grep qgs_spec_rk_s4 $f | head -40) > $O/r04_pmc_${c}_size.csv; done
cp $O/trace_summary.txt $O/r04_trace_summary.txt
ls -la $O
