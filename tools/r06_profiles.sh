#!/bin/bash
# GPU box: round-6 evidence for profiles/.  Kernel traces of the bench command in steady state (as rounds 3-4), the two HBM PMC
# passes (FETCH_SIZE / WRITE_SIZE, each its own run, --kernel-trace only, guides/MI355X_MICROARCH.md), and two SQ passes -- all
# summarised PER (KERNEL, GRID) CLASS: one row for qgs_spec_rk_s4 at 65 536 x 1000 (grid 65536), another for its 1 048 576-member
# launch, another for the 100-step launches, each with its counters per launch (the issue arithmetic of the
# headline -- fp64 instructions x 4 cycles over wave cycles at the measured clock -- can be redone from r06_pmc_by_class.csv alone).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06p
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o bench -- python3 $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extra-configs --no-cold-start > $O/bench_headline.json 2> $O/trace.log
rocprofv3 --kernel-trace --output-format csv -d $O/trace_configs -o bench -- python3 $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-cold-start > $O/bench_configs.json 2> $O/trace_configs.log
python3 $R/tools/r03_trace_summary.py $O r06 > $O/r06_trace_summary.txt 2>&1
B="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cold-start"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- $B > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/sq1 -o p -- $B > $O/sq1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_WAVES SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/sq2 -o p -- $B > $O/sq2.log 2>&1
python3 - $O <<'PY'
import csv, glob, collections, json, re, sys
O = sys.argv[1]
def rows(d):
    fs = glob.glob(O + '/' + d + '/**/*counter_collection.csv', recursive=True)
    return list(csv.DictReader(open(fs[0]))) if fs else []
# class = (kernel, grid, duration class: a class ends where the sorted durations of one (kernel, grid) jump by more than 1.6x)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('fetch', 'write', 'sq1', 'sq2'):
    per = collections.defaultdict(list)
    for r in rows(d):
        k = re.sub(r'^(\w+::)+', '', r['Kernel_Name'].split('(')[0])          # (generic kernels: without namespace and argument list)
        if k.startswith('qgs_') or 'batched_qr' in k:
            per[(k, int(r['Grid_Size']), r['Counter_Name'])].append((float(r['Counter_Value']), (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-6))
    for (k, g, c), v in per.items():
        order = sorted(x[1] for x in v)
        edges, first = [], order[0]
        for x in order:
            if x > 1.6 * first:
                edges.append(x); first = x
        for val, dur in v:
            cls = sum(1 for e in edges if dur >= e)
            acc[(k, g, cls)][c].append((val, dur))
out, table = {}, []
for (k, g, cls), cs in sorted(acc.items()):
    name = '%s@grid%d%s' % (k, g, ('/class%d' % cls) if cls else '')
    e = {'kernel': k, 'grid': g, 'duration_class': cls}
    for c, v in cs.items():
        v2 = v[1:] if len(v) > 2 else v                      # (the first launch of a class: cold caches)
        e[c] = sum(x[0] for x in v2) / len(v2)
        e[c + '_launches'] = len(v2)
        e.setdefault('mean_ms', sum(x[1] for x in v2) / len(v2))
    if 'FETCH_SIZE' in e and 'WRITE_SIZE' in e:
        e['hbm_bytes_per_launch'] = int((2 * e['FETCH_SIZE'] + e['WRITE_SIZE']) * 1024)     # FETCH doubled: the gfx950 correction of the guide
    if 'GRBM_GUI_ACTIVE' in e and e.get('mean_ms'):
        e['grbm_clock_ghz'] = e['GRBM_GUI_ACTIVE'] / 8.0 / (e['mean_ms'] * 1e6)                # the counter sums the 8 XCDs
    out[name] = e
    table.append(e)
json.dump(out, open(O + '/r06_pmc_by_class.json', 'w'), indent=1)
cols = ['kernel', 'grid', 'duration_class', 'mean_ms', 'SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_ACTIVE_INST_VALU', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY',
        'SQ_INSTS_SALU', 'SQ_INSTS_SMEM', 'SQ_INSTS_LDS', 'SQ_LDS_IDX_ACTIVE', 'SQ_LDS_BANK_CONFLICT', 'SQC_DCACHE_REQ', 'SQC_DCACHE_MISSES', 'SQ_INSTS_VMEM_RD',
        'GRBM_GUI_ACTIVE', 'grbm_clock_ghz', 'FETCH_SIZE', 'WRITE_SIZE', 'hbm_bytes_per_launch']
with open(O + '/r06_pmc_by_class.csv', 'w') as f:
    w = csv.writer(f)
    w.writerow(cols)
    for e in table:
        w.writerow([e.get(c, '') for c in cols])
# the traffic table bench.py looks up by (kernel, grid, duration class): one entry per class, no entry under a bare kernel name
traffic = {}
for name, e in out.items():
    if 'hbm_bytes_per_launch' not in e:
        continue
    traffic[name] = {'fetch_size_kib_raw': e['FETCH_SIZE'], 'write_size_kib_raw': e['WRITE_SIZE'], 'hbm_bytes_per_launch': e['hbm_bytes_per_launch'],
                     'mean_ms': e['mean_ms'], 'grid': e['grid'], 'round': 6}
json.dump(traffic, open(O + '/r06_hbm_traffic.json', 'w'), indent=1)
for name in sorted(out):
    e = out[name]
    print('%-52s %9.4f ms  VALU %12.0f  WAVE_CYC %14.0f  WAIT_ANY %13.0f  clock %5.2f GHz  HBM %s' % (name, e.get('mean_ms', 0), e.get('SQ_INSTS_VALU', 0), e.get('SQ_WAVE_CYCLES', 0), e.get('SQ_WAIT_ANY', 0), e.get('grbm_clock_ghz', 0), e.get('hbm_bytes_per_launch', '-')))
PY
for c in fetch write; do f=$(find $O/$c -name '*counter_collection.csv' | head -1); [ -n "$f" ] && (head -1 $f; grep qgs_spec_rk_s4 $f | head -40) > $O/r06_pmc_${c}_size.csv; done
# what goes to profiles/: r06_bench_kernel_stats.csv, r06_bench_dispatches.csv, r06_trace_summary.txt, r06_pmc_by_class.{csv,json} (-> also profiles/pmc_by_class.json),
# r06_hbm_traffic.json (-> also profiles/hbm_traffic.json), r06_pmc_{fetch,write}_size.csv, bench_configs.json (-> r06_bench.json)
ls $O | head -30
