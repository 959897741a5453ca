#!/bin/bash
# GPU box: HBM traffic (TCC FETCH_SIZE / WRITE_SIZE, separate passes, --kernel-trace only) of every kernel the bench line's `configs`
# entries launch; summarised into gpurun_out/r03t/hbm_traffic_configs.json (per launch, FETCH doubled as guides/MI355X_MICROARCH.md
# prescribes for gfx950, counters in KiB).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03t
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- $B > $O/write.log 2>&1
python3 - $O <<'PY'
import csv, glob, collections, json, sys
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('fetch', 'write'):
    fs = glob.glob(O + '/' + d + '/**/*counter_collection.csv', recursive=True)
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name']
        if k.startswith('qgs_'):
            dur = (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-6
            acc[k][r['Counter_Name']].append((float(r['Counter_Value']), dur, int(r['Grid_Size'])))
out = {}
for k, cs in acc.items():
    # one kernel name may serve launches of different sizes: keep the class with the longest launches (the `configs` entry / headline)
    f, w = cs.get('FETCH_SIZE', []), cs.get('WRITE_SIZE', [])
    def top(v):
        if not v: return []
        mx = max(x[1] for x in v)
        return [x for x in v if x[1] > mx / 1.6]
    f, w = top(f), top(w)
    if not f or not w: continue
    fk, wk = sum(x[0] for x in f) / len(f), sum(x[0] for x in w) / len(w)
    out[k] = {'fetch_size_kib_raw': fk, 'write_size_kib_raw': wk, 'hbm_bytes_per_launch': int((2 * fk + wk) * 1024),
              'launches': [len(f), len(w)], 'mean_ms': sum(x[1] for x in f) / len(f), 'grid': sorted(set(x[2] for x in f))}
json.dump(out, open(O + '/hbm_traffic_configs.json', 'w'), indent=1)
for k, v in sorted(out.items()):
    print(k, v)
PY
