"""Summarise rocprofv3 --kernel-trace CSVs of bench.py: per kernel the dispatches after the warm-up ones, their mean / min / max,
next to the bench line's own numbers (ms_per_step, roofline.kernel_ms and the `configs` entries).  Writes
<out>/r03_bench_kernel_stats.csv (steady-state summary), <out>/r03_bench_dispatches.csv (per-dispatch rows) and prints the
cross-check."""
import csv
import glob
import json
import sys

O = sys.argv[1]
PREFIX = sys.argv[2] if len(sys.argv) > 2 else 'r03'          # round tag of the two output files


def dispatches(sub):
    fs = glob.glob(O + '/' + sub + '/**/*kernel_trace.csv', recursive=True)
    rows = []
    for f in fs:
        for r in csv.DictReader(open(f)):
            rows.append((r['Kernel_Name'] + ' grid=' + r.get('Grid_Size', '?'), int(r['Start_Timestamp']), int(r['End_Timestamp'])))
    rows.sort(key=lambda q: q[1])
    return rows


def line(path):
    for ln in open(path):
        if ln.startswith('{'):
            return json.loads(ln)
    return {}


out_rows, disp_rows = [], []
for sub, jpath, skip in (('trace', 'bench_headline.json', 10), ('trace_configs', 'bench_configs.json', None)):
    bench = line(O + '/' + jpath)
    by0, by = {}, {}
    for k, s, e in dispatches(sub):
        by0.setdefault(k, []).append((e - s) * 1e-6)
    # one kernel name at one grid size may serve launches of different lengths (1000-step passes and 100-step `configs` entries):
    # split such a list into duration classes (a class ends where the sorted durations jump by more than 1.6x)
    for k, v in by0.items():
        order = sorted(v)
        edges, first = [], order[0]
        for x in order:
            if x > 1.6 * first:
                edges.append(x)
                first = x
        for x in v:
            c = sum(1 for e in edges if x >= e)
            by.setdefault(k + (' class%d' % c if edges else ''), []).append(x)
    for k, v in sorted(by.items()):
        if not k.startswith('qgs_') and 'kernel' not in k:
            continue
        # warm-up dispatches: the first `skip` of the headline run; for the configs run every entry's own first launch(es)
        drop = skip if skip is not None else max(1, len(v) // 10)
        steady = v[drop:] if len(v) > drop else v
        out_rows.append({'run': sub, 'kernel': k, 'dispatches': len(v), 'dropped_warmup': len(v) - len(steady),
                         'mean_ms': sum(steady) / len(steady), 'min_ms': min(steady), 'max_ms': max(steady)})
        for i, ms in enumerate(v):
            disp_rows.append({'run': sub, 'kernel': k, 'index': i, 'ms': ms, 'counted': int(i >= drop or len(v) <= drop)})
    if bench:
        print(sub, 'bench line: ms_per_step %.4f  roofline.kernel_ms %.4f' % (bench['ms_per_step'], bench['roofline']['kernel_ms']))
        for k, v in bench.get('configs', {}).items():
            if isinstance(v, dict):
                print('   ', k, v.get('kernel'), v.get('ms', v.get('ms_per_call')))
for name, rows in ((PREFIX + '_bench_kernel_stats.csv', out_rows), (PREFIX + '_bench_dispatches.csv', disp_rows)):
    with open(O + '/' + name, 'w', newline='') as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
for r in out_rows:
    if r['dispatches'] >= 3:
        print('%-14s %-52s n=%4d (-%d)  mean %.4f  min %.4f  max %.4f ms' % (r['run'], r['kernel'][:52], r['dispatches'], r['dropped_warmup'], r['mean_ms'], r['min_ms'], r['max_ms']))
