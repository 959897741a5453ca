#!/bin/bash
# GPU box: N back-to-back runs of the GPU test suite, each a fresh pytest process, under one long-lived shell -- the way the GPU
# write faults of round 4 were found (about one run in ten to twenty died with "Memory access fault by GPU ... Write access to
# a read-only page", profiles/r04_gpu_suite_runs.txt).  One line per run: exit code, pytest's summary, seconds, and any fault
# text found in the run's output.     tools/soak_gpu_suite.sh [runs, default 30] [pytest selection, default: tests -m gpu]
N=${1:-30}
shift
SEL=${@:-tests -m gpu}          # (a selection replaces the default: name test files, or "tests -m gpu -k ...")
out=gpurun_out/soak_gpu_suite.txt
mkdir -p gpurun_out
echo "# $(date -u +%FT%TZ) $N runs of: python -m pytest -x -q $SEL   (tree $(git rev-parse --short HEAD 2>/dev/null || echo snapshot))" > $out
fail=0
for i in $(seq 1 $N); do
    t0=$(date +%s)
    python -m pytest -x -q $SEL > /tmp/soak_run.log 2>&1
    rc=$?
    t1=$(date +%s)
    fault=$(grep -a -m1 -o "Memory access fault[^\"]*" /tmp/soak_run.log)
    echo "run $i rc=$rc $(tail -1 /tmp/soak_run.log) [$((t1 - t0)) s] ${fault}" >> $out
    if [ $rc -ne 0 ]; then fail=$((fail + 1)); grep -a -n -m1 -A60 "Fatal Python error\|Traceback\|^E  " /tmp/soak_run.log >> $out; echo "..." >> $out; tail -30 /tmp/soak_run.log >> $out; cp /tmp/soak_run.log gpurun_out/soak_failed_run_$i.log; fi
done
echo "# failed runs: $fail of $N" >> $out
tail -3 $out
