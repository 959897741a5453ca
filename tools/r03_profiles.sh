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
