#!/bin/bash
# GPU box: test suite, smoke, bench line, rocprofv3 kernel stats of the bench command.
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02/pytest_gpu.log
tail -15 gpurun_out/r02/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02/smoke.log 2>&1; tail -3 gpurun_out/r02/smoke.log
python bench.py > gpurun_out/r02/bench.json 2> gpurun_out/r02/bench.err; tail -c 6000 gpurun_out/r02/bench.json; tail -5 gpurun_out/r02/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r02/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/r02/prof_bench -name "*stats*" | head; 
