#!/bin/bash
# GPU box: the whole GPU suite, the A/B of the LDS tangent kernels, and a bench line (builder's copy: profiles/r06_bench.json)
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r06_final_pytest.txt 2>&1; tail -5 gpurun_out/r06_final_pytest.txt
export QGS_HIP_CACHE_DIR=/tmp/kc_tgllds; mkdir -p $QGS_HIP_CACHE_DIR
timeout 1200 python tools/r06_tgllds_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_tgllds_ab.txt; cat gpurun_out/r06_tgllds_ab.txt
unset QGS_HIP_CACHE_DIR
timeout 1200 python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; tail -c 600 gpurun_out/r06_bench.json
