#!/bin/bash
# GPU box, developer build: the generator knobs of the hand-scheduled stage body on the LDS tangent kernels of MAOOAM 6x6
# (tools/r06_tgllds_ab.py: 1 024 members x 228 vectors x 10 sub-steps; every variant a process of its own, the default first and last)
export RK_AB_LIB=qgs_amd/libqgs_hip_dev.so QGS_HIP_CACHE_DIR=/tmp/kc_tgk; mkdir -p $QGS_HIP_CACHE_DIR
out=gpurun_out/r06_tgllds_knobs.txt
timeout 3000 python tools/r06_tgllds_ab.py default: "$@" default: 2>&1 | grep -v amdgpu.ids > $out
cat $out
