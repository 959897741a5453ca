#!/bin/bash
# GPU box: parity of the hand-scheduled LDS tangent kernels (tests), then their time against the compiler-scheduled ones
# (tools/r06_tgllds_ab.py; developer build for the generator knobs).   usage: tools/r06_tgllds_ab.sh [variant ...]
out=gpurun_out/r06_tgllds_ab.txt
: > $out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "hand_scheduled_lds_tangent" 2>&1 | tail -15 >> $out
export QGS_HIP_CACHE_DIR=/tmp/kc_tgllds RK_AB_LIB=qgs_amd/libqgs_hip_dev.so; mkdir -p $QGS_HIP_CACHE_DIR
timeout 2400 python tools/r06_tgllds_ab.py "$@" 2>&1 | grep -v amdgpu.ids >> $out
cat $out
