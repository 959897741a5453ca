#!/bin/bash
# GPU box, developer build: config 4 (16 384 members x 36 tangent vectors x 10 sub-steps, 100 calls back to back) with the
# compiler-scheduled pair kernel (QGS_HIP_TGL_ASM=0) and the hand-scheduled one, variants in one process (tools/tgls_ab.py).
export RK_AB_LIB=qgs_amd/libqgs_hip_dev.so QGS_HIP_CACHE_DIR=/tmp/kc_tgl; mkdir -p $QGS_HIP_CACHE_DIR
python tools/tgls_ab.py "$@" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_tgls_ab.txt
