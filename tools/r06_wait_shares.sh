#!/bin/bash
# GPU box, developer build: where the waits of the hand-scheduled LDS kernels come from -- the stage body timed without its barriers,
# without its LDS waits, without its vector-memory waits (QGS_HIP_LDS_ASM_SKIP = 1 / 2 / 4 and sums; the results of these runs are wrong,
# only the time counts).  Config 3 (tools/lds228_time.py) and the tangent kernel (tools/r06_tgllds_ab.py).
export RK_AB_LIB=qgs_amd/libqgs_hip_dev.so QGS_HIP_CACHE_DIR=/tmp/kc_ws; mkdir -p $QGS_HIP_CACHE_DIR
out=gpurun_out/r06_wait_shares.txt; : > $out
for v in 0 1 2 4 3 6 7 0; do
  echo "== QGS_HIP_LDS_ASM_SKIP=$v" >> $out
  QGS_HIP_LDS_ASM_SKIP=$v timeout 600 python tools/lds228_time.py 2>&1 | grep -v amdgpu.ids | cut -c1-60 >> $out
done
timeout 2400 python tools/r06_tgllds_ab.py skip0: skip1:QGS_HIP_LDS_ASM_SKIP=1 skip2:QGS_HIP_LDS_ASM_SKIP=2 skip4:QGS_HIP_LDS_ASM_SKIP=4 skip7:QGS_HIP_LDS_ASM_SKIP=7 skip0: 2>&1 | grep -v amdgpu.ids | grep -v 'diff' | cut -c1-150 >> $out
cat $out
