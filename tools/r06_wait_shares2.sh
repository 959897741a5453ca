#!/bin/bash
# GPU box, developer build: second round of the timing experiments on the config-3 stage body (QGS_HIP_LDS_ASM_SKIP bits: 1 barriers, 2 LDS
# waits, 4 vector-memory waits, 8 all loads, 16 spacing s_nop, 32 plain FMA for the DPP form; wrong results, the time and the clock count)
export RK_AB_LIB=qgs_amd/libqgs_hip_dev.so QGS_HIP_CACHE_DIR=/tmp/kc_ws; mkdir -p $QGS_HIP_CACHE_DIR
out=gpurun_out/r06_wait_shares2.txt; : > $out
for v in 0 7 32 39 16 8 15 47 63 39 7 0; do
  echo "== QGS_HIP_LDS_ASM_SKIP=$v" >> $out
  QGS_HIP_LDS_ASM_SKIP=$v timeout 600 python tools/lds228_time.py 2>&1 | grep -v amdgpu.ids | cut -c1-25 >> $out
done
cat $out
