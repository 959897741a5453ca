#!/bin/bash
# GPU box, developer build: config 3 with the coefficients as scalar operands (QGS_HIP_LDS_ASM_COEF=0) instead of the DPP broadcast:
# time, shader clock (in-kernel probe) and difference from the generic kernel; QGS_HIP_LDS_ASM_TOUCH = chunks the scalar cache is warmed ahead
export RK_AB_LIB=qgs_amd/libqgs_hip_dev.so QGS_HIP_CACHE_DIR=/tmp/kc_ws; mkdir -p $QGS_HIP_CACHE_DIR
out=gpurun_out/r06_coef_clock.txt; : > $out
for v in "QGS_HIP_LDS_ASM_COEF=1" "QGS_HIP_LDS_ASM_COEF=0 QGS_HIP_LDS_ASM_TOUCH=-1" "QGS_HIP_LDS_ASM_COEF=0 QGS_HIP_LDS_ASM_TOUCH=0" "QGS_HIP_LDS_ASM_COEF=0 QGS_HIP_LDS_ASM_TOUCH=2" "QGS_HIP_LDS_ASM_COEF=0 QGS_HIP_LDS_ASM_TOUCH=4" "QGS_HIP_LDS_ASM_COEF=0 QGS_HIP_LDS_ASM_TOUCH=8" "QGS_HIP_LDS_ASM_COEF=0 QGS_HIP_LDS_ASM_TOUCH=16" "QGS_HIP_LDS_ASM_COEF=0 QGS_HIP_LDS_ASM_TOUCH=4 QGS_HIP_LDS_ASM_CHUNK=8" "QGS_HIP_LDS_ASM_COEF=1"; do
  echo "== $v" >> $out
  env $v timeout 600 python tools/lds228_time.py 2>&1 | grep -v amdgpu.ids | sed 's/{.*}//' >> $out
done
cat $out
