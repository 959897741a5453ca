#!/bin/bash
# GPU box, developer build: batched QR 16 384 x 36x36 (tools/qr_bench.py) with the workgroups of a CU started out of phase
export RK_AB_LIB=qgs_amd/libqgs_hip_dev.so QGS_HIP_CACHE_DIR=/tmp/kc_qr; mkdir -p $QGS_HIP_CACHE_DIR
out=gpurun_out/r06_qr_stagger.txt; : > $out
for v in "QGS_HIP_QR_STAGGER=0" "QGS_HIP_QR_STAGGER=1 QGS_HIP_QR_STAGGER_BIT=8" "QGS_HIP_QR_STAGGER=2 QGS_HIP_QR_STAGGER_BIT=8" "QGS_HIP_QR_STAGGER=3 QGS_HIP_QR_STAGGER_BIT=8" "QGS_HIP_QR_STAGGER=2 QGS_HIP_QR_STAGGER_BIT=0" "QGS_HIP_QR_STAGGER=2 QGS_HIP_QR_STAGGER_BIT=3" "QGS_HIP_QR_STAGGER=2 QGS_HIP_QR_STAGGER_BIT=9" "QGS_HIP_QR_STAGGER=0"; do
  echo "== $v" >> $out
  env $v python tools/qr_bench.py 2>&1 | grep -v amdgpu.ids | head -1 >> $out
done
cat $out
