#!/bin/bash
# GPU box, developer build (qgs_amd/libqgs_hip_dev.so: make -C qgs_amd/csrc DEV=1 OUT=../libqgs_hip_dev.so): time of the ndim-228
# LDS-resident stepper (65 536 members x 100 RK4 steps) under the workgroup-shape / factor-cache knobs VERDICT r05 item 1(a) names.
# Each variant is its own process with its own code object in a scratch cache; the shipped shape (16 wavefronts, cache 20) runs
# first, in the middle and last (box drift).  usage: tools/r06_lds228_variants.sh "W CAP [EXTRA=1 ...]" ...
out=gpurun_out/r06_lds228_variants.txt
: > $out
export QGS_HIP_CACHE_DIR=/tmp/kc_variants RK_AB_LIB=qgs_amd/libqgs_hip_dev.so; mkdir -p $QGS_HIP_CACHE_DIR
run() {
  echo "== waves $1 cap $2 ${@:3}" >> $out
  t0=$(date +%s.%N)
  env QGS_HIP_LDS_WAVES=$1 QGS_HIP_LDS_CAP=$2 "${@:3}" timeout 900 python tools/lds228_time.py 2>&1 | grep -v amdgpu.ids >> $out
  env QGS_HIP_LDS_WAVES=$1 QGS_HIP_LDS_CAP=$2 "${@:3}" timeout 900 python tools/lds228_time.py 2>&1 | grep -v amdgpu.ids >> $out
  echo "   (both processes incl. compilation: $(python3 -c "import time;print('%.0f s' % ($(date +%s.%N)-$t0))"))" >> $out
}
n=$#; i=0
run 16 20
for v in "$@"; do
  run $v
  i=$((i+1)); [ $i -eq $((n/2)) ] && run 16 20
done
run 16 20
cat $out
