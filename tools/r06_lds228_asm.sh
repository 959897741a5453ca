#!/bin/bash
# GPU box, developer build: time of the hand-scheduled ndim-228 LDS stepper (qgs_spec_rkldsa<W>, 65 536 members x 100 RK4 steps)
# under its generator knobs, the compiler-scheduled kernel (QGS_HIP_LDS_ASM=0) first and last.
#   usage: tools/r06_lds228_asm.sh "QGS_HIP_LDS_ASM_CAP=20 QGS_HIP_LDS_ASM_LANES=3" ...
out=gpurun_out/r06_lds228_asm.txt
: > $out
export QGS_HIP_CACHE_DIR=/tmp/kc_variants RK_AB_LIB=qgs_amd/libqgs_hip_dev.so; mkdir -p $QGS_HIP_CACHE_DIR
run() {
  echo "== $@" >> $out
  env "$@" timeout 900 python tools/lds228_time.py 2>&1 | grep -v amdgpu.ids >> $out
  env "$@" timeout 900 python tools/lds228_time.py 2>&1 | grep -v amdgpu.ids >> $out
}
run QGS_HIP_LDS_ASM=0
for v in "$@"; do run $v; done
run QGS_HIP_LDS_ASM=0
cat $out
