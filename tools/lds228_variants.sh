#!/bin/bash
# Developer script: timing of qgs_spec_rklds16 (MAOOAM 6x6, 65 536 members x 100 steps) under generator knobs, each variant with its
# own code object in a scratch cache.  usage: lds228_variants.sh "VAR=1 VAR2=3" "..." ...
out=gpurun_out/lds228_variants.txt
: > $out
export QGS_HIP_CACHE_DIR=/tmp/kc_variants; mkdir -p $QGS_HIP_CACHE_DIR
for v in "$@"; do
  echo "== $v" >> $out
  env $v timeout 900 python tools/lds228_time.py 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
