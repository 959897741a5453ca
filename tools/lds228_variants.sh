#!/bin/bash
# Developer script: timing of qgs_spec_rklds16 (MAOOAM 6x6, 65 536 members x 100 steps) under generator knobs.
# Each variant compiles its own code object into a scratch cache (20-40 s each).  usage: lds228_variants.sh "VAR=1 VAR2=3" "..." ...
# NOLSO=1 in a variant: compile without the SI load/store optimizer (keeps ds_read_b64 unpaired).
out=gpurun_out/lds228_variants.txt
: > $out
export QGS_HIP_CACHE_DIR=/tmp/kc_variants; mkdir -p $QGS_HIP_CACHE_DIR
for v in "$@"; do
  echo "== $v" >> $out
  if [[ "$v" == *NOLSO=1* ]]; then export QGS_HIP_EXTRA_FLAGS="-Xclang -target-feature -Xclang -load-store-opt"; else unset QGS_HIP_EXTRA_FLAGS; fi
  env $v timeout 900 python tools/lds228_time.py 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
