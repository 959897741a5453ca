#!/bin/bash
# Developer script: timing of qgs_spec_rklds16 (MAOOAM 6x6, 65 536 members x 100 steps) under generator knobs.
# Each variant compiles its own code object into a scratch cache (20-40 s each).
out=gpurun_out/lds228_variants.txt
: > $out
export QGS_HIP_CACHE_DIR=/tmp/kc_variants; mkdir -p $QGS_HIP_CACHE_DIR
run() { echo "== $*" >> $out; env "$@" timeout 600 python tools/lds228_time.py >> $out 2>&1; }
run A=0
run QGS_HIP_LDS_DEBUG=2
run QGS_HIP_LDS_DEBUG=4
run QGS_HIP_LDS_DEBUG=1
run QGS_HIP_LDS_DEBUG=5
run QGS_HIP_KTAB_GROUP=8
run QGS_HIP_LDS_DPP=1
run QGS_HIP_LDS_DPP=1 QGS_HIP_LDS_CAP=18
run QGS_HIP_LDS_WAVES=8 QGS_HIP_LDS_CAP=40
run QGS_HIP_LDS_WAVES=12
run QGS_HIP_LDS_CAP=16
cat $out
