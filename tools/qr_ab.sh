#!/bin/bash
# Developer script (GPU box): batched QR, the previous library (tools/build_prev_lib.sh) against the working tree's, then the
# layout variants of the generator in a developer build (make -C qgs_amd/csrc DEV=1 OUT=../libqgs_hip_dev.so).
out=gpurun_out/qr_ab.txt
: > $out
python -m pytest tests/test_gpu_lyapunov.py -x -q -k "batched_qr" 2>&1 | tail -3 >> $out
echo "== previous library" >> $out
RK_AB_LIB=qgs_amd/libqgs_hip_old.so python tools/qr_bench.py >> $out 2>&1
echo "== working tree" >> $out
python tools/qr_bench.py >> $out 2>&1
for v in "$@"; do
    echo "== dev build, $v" >> $out
    env $v RK_AB_LIB=qgs_amd/libqgs_hip_dev.so python tools/qr_bench.py >> $out 2>&1
done
cat $out
