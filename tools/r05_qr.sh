#!/bin/bash
# Developer script (GPU box): phases of the batched QR kernel variants + the QR and pageable-result tests
out=gpurun_out/r05_qr_phases.txt
: > $out
for b in tools/ubench/qr_phases_*; do
    [ -x "$b" ] || continue
    echo "== $b" >> $out
    $b 16384 >> $out 2>&1
done
cat $out
