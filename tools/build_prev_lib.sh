#!/bin/bash
# Developer script: build the library of the last commit next to the working tree's (qgs_amd/libqgs_hip_old.so), for
# old-vs-new comparisons on one GPU box: RK_AB_LIB=qgs_amd/libqgs_hip_old.so python tools/rk_ab.py A=0
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/qgs_prev && git -C "$root" worktree add -f /tmp/qgs_prev HEAD > /dev/null 2>&1
make -C /tmp/qgs_prev/qgs_amd/csrc > /dev/null 2>&1
cp /tmp/qgs_prev/qgs_amd/libqgs_hip.so "$root/qgs_amd/libqgs_hip_old.so"
git -C "$root" worktree remove --force /tmp/qgs_prev
echo "built $root/qgs_amd/libqgs_hip_old.so from $(git -C "$root" rev-parse --short HEAD)"
