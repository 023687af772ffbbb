#!/bin/bash
# round 6, session 14: every conv_v5 tile on the K = 1440 3x3 layers (with / without residual), every tile on the L1 stride-2 conv
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s14
mkdir -p $O
export TMPDIR=/tmp
for sh in l4_3x3r l23_3x3 l6_3x3r; do
  timeout 200 build/convbench $sh 20 nv5:run > $O/v5_tiles_$sh.txt 2>&1
done
timeout 200 build/convbench l1_s2 20 all > $O/l1_all.txt 2>&1
timeout 200 build/convbench l2_3x3 20 nv5: > $O/l2_v5.txt 2>&1
ls -la $O > $O/ls.log
