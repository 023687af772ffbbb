#!/bin/bash
# round 6, session 26: RL (first re-fetched tap) 3 / 4 / 5 of the four-row fused bottleneck without stamps
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s26
mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do CONVBENCH_FUSED=1 timeout 300 build/convbench l2_3x3 30 nv5:strip64 ndev:strip64x80/2x5/r4/nb2rl4-nostamps ndev:strip64x80/2x5/r4/nb2rl3-nostamps >> $O/fused_rl.txt 2>&1; done
ls -la $O > $O/ls.log
