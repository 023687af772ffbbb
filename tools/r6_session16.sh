#!/bin/bash
# round 6, session 16: the fused bottleneck kernels in tools/convbench (CONVBENCH_FUSED=1), phase stamps of the four-row kernel
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s16
mkdir -p $O
export TMPDIR=/tmp
CONVBENCH_FUSED=1 timeout 300 build/convbench l2_3x3 20 nv5:strip ndev:strip64x80/2x5/r4/nb2rl5 > $O/fused_l2_b.txt 2>&1
ls -la $O > $O/ls.log
