#!/bin/bash
# round 6, session 25: the four-row fused bottleneck with the conversion's reads / MFMAs / SiLUs / writes batched per task
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s25
mkdir -p $O
export TMPDIR=/tmp
CONVBENCH_FUSED=1 timeout 300 build/convbench l2_3x3 20 nv5:strip ndev:strip64x80/2x5/r4/nb2rl5 ndev:strip64x80/2x5/r4/nb2rl3 > $O/fused_l2.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_headline.py -q -x --timeout 800 -k "fused_bottleneck or strip_and_fused" > $O/pytest_c80.log 2>&1; echo "exit $?" >> $O/pytest_c80.log
timeout 600 python tools/c80_ab.py --dtype bf16 --rounds 2 --reps 6 > $O/c80_ab.txt 2>&1
ls -la $O > $O/ls.log
