#!/bin/bash
# round 6, session 28: the two batch-16 shapes of the fp8 table that lost their entries with conv_v4: autotune at batch 16, fp8 mode
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s28
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python tools/autotune.py --dtype fp8 --batch 16 --iters 5 --reps 2 --family-from megadetector_amd/tuned_cfgs_fp8.json --out $O/tuned_fp8_b16.json > $O/autotune_fp8_b16.log 2>&1
ls -la $O > $O/ls.log
