#!/bin/bash
# round 3, call I: re-tune the bf16 table at batch 32 (conv_v2 pointwise instantiations, 8-wave conv_v5 tiles), bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3i
mkdir -p $O
export TMPDIR=/tmp
cp megadetector_amd/tuned_cfgs.json $O/tuned_cfgs.json
timeout 1200 python tools/autotune.py --only "v2:,v5:run320x160,v5:run160x320,v5:run128x160,v5:run256x160" --out $O/tuned_cfgs.json --table $O/autotune_b32.txt > $O/autotune.log 2>&1
cp $O/tuned_cfgs.json megadetector_amd/tuned_cfgs.json
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --profile-out $O/ops_b32.json > $O/bench.log 2>&1
ls -la $O > $O/ls.log
