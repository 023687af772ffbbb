#!/bin/bash
# round 3, call C: ablations of the 8-wave conv_v5 tile, re-tune of the bf16 table with the new configurations, bench, x6 precision test
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3c
mkdir -p $O
export TMPDIR=/tmp
for s in l26_3x3 l6_3x3r; do timeout 300 build/convbench $s 20 nv5:run128x160 nv5:run320x160 t0 t3 t4 t5 t6 >> $O/convbench_ablate.log 2>&1; done
timeout 600 python -m pytest tests/test_gpu_precision_x6.py -q -s --timeout 500 > $O/pytest_precision.log 2>&1; echo "pytest exit $?" >> $O/pytest_precision.log
cp megadetector_amd/tuned_cfgs.json $O/tuned_cfgs.json
timeout 900 python tools/autotune.py --only "v5:run160x320,v5:run320x160,v7:" --out $O/tuned_cfgs.json --table $O/autotune_only_b32.txt > $O/autotune.log 2>&1
cp $O/tuned_cfgs.json megadetector_amd/tuned_cfgs.json
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --profile-out $O/ops_b32.json > $O/bench.log 2>&1
ls -la $O > $O/ls.log
