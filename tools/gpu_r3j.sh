#!/bin/bash
# round 3, call J: fp8 table entries at batch 64 (+ the pointwise conv_v2 instantiations at batch 32), fp8 tests
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3j
mkdir -p $O
export TMPDIR=/tmp
cp megadetector_amd/tuned_cfgs_fp8.json $O/tuned_cfgs_fp8.json
timeout 900 python tools/autotune.py --dtype fp8 --only "v2:" --out $O/tuned_cfgs_fp8.json --table $O/autotune_fp8_b32.txt > $O/autotune_fp8.log 2>&1
timeout 900 python tools/autotune.py --dtype fp8 --batch 64 --only "v2:" --out $O/tuned_cfgs_fp8.json --table $O/autotune_fp8_b64.txt > $O/autotune_fp8_b64.log 2>&1
cp $O/tuned_cfgs_fp8.json megadetector_amd/tuned_cfgs_fp8.json
timeout 900 python -m pytest tests/test_gpu_fp8.py -q -s --timeout 800 > $O/pytest_fp8.log 2>&1; echo "pytest exit $?" >> $O/pytest_fp8.log
timeout 600 python bench.py --dtype fp8 --batch 64 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_fp8_b64.log 2>&1
ls -la $O > $O/ls.log
