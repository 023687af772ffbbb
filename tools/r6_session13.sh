#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s13
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --profile-out $O/ops_bf16.json > $O/bench_bf16.log 2>&1
timeout 300 python bench.py --dtype fp16 --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --profile-out $O/ops_fp16.json > $O/bench_fp16.log 2>&1
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs > $O/bench_bf16_b.log 2>&1
timeout 300 python bench.py --dtype fp16 --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs > $O/bench_fp16_b.log 2>&1
timeout 600 python -m pytest tests/test_gpu_fp8.py -q -x --timeout 600 > $O/pytest_fp8.log 2>&1; echo "exit $?" >> $O/pytest_fp8.log
ls -la $O > $O/ls.log
