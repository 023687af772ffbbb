#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s29
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python bench.py --dtype fp8 --batch 16 --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs > $O/bench_fp8_b16.log 2>&1
timeout 600 python -m pytest tests/test_gpu_fp8.py -q -x --timeout 500 > $O/pytest_fp8.log 2>&1; echo "exit $?" >> $O/pytest_fp8.log
