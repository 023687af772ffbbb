#!/bin/bash
# round 6, session 23: after the batch-1 table entry: the whole GPU suite, the small-batch lines, the headline
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s23
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rP --timeout 900 > $O/pytest_gpu_full.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu_full.log
grep -E "^\.|passed|failed|pytest exit|x6 checkpoint|sparse x6|worst|predictions:|fp8 x6|conf|tile configurations" $O/pytest_gpu_full.log | cut -c1-600 > $O/pytest_gpu.log
for b in 1 2 4 8 16; do timeout 200 python bench.py --batch $b --steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs > $O/bench_b$b.log 2>&1; done
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra-configs > $O/bench_b32.log 2>&1
ls -la $O > $O/ls.log
