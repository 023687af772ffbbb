#!/bin/bash
# round 6, session 30: after the small-batch table refresh: the whole GPU suite, the small-batch lines in both 16-bit types
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s30
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
for b in 1 2 4 8 16; do timeout 200 python bench.py --batch $b --steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs > $O/bench_b$b.log 2>&1; done
for b in 4 8 16; do timeout 200 python bench.py --dtype fp16 --batch $b --steps 100 --warmup 10 --no-cpu-baseline --no-extra-configs > $O/bench_fp16_b$b.log 2>&1; done
ls -la $O > $O/ls.log
