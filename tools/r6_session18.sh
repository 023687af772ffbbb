#!/bin/bash
# round 6, session 18: golden tile lists re-recorded, the whole GPU suite and the headline on HEAD
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s18
mkdir -p $O
export TMPDIR=/tmp
cp tests/golden/bench_tiles.json $O/bench_tiles_committed.json
timeout 300 python tools/dump_bench_tiles.py > $O/dump_tiles.log 2>&1; cp tests/golden/bench_tiles.json $O/bench_tiles.json
cmp $O/bench_tiles_committed.json $O/bench_tiles.json > $O/tiles_cmp.txt 2>&1; echo "cmp exit $?" >> $O/tiles_cmp.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.log 2>&1
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra-configs > $O/bench_b.log 2>&1
ls -la $O > $O/ls.log
