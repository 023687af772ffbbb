#!/bin/bash
# round 6, session 5: bilinear letterbox v2 (batched loads, v_perm + v_dot2), tile lists after the Detect adoption, headline
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s5
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -k "preprocess or modern_mode" > $O/pytest_pre.log 2>&1; echo "exit $?" >> $O/pytest_pre.log
for src in 1536x2048 1080x1920 1600x2400; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --src $src > $O/bench_$src.log 2>&1
done
timeout 300 python tools/dump_bench_tiles.py > $O/dump_tiles.log 2>&1; cp tests/golden/bench_tiles.json $O/bench_tiles.json
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra-configs --profile-out $O/ops_b32.json > $O/bench.log 2>&1
timeout 900 python -m pytest tests/test_gpu_headline.py -q -x --timeout 900 > $O/pytest_headline.log 2>&1; echo "exit $?" >> $O/pytest_headline.log
ls -la $O > $O/ls.log
