#!/bin/bash
# round 6, session 17: the four-row fused bottleneck against the strip kernel inside whole forwards, four shapes x two storage types
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s17
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_headline.py -q -x --timeout 800 -k "fused_bottleneck or strip_and_fused" > $O/pytest_c80.log 2>&1; echo "exit $?" >> $O/pytest_c80.log
for dt in bf16 fp16; do
  for shp in 1280x1280 960x1280 896x1280 768x1280; do
    timeout 600 python tools/c80_ab.py --dtype $dt --shape $shp --rounds 2 --reps 6 >> $O/c80_ab.txt 2>&1
  done
done
for b in 1 2 4 8 16; do timeout 300 python tools/c80_ab.py --dtype bf16 --batch $b --rounds 2 --reps 10 >> $O/c80_ab_small.txt 2>&1; done
ls -la $O > $O/ls.log
