#!/bin/bash
# round 6, session 3: GPU suite after the Detect decode fusion / SPPF pool change, bench line, fp8 tile stamps
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s3
mkdir -p $O
export TMPDIR=/tmp
CB=build/convbench
timeout 300 python -m pytest tests/test_gpu_headline.py -q -x --timeout 600 -k "detect_decode or fused_bottleneck" -rP > $O/pytest_decode.log 2>&1; echo "exit $?" >> $O/pytest_decode.log
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra-configs --profile-out $O/ops_b32.json > $O/bench.log 2>&1
echo "bench exit $?" >> $O/bench.log
for sh in l26_3x3 l6_3x3r l23_3x3 l29_3x3; do
  echo "== $sh: fp8 tiles (f0 stamps, f1 no DMA, f2 no stores, f3 no stores no SiLU, f4 neither + no DMA on 128x160/2x2; f5 stamps, f6 no DMA, f7 no stores, f8 all three on 256x160/4x2) and the bf16 8-wave tile" >> $O/fp8_stamps.txt
  timeout 300 $CB $sh 20 nf8:run128x160 nf8:run256x160 f0 f1 f2 f3 f4 f5 f6 f7 f8 nv5:run320x160 >> $O/fp8_stamps.txt 2>&1
done
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
ls -la $O > $O/ls.log
