#!/bin/bash
# round 6, session 4: streaming bilinear letterbox (tests + real-shape bench), Detect tiles with the decode fused
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s4
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -k "preprocess or modern_mode" > $O/pytest_pre.log 2>&1; echo "exit $?" >> $O/pytest_pre.log
timeout 600 python -m pytest tests/test_gpu_headline.py -q -x --timeout 900 -k "real_letterbox" > $O/pytest_real.log 2>&1; echo "exit $?" >> $O/pytest_real.log
for src in 1536x2048 1080x1920 1600x2400; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --src $src > $O/bench_$src.log 2>&1
done
for dt in bf16 fp16; do for shp in 1280x1280 960x1280 768x1280 896x1280; do
  timeout 300 python tools/tune_detect_fused.py --dtype $dt --shape $shp >> $O/detect_fused.jsonl 2>> $O/detect_fused.err
done; done
ls -la $O > $O/ls.log
