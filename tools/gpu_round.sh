#!/bin/bash
# One GPU-box visit: parity tests, diagnostics, autotune, bench, rocprof.  Everything that
# should come back goes under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
STAGE="${1:-all}"
echo "== rocminfo ==" > gpurun_out/env.log
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; free -g | head -2) >> gpurun_out/env.log 2>&1

if [ "$STAGE" = "all" ] || [ "$STAGE" = "test" ]; then
  timeout 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
  timeout 300 python tests/gpu_diag.py YOLOV5N6_TEST 256 2 > gpurun_out/diag_n6.log 2>&1
fi
if [ "$STAGE" = "all" ] || [ "$STAGE" = "tune" ]; then
  timeout 600 python tools/autotune.py --out gpurun_out/tuned_cfgs.json > gpurun_out/autotune.log 2>&1
  cp gpurun_out/tuned_cfgs.json megadetector_amd/tuned_cfgs.json 2>/dev/null
fi
if [ "$STAGE" = "all" ] || [ "$STAGE" = "bench" ]; then
  timeout 900 python bench.py --steps 10 --warmup 3 --profile-out gpurun_out/ops_b32.json > gpurun_out/bench.log 2>&1
  echo "bench exit $?" >> gpurun_out/bench.log
fi
if [ "$STAGE" = "all" ] || [ "$STAGE" = "prof" ]; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o r1 -- \
     python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --lean > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
  find gpurun_out/prof -name "*stats*" | head >> gpurun_out/rocprof.log
  # keep only the small summaries
  find gpurun_out/prof -type f ! -name "*stats*" -size +2M -delete 2>/dev/null
fi
if [ "$STAGE" = "all" ] || [ "$STAGE" = "traffic" ]; then
  # HBM bytes per step: counters only (no tracing), FETCH_SIZE and WRITE_SIZE in separate passes
  R="$PWD"
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --pmc $C -d "$R/gpurun_out/traffic_$C" -o t --output-format csv -- \
       python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --lean > "$R/gpurun_out/traffic_$C.log" 2>&1)
  done
  python tools/hbm_traffic.py gpurun_out/traffic_FETCH_SIZE gpurun_out/traffic_WRITE_SIZE YOLOV5X6_MD:32:1280 gpurun_out/hbm_traffic.json > gpurun_out/hbm_traffic.log 2>&1
  find gpurun_out/traffic_FETCH_SIZE gpurun_out/traffic_WRITE_SIZE -type f -size +1M -delete 2>/dev/null
fi
ls -la gpurun_out >> gpurun_out/env.log
