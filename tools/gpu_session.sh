#!/bin/bash
# One GPU-box visit (stages picked on the command line; any other argument is a command line run verbatim); everything that should come back goes under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for STAGE in "$@"; do
case "$STAGE" in
  probe)
    timeout 60 build/probe_fp8 > gpurun_out/probe_fp8.log 2>&1; echo "exit $?" >> gpurun_out/probe_fp8.log ;;
  test)
    timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -x -s > gpurun_out/pytest_gpu.log 2>&1
    echo "pytest exit $?" >> gpurun_out/pytest_gpu.log ;;
  testall)
    timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/pytest_gpu.log 2>&1
    echo "pytest exit $?" >> gpurun_out/pytest_gpu.log ;;
  acc)
    timeout 900 python tests/accuracy_report.py --x6 > gpurun_out/accuracy_x6.txt 2>&1 ;;
  bench)
    timeout 900 python bench.py --steps 50 --warmup 5 --profile-out gpurun_out/ops_b32.json > gpurun_out/bench.log 2>&1
    echo "bench exit $?" >> gpurun_out/bench.log ;;
  benchlean)
    timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --profile-out gpurun_out/ops_b32.json > gpurun_out/bench.log 2>&1
    echo "bench exit $?" >> gpurun_out/bench.log ;;
  prof)
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o run -- \
       python "$OLDPWD/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --lean > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
    find gpurun_out/prof -name "*stats*" | head >> gpurun_out/rocprof.log
    find gpurun_out/prof -type f ! -name "*stats*" -size +2M -delete 2>/dev/null ;;
  traffic)
    R="$PWD"
    for C in FETCH_SIZE WRITE_SIZE; do
      (cd /tmp && timeout 600 rocprofv3 --pmc $C -d "$R/gpurun_out/traffic_$C" -o t --output-format csv -- \
         python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --lean > "$R/gpurun_out/traffic_$C.log" 2>&1)
    done
    python tools/hbm_traffic.py gpurun_out/traffic_FETCH_SIZE gpurun_out/traffic_WRITE_SIZE YOLOV5X6_MD:32:1280 gpurun_out/hbm_traffic.json > gpurun_out/hbm_traffic.log 2>&1
    find gpurun_out/traffic_FETCH_SIZE gpurun_out/traffic_WRITE_SIZE -type f -size +1M -delete 2>/dev/null ;;
  *)
    # anything else: a command line to run verbatim, output to gpurun_out/cmd.log (appended)
    echo "== $STAGE" >> gpurun_out/cmd.log
    timeout 900 bash -c "$STAGE" >> gpurun_out/cmd.log 2>&1
    echo "exit $?" >> gpurun_out/cmd.log ;;
esac
done
ls -la gpurun_out > gpurun_out/ls.log
