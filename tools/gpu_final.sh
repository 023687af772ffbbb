#!/bin/bash
# Round-3 final measurement session (everything committed under profiles/r3_* comes from gpurun_out/final3/): re-tunes of the
# bf16 / fp16 / fp8 tables for the kernels of this round, the GPU test suite and smoke on the final tables, bench lines,
# rocprofv3 kernel trace and the counter passes (separate runs, counters only).  Optional parts are skipped once the
# session has used its time (T_MAX seconds).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/final3
mkdir -p $O
export TMPDIR=/tmp
T0=$(date +%s); T_MAX=${T_MAX:-900}
left() { [ $(( $(date +%s) - T0 )) -lt $T_MAX ]; }
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -6; nproc) > $O/env.txt 2>&1
ONLY="v2:,v5:run320x160,v5:run160x320,v5:run128x160,v5:run256x160"
cp megadetector_amd/tuned_cfgs.json $O/tuned_cfgs.json
timeout 400 python tools/autotune.py --only "$ONLY" --out $O/tuned_cfgs.json --table $O/autotune_bf16_b32.txt > $O/autotune_bf16.log 2>&1 && cp $O/tuned_cfgs.json megadetector_amd/tuned_cfgs.json
cp megadetector_amd/tuned_cfgs_fp16.json $O/tuned_cfgs_fp16.json
timeout 400 python tools/autotune.py --dtype fp16 --only "$ONLY" --out $O/tuned_cfgs_fp16.json --table $O/autotune_fp16_b32.txt > $O/autotune_fp16.log 2>&1 && cp $O/tuned_cfgs_fp16.json megadetector_amd/tuned_cfgs_fp16.json
cp megadetector_amd/tuned_cfgs_fp8.json $O/tuned_cfgs_fp8.json
timeout 400 python tools/autotune.py --dtype fp8 --batch 64 --only "v2:" --out $O/tuned_cfgs_fp8.json --table $O/autotune_fp8_b64.txt > $O/autotune_fp8_b64.log 2>&1 && cp $O/tuned_cfgs_fp8.json megadetector_amd/tuned_cfgs_fp8.json
echo "retunes done at $(( $(date +%s) - T0 )) s" > $O/timing.log
# ---- tests and smoke on the final tables ----
timeout 900 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log
echo "tests done at $(( $(date +%s) - T0 )) s" >> $O/timing.log
# ---- bench lines ----
timeout 600 python bench.py --steps 100 --warmup 10 --profile-out $O/ops_b32.json > $O/bench.log 2>&1
timeout 300 python bench.py --dtype fp8 --batch 64 --steps 50 --warmup 5 --no-cpu-baseline --profile-out $O/ops_fp8_b64.json > $O/bench_fp8_b64.log 2>&1
timeout 300 python bench.py --dtype fp16 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_fp16.log 2>&1
echo "main benches done at $(( $(date +%s) - T0 )) s" >> $O/timing.log
# ---- rocprofv3: kernel trace of the bench command, then the counter passes ----
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$O/prof" -o r3 -- \
   python "$OLDPWD/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --lean > "$OLDPWD/$O/rocprof.log" 2>&1)
find $O/prof -type f ! -name "*stats*" -size +2M -delete 2>/dev/null
R="$PWD"
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $C -d "$R/$O/traffic_$C" -o t --output-format csv -- \
     python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --lean > "$R/$O/traffic_$C.log" 2>&1)
done
python tools/hbm_traffic.py $O/traffic_FETCH_SIZE $O/traffic_WRITE_SIZE YOLOV5X6_MD:32:1280 $O/hbm_traffic.json > $O/hbm_traffic.log 2>&1
find $O/traffic_FETCH_SIZE $O/traffic_WRITE_SIZE -type f -size +1M -delete 2>/dev/null
echo "trace + traffic done at $(( $(date +%s) - T0 )) s" >> $O/timing.log
left && { bash tools/pmc_bench.sh final3 > $O/pmc_bench.log 2>&1; cp gpurun_out/pmc_final3.txt gpurun_out/pmc_final3.json $O/ 2>/dev/null; }
echo "pmc done at $(( $(date +%s) - T0 )) s" >> $O/timing.log
left && timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --src 1536x2048 > $O/bench_real43.log 2>&1
left && timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --src 1080x1920 > $O/bench_video_1080p.log 2>&1
left && timeout 300 python bench.py --dtype fp8 --batch 32 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_fp8_b32.log 2>&1
for b in 1 8 16 2 4; do
  left && timeout 200 python bench.py --batch $b --steps 60 --warmup 10 --no-cpu-baseline --lean > $O/bench_b$b.log 2>&1
done
left && timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --host-fed > $O/bench_hostfed.log 2>&1
echo "all done at $(( $(date +%s) - T0 )) s" >> $O/timing.log
ls -laR $O > $O/ls.log
