#!/bin/bash
# Round-2 measurement session: everything that is committed under profiles/ comes from here (gpurun_out/final/).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -6; nproc) > $O/env.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log
timeout 900 python bench.py --steps 100 --warmup 10 --profile-out $O/ops_b32.json > $O/bench.log 2>&1
timeout 600 python bench.py --dtype fp8 --batch 64 --steps 50 --warmup 5 --no-cpu-baseline --profile-out $O/ops_fp8_b64.json > $O/bench_fp8_b64.log 2>&1
timeout 600 python bench.py --dtype fp8 --batch 32 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_fp8_b32.log 2>&1
timeout 600 python bench.py --dtype fp16 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_fp16.log 2>&1
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --host-fed > $O/bench_hostfed.log 2>&1
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --src 1536x2048 > $O/bench_real43.log 2>&1
for b in 1 2 4 8 16; do
  timeout 300 python bench.py --batch $b --steps 60 --warmup 10 --no-cpu-baseline --lean > $O/bench_b$b.log 2>&1
done
MDHIP_BENCH_ONE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
   bench.py --gpus 2 --steps 10 --warmup 3 --batch 8 --no-cpu-baseline --lean > $O/bench_2rank_onegpu.log 2>&1
timeout 900 python tests/accuracy_report.py --x6 > $O/accuracy_x6.txt 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$O/prof" -o r2 -- \
   python "$OLDPWD/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --lean > "$OLDPWD/$O/rocprof.log" 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$O/prof_fp8" -o r2f8 -- \
   python "$OLDPWD/bench.py" --dtype fp8 --batch 64 --steps 6 --warmup 2 --no-cpu-baseline --lean > "$OLDPWD/$O/rocprof_fp8.log" 2>&1)
find $O/prof $O/prof_fp8 -type f ! -name "*stats*" -size +2M -delete 2>/dev/null
R="$PWD"
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C -d "$R/$O/traffic_$C" -o t --output-format csv -- \
     python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --lean > "$R/$O/traffic_$C.log" 2>&1)
done
python tools/hbm_traffic.py $O/traffic_FETCH_SIZE $O/traffic_WRITE_SIZE YOLOV5X6_MD:32:1280 $O/hbm_traffic.json > $O/hbm_traffic.log 2>&1
find $O/traffic_FETCH_SIZE $O/traffic_WRITE_SIZE -type f -size +1M -delete 2>/dev/null
ls -laR $O > $O/ls.log
