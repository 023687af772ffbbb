#!/bin/bash
# Round 3: re-tune of every table entry group that the final session does not cover -- batches 1 .. 16 at 1280 x 1280 (same
# kernel family as the batch-32 entry of every layer: --family-from), the 4:3 / 16:9 / 3:2 letterbox shapes at batch 32 --
# for bf16 and fp16 storage; then the batch-invariance tests and the bench lines that use those entries.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/retune
mkdir -p $O
ONLY="v2:,v5:run320x160,v5:run160x320,v5:run128x160,v5:run256x160,v5:run128x80,v5:run192x80,v5:strip"
for DT in bf16 fp16; do
  T=tuned_cfgs.json; [ $DT = fp16 ] && T=tuned_cfgs_fp16.json
  cp megadetector_amd/$T $O/$T; cp megadetector_amd/$T $O/canon_$DT.json
  for b in 1 2 4 8 16; do
    timeout 300 python tools/autotune.py --dtype $DT --batch $b --iters 10 --family-from $O/canon_$DT.json --out $O/$T --table $O/table_${DT}_b$b.txt > $O/autotune_${DT}_b$b.log 2>&1 || echo "autotune $DT b$b failed"
  done
  for hw in 960x1280 768x1280 896x1280; do
    timeout 300 python tools/autotune.py --dtype $DT --hw $hw --only "$ONLY" --out $O/$T --table $O/table_${DT}_$hw.txt > $O/autotune_${DT}_$hw.log 2>&1 || echo "autotune $DT $hw failed"
  done
  cp $O/$T megadetector_amd/$T
done
timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3 | tee $O/pytest.log
for b in 1 2 4 8 16; do
  timeout 200 python bench.py --batch $b --steps 60 --warmup 10 --no-cpu-baseline --lean > $O/bench_b$b.log 2>&1
done
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --src 1536x2048 > $O/bench_real43.log 2>&1
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --src 1080x1920 > $O/bench_video_1080p.log 2>&1
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --src 1600x2400 > $O/bench_real32.log 2>&1
timeout 300 python bench.py --dtype fp16 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_fp16.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/retune/bench_*.log')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1]); print(f, d['value'], d['ms_per_step'], d['config']['workload'][:60])
    except Exception as e: print(f,'ERR',e, open(f).read()[-200:])
PY
