#!/bin/bash
# tile table for the fp16 storage mode: batch 32 first (fixes the families), then the small batches
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/tuned_cfgs_fp16.json
timeout 600 python tools/autotune.py --dtype fp16 --batch 32 --out gpurun_out/tuned_cfgs_fp16.json --table gpurun_out/tuned_table_fp16_b32.txt > gpurun_out/autotune_fp16_b32.log 2>&1
cp gpurun_out/tuned_cfgs_fp16.json gpurun_out/tuned_canon_fp16.json
for b in 1 2 4 8 16; do
  timeout 600 python tools/autotune.py --dtype fp16 --batch $b --iters 20 --family-from gpurun_out/tuned_canon_fp16.json \
      --out gpurun_out/tuned_cfgs_fp16.json --table gpurun_out/tuned_table_fp16_b$b.txt > gpurun_out/autotune_fp16_b$b.log 2>&1
done
cp gpurun_out/tuned_cfgs_fp16.json megadetector_amd/tuned_cfgs_fp16.json
