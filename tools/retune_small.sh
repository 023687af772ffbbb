#!/bin/bash
# Re-tune of the small batch sizes after adding tile configurations (run on the GPU box):
#   bash tools/retune_small.sh "<name prefixes for --only>" [batches]
# merges into copies of the shipped tables under gpurun_out/ (bf16 and fp16 storage).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
ONLY="$1"; BATCHES="${2:-1 2 4 8}"
cp megadetector_amd/tuned_cfgs.json gpurun_out/tuned_cfgs.json
cp megadetector_amd/tuned_cfgs_fp16.json gpurun_out/tuned_cfgs_fp16.json
for B in $BATCHES; do
  timeout 600 python tools/autotune.py --only "$ONLY" --batch $B --out gpurun_out/tuned_cfgs.json > gpurun_out/retune_bf16_b$B.txt 2>&1
  timeout 600 python tools/autotune.py --only "$ONLY" --batch $B --dtype fp16 --out gpurun_out/tuned_cfgs_fp16.json > gpurun_out/retune_fp16_b$B.txt 2>&1
done
tail -n 3 gpurun_out/retune_*.txt
