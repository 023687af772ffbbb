#!/bin/bash
# Adds small-batch entries (same kernel family as the batch-32 entry of every layer) to the tile table.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
cp megadetector_amd/tuned_cfgs.json gpurun_out/tuned_cfgs.json
cp megadetector_amd/tuned_cfgs.json gpurun_out/tuned_canon.json
for b in 1 2 4 8 16; do
  timeout 600 python tools/autotune.py --batch $b --iters 20 --family-from gpurun_out/tuned_canon.json \
      --out gpurun_out/tuned_cfgs.json --table gpurun_out/tuned_table_b$b.txt > gpurun_out/autotune_b$b.log 2>&1
  echo "batch $b: exit $?"
done
cp gpurun_out/tuned_cfgs.json megadetector_amd/tuned_cfgs.json
