#!/bin/bash
# Adds tile-table entries measured at the letterboxed shape of 4:3 camera-trap frames (960x1280) to the shipped tables
# (same kernel family as the batch-32 1280x1280 entry of every layer).  GPU box: bash tools/tune_real_shape.sh [batches]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
BATCHES="${1:-32}"
for DT in bf16 fp16; do
  SRC=megadetector_amd/tuned_cfgs.json; [ $DT = fp16 ] && SRC=megadetector_amd/tuned_cfgs_fp16.json
  OUT=gpurun_out/real_$(basename $SRC)
  cp $SRC $OUT; cp $SRC gpurun_out/real_canon_$DT.json
  for b in $BATCHES; do
    timeout 900 python tools/autotune.py --batch $b --hw 960x1280 --iters 5 --dtype $DT --family-from gpurun_out/real_canon_$DT.json \
        --out $OUT --table gpurun_out/real_table_${DT}_b$b.txt > gpurun_out/real_autotune_${DT}_b$b.log 2>&1
    echo "$DT batch $b: exit $?"
  done
done
