#!/bin/bash
# Round 6: refresh of the batch 1 / 2 / 4 / 8 / 16 entries of both 16-bit tile tables at 1280 x 1280 with every kernel family of today
# (most of them date from rounds 2 - 3): a full autotune per batch size, adopted per entry only where the winner is >= 4 % faster in the
# same run and of the same summation-order family (tools/adopt_same_family.py).  The 80-channel bottleneck 3x3s keep their entry: the
# table names the fused four-row kernel for them, which an isolated op timing cannot price (tools/c80_ab.py does).
# usage on the GPU box:  bash tools/retune_small_batches.sh     -> gpurun_out/retune_small/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/retune_small
mkdir -p $O
python - > $O/names.json <<'PY'
import json, sys
sys.path.insert(0, '.')
from megadetector_amd import weights_io, yolo_yaml
from megadetector_amd.hip_backend import HipContext
ctx = HipContext(weights_io.synthetic_weights(yolo_yaml.YOLOV5N6_TEST, seed=1), device=0, dtype='bf16', max_batch=1, max_h=256, max_w=256)
print(json.dumps([ctx.conv_cfg_name(c) for c in range(ctx.num_conv_cfgs())]))
ctx.close()
PY
for DT in bf16 fp16; do
  T=megadetector_amd/tuned_cfgs.json; [ $DT = fp16 ] && T=megadetector_amd/tuned_cfgs_fp16.json
  cp $T $O/before_$(basename $T)
  for B in 1 2 4 8 16; do
    cp $T $O/retuned_${DT}_b$B.json
    timeout 300 python tools/autotune.py --dtype $DT --batch $B --iters 10 --reps 3 --family-from $T --out $O/retuned_${DT}_b$B.json --table $O/table_${DT}_b$B.txt > $O/autotune_${DT}_b$B.log 2>&1 || echo "autotune $DT $B failed"
    echo "== $DT batch $B" | tee -a $O/adopted.txt
    python tools/adopt_same_family.py $T $O/retuned_${DT}_b$B.json $O/table_${DT}_b$B.txt --names $O/names.json | tee -a $O/adopted.txt
  done
  cp $T $O/after_$(basename $T)
done
ls -la $O > $O/ls.log
