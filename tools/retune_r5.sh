#!/bin/bash
# Round 5: refresh of the batch-32 entries of both 16-bit tile tables at the four letterbox shapes after this round's register work in
# conv_v5 (a full autotune of every configuration per shape, ~25 s each), adopted per entry only where the winner is >= 4 % faster in the
# same run and of the same summation-order family (tools/adopt_same_family.py), plus conv_v7 where >= 3 % (tools/adopt_entries.py).
# usage on the GPU box:  bash tools/retune_r5.sh     -> gpurun_out/retune_r5/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/retune_r5
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
  for hw in 1280x1280 768x1280 960x1280 896x1280; do
    cp $T $O/retuned_${DT}_$hw.json
    timeout 300 python tools/autotune.py --dtype $DT --hw $hw --iters 8 --reps 3 --out $O/retuned_${DT}_$hw.json --table $O/table_${DT}_$hw.txt > $O/autotune_${DT}_$hw.log 2>&1 || echo "autotune $DT $hw failed"
    echo "== $DT $hw" | tee -a $O/adopted.txt
    python tools/adopt_same_family.py $T $O/retuned_${DT}_$hw.json $O/table_${DT}_$hw.txt --names $O/names.json | tee -a $O/adopted.txt
    python tools/adopt_entries.py $T $O/retuned_${DT}_$hw.json $O/table_${DT}_$hw.txt --names $O/names.json --prefix v7: | tee -a $O/adopted.txt
  done
  cp $T $O/
done
