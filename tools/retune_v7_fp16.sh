#!/bin/bash
# Round 5: the stride-2 row-run kernel (conv_v7) for fp16 storage at the real letterbox shapes.  The bf16 table got its v7
# entries at 768 / 896 / 960 x 1280 in round 4; the fp16 table (the storage type a user gets by default) had them at
# 1280 x 1280 only (tests/golden/bench_tiles.json, round-5 recording: "fp16:32x768x1280" launched no v7 configuration).
# Measures v7 against the current entry of every stride-2 layer at batch 32 and adopts it where it wins by >= 3 %
# (tools/adopt_entries.py: a configuration of another summation-order family fixes the layer's kernel for every batch).
# usage on the GPU box:  bash tools/retune_v7_fp16.sh
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/retune_v7_fp16
mkdir -p $O
python - > $O/names.json <<'PY'
import json, sys
sys.path.insert(0, '.')
from megadetector_amd import weights_io, yolo_yaml
from megadetector_amd.hip_backend import HipContext
ctx = HipContext(weights_io.synthetic_weights(yolo_yaml.YOLOV5N6_TEST, seed=1), device=0, dtype='fp16', max_batch=1, max_h=256, max_w=256)
print(json.dumps([ctx.conv_cfg_name(c) for c in range(ctx.num_conv_cfgs())]))
ctx.close()
PY
T=megadetector_amd/tuned_cfgs_fp16.json
for hw in 768x1280 960x1280 896x1280; do
  cp $T $O/retuned_$hw.json
  timeout 300 python tools/autotune.py --dtype fp16 --hw $hw --only "v7:" --out $O/retuned_$hw.json --table $O/table_fp16_$hw.txt > $O/autotune_fp16_$hw.log 2>&1 || echo "autotune fp16 $hw failed"
  python tools/adopt_entries.py $T $O/retuned_$hw.json $O/table_fp16_$hw.txt --names $O/names.json --prefix v7: | tee -a $O/adopted.txt
done
cp $T $O/tuned_cfgs_fp16.json
