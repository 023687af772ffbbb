#!/bin/bash
# Copies what a tools/gpu_final.sh session left under gpurun_out/final<R>/ into profiles/r<R>_* (the files DESIGN.md and
# profiles/README.md cite).  usage (build container, after the gpurun call returned):  ROUND=5 bash tools/collect_final.sh
set -u
cd "$(dirname "$0")/.."
R=${ROUND:-6}
F=gpurun_out/final$R
P=profiles/r$R
j() { grep '^{' "$1" | tail -1; }
j $F/bench.log > ${P}_bench.json
j $F/bench_fp8_b64.log > ${P}_bench_fp8_b64.json
j $F/bench_fp16.log > ${P}_bench_fp16.json
j $F/bench_hostfed.log > ${P}_bench_hostfed.json
j $F/bench_real43.log > ${P}_bench_real_1536x2048.json
j $F/bench_real32.log > ${P}_bench_real_1600x2400.json
j $F/bench_video_1080p.log > ${P}_bench_video_1080p.json
j $F/bench_2rank_one_gpu_gloo.log > ${P}_bench_2rank_one_gpu_gloo.json
j $F/bench_selflaunch_2rank.log > ${P}_bench_selflaunch_2rank.json
j $F/bench_pinned.log > ${P}_bench_pinned.json
(for b in 1 2 4 8 16; do j $F/bench_b$b.log; done) > ${P}_bench_small_batches.jsonl
(for s in 1080x1920 1536x2048 1600x2400; do j $F/bench_fp16_$s.log; done) > ${P}_bench_fp16_real_shapes.jsonl
python - "$F" > ${P}_bench_nms_stream.txt <<'PY'
import json, sys
f = sys.argv[1]
for label, name in (('NMS + D2H in line behind the forward (--nms-inline)', 'bench_nms_inline.log'), ('own stream (default)', 'bench_nms_own.log')):
    d = json.loads([l for l in open(f + '/' + name) if l.startswith('{')][-1])
    print('{}: {} images/s, {} ms / step'.format(label, d['value'], d['ms_per_step']))
PY
cp $F/prof/r${R}_kernel_stats.csv ${P}_bench_kernel_stats.csv
cp $F/ops_b32.json ${P}_ops_b32.json
cp $F/ops_fp8_b64.json ${P}_ops_fp8_b64.json
[ -s $F/hbm_traffic.json ] || python tools/hbm_traffic.py $F/traffic_FETCH_SIZE $F/traffic_WRITE_SIZE YOLOV5X6_MD:32:1280 $F/hbm_traffic.json
cp $F/hbm_traffic.json ${P}_hbm_traffic.json
[ -s $F/hbm_traffic_by_kernel.txt ] && cp $F/hbm_traffic_by_kernel.txt ${P}_hbm_traffic_by_kernel.txt
[ -s $F/l2_hit_rate_by_kernel.txt ] && grep -v amdgpu $F/l2_hit_rate_by_kernel.txt > ${P}_l2_hit_rate_by_kernel.txt
[ -s $F/bench_8rank_one_gpu_gloo.log ] && j $F/bench_8rank_one_gpu_gloo.log > ${P}_bench_8rank_one_gpu_gloo.json
[ -s $F/bench_2rank_rccl_refused.log ] && j $F/bench_2rank_rccl_refused.log > ${P}_bench_2rank_rccl_refused.json
[ -s $F/letterbox_bench.txt ] && grep -v amdgpu $F/letterbox_bench.txt > ${P}_letterbox_final_session.txt
cp gpurun_out/pmc_final$R.txt ${P}_pmc_bench_kernels.txt
cp gpurun_out/pmc_final$R.json ${P}_pmc_bench_kernels.json
cp $F/e2e_feed.json ${P}_e2e_feed.json
cp $F/placement_box.txt ${P}_placement_box.txt
grep -v amdgpu $F/accuracy_x6.txt > ${P}_accuracy_x6.txt
cp $F/pytest_gpu.log ${P}_pytest_gpu.log
grep -v amdgpu $F/smoke.log > ${P}_smoke.log
cp $F/env.txt ${P}_env.txt
(cat $F/bench_pinned.err $F/bench_selflaunch_2rank.err | grep -v amdgpu | head -12) > ${P}_bench_launch_stderr.txt
cp $F/bench_tiles.json tests/golden/bench_tiles.json
echo "collected $F -> ${P}_*"
