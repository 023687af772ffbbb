#!/bin/bash
# round 6, session 20: the multi-rank lines (one GPU, gloo hook) after the CPU-baseline thread sweep was bounded
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s20
mkdir -p $O
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
T0=$(date +%s)
st() { echo "$1 at $(( $(date +%s) - T0 )) s" >> $O/timing.log; }
MDHIP_BENCH_DUMP_AFTER=250 MDHIP_BENCH_ONE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --batch 8 --steps 30 --warmup 5 > $O/bench_2rank_one_gpu_gloo.log 2>&1; st "torchrun 2"
MDHIP_BENCH_DUMP_AFTER=250 MDHIP_BENCH_ONE_GPU=1 timeout 300 python bench.py --gpus 2 --batch 8 --steps 30 --warmup 5 > $O/bench_selflaunch_2rank.log 2> $O/bench_selflaunch_2rank.err; st "selflaunch 2"
MDHIP_BENCH_DUMP_AFTER=350 MDHIP_BENCH_ONE_GPU=1 timeout 400 python bench.py --gpus 8 --batch 2 --steps 20 --warmup 3 > $O/bench_8rank_one_gpu_gloo.log 2> $O/bench_8rank_one_gpu_gloo.err; st "selflaunch 8"
timeout 600 python bench.py > $O/bench.log 2>&1; st "headline"
ls -la $O > $O/ls.log
