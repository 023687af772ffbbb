#!/bin/bash
# round 6, session 21: a REAL RCCL failure on a GPU box (two ranks on one device: RCCL refuses the duplicate) -> the gloo fall-back
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s21
mkdir -p $O
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
T0=$(date +%s)
MDHIP_BENCH_DUMP_AFTER=200 MDHIP_BENCH_ONE_GPU=try timeout 280 python bench.py --gpus 2 --batch 8 --steps 30 --warmup 5 > $O/bench_2rank_rccl_refused.log 2> $O/bench_2rank_rccl_refused.err; echo "exit $? at $(( $(date +%s) - T0 )) s" >> $O/bench_2rank_rccl_refused.log
ls -la $O > $O/ls.log
