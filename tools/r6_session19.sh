#!/bin/bash
# round 6, session 19: why the 2-rank one-GPU runs of the final session ran into their timeout
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s19
mkdir -p $O
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
date +%s > $O/t0
MDHIP_BENCH_DUMP_AFTER=100 MDHIP_BENCH_ONE_GPU=1 timeout 200 python bench.py --gpus 2 --batch 8 --steps 30 --warmup 5 > $O/selflaunch_2.log 2> $O/selflaunch_2.err; echo "exit $? at $(( $(date +%s) - $(cat $O/t0) )) s" >> $O/selflaunch_2.log
MDHIP_BENCH_DUMP_AFTER=100 MDHIP_BENCH_ONE_GPU=1 timeout 200 python bench.py --gpus 2 --batch 8 --steps 30 --warmup 5 --no-cpu-baseline > $O/selflaunch_2_nocpu.log 2> $O/selflaunch_2_nocpu.err; echo "exit $? at $(( $(date +%s) - $(cat $O/t0) )) s" >> $O/selflaunch_2_nocpu.log
MDHIP_BENCH_DUMP_AFTER=100 MDHIP_BENCH_ONE_GPU=1 timeout 200 python bench.py --gpus 2 --batch 2 --steps 20 --warmup 3 --no-cpu-baseline > $O/selflaunch_2_b2.log 2> $O/selflaunch_2_b2.err; echo "exit $? at $(( $(date +%s) - $(cat $O/t0) )) s" >> $O/selflaunch_2_b2.log
ls -la $O > $O/ls.log
