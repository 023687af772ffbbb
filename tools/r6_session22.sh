#!/bin/bash
# round 6, session 22: the 80-channel block at batch 1: the table's unfused choice against the fused strip kernels
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s22
mkdir -p $O
export TMPDIR=/tmp
for dt in bf16 fp16; do timeout 300 python tools/c80_ab.py --dtype $dt --batch 1 --rounds 3 --reps 20 >> $O/c80_ab_b1.txt 2>&1; done
ls -la $O > $O/ls.log
