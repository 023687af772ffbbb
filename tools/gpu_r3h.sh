#!/bin/bash
# round 3, call H: pointwise instantiation of conv_v2 (tile set-up without divisions) + runtime fast path in conv_igemm
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3h
mkdir -p $O
export TMPDIR=/tmp
for s in small1 odd; do timeout 120 build/convbench $s 2 all >> $O/convbench_check.log 2>&1; done
for s in l26_1x1 l23_1x1 l2_cv3 l26_cv3 l2_1x1; do timeout 300 build/convbench $s 20 nv2:160x160 nv2:320x160 n128x128/2x2/s2 >> $O/convbench_1x1.log 2>&1; done
for s in l1_s2 l3_s2; do timeout 300 build/convbench $s 20 nv2:160x160 >> $O/convbench_1x1.log 2>&1; done
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --profile-out $O/ops_b32.json > $O/bench.log 2>&1
ls -la $O > $O/ls.log
