#!/bin/bash
# round 3, call G: pointwise fast path of the tile set-up + vector bias loads in conv_v2 / conv_igemm: 1x1 shapes, bench, GPU suite
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3g
mkdir -p $O
export TMPDIR=/tmp
for s in small1 small smalls2 odd; do timeout 120 build/convbench $s 2 all >> $O/convbench_check.log 2>&1; done
for s in l26_1x1 l23_1x1 l2_cv3 l26_cv3 l2_1x1; do timeout 300 build/convbench $s 20 nv2:160x160 nv2:320x160 n128x128/2x2/s2 p0 >> $O/convbench_1x1.log 2>&1; done
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --profile-out $O/ops_b32.json > $O/bench.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
ls -la $O > $O/ls.log
