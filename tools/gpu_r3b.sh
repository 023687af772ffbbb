#!/bin/bash
# round 3, call B: conv_v7 (role-split schedule) correctness + speed next to conv_v5, then the GPU test suite
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3b
mkdir -p $O
export TMPDIR=/tmp
for s in q320 q160 p40; do timeout 120 build/convbench $s 2 nv7: s0 s1 s2 s3 >> $O/convbench_check.log 2>&1; done
for s in l26_3x3 l6_3x3r; do timeout 300 build/convbench $s 20 nv5:run128x160 nv5:run160x320 nv5:run320x160 nv7: s0 s1 s2 s3 >> $O/convbench_perf.log 2>&1; done
for s in l23_3x3 l29_3x3 l32_3x3; do timeout 300 build/convbench $s 20 nv5:run128x160 nv5:run320x160 nv7: s3 >> $O/convbench_perf.log 2>&1; done
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
ls -la $O > $O/ls.log
