#!/bin/bash
# round 3, call A: GPU tests on the tree + first look at the 8-wave 80x80-wave-tile configurations of conv_v5
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3a
mkdir -p $O
export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -6; nproc) > $O/env.txt 2>&1
for s in q320 q160 p40 odd; do timeout 120 build/convbench $s 2 nv5:run >> $O/convbench_check.log 2>&1; done
for s in l26_3x3 l6_3x3r l23_3x3 l29_3x3; do timeout 300 build/convbench $s 20 nv5:run128x160 nv5:run160x320 nv5:run320x160 nv5:run256x160 >> $O/convbench_perf.log 2>&1; done
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -x > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench.log 2>&1
ls -la $O > $O/ls.log
