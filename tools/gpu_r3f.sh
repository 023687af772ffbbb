#!/bin/bash
# round 3, call F: residual prefetch depth in the 8-wave conv_v5 tiles, ablations of conv_v2 on the 1x1 shapes, precision test, GPU suite
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3f
mkdir -p $O
export TMPDIR=/tmp
for s in l6_3x3r l26_3x3; do timeout 300 build/convbench $s 20 nv5:run128x160 nv5:run320x160 nv5:run160x320 >> $O/convbench_res.log 2>&1; done
for s in l26_1x1 l23_1x1 l2_cv3 l26_cv3 l2_1x1; do timeout 300 build/convbench $s 20 nv2:160x160 nv2:320x160 p0 p5 p2 >> $O/convbench_1x1.log 2>&1; done
for s in l1_s2 l3_s2 l5_s2; do timeout 300 build/convbench $s 20 nv2:160x160 nv2:320x160 p0 p5 >> $O/convbench_s2.log 2>&1; done
timeout 600 python -m pytest tests/test_gpu_precision_x6.py -q -s --timeout 500 > $O/pytest_precision.log 2>&1; echo "pytest exit $?" >> $O/pytest_precision.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --deselect tests/test_gpu_precision_x6.py > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
ls -la $O > $O/ls.log
