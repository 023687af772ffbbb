#!/bin/bash
# round 3, call D: conv_v7 continuous-DMA schedule next to conv_v5; x6 precision test; GPU suite
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3d
mkdir -p $O
export TMPDIR=/tmp
for s in q320 q160 p40; do timeout 120 build/convbench $s 2 nv7: s0 s1 s3 >> $O/convbench_check.log 2>&1; done
for s in l26_3x3 l6_3x3r; do timeout 300 build/convbench $s 20 nv5:run128x160 nv5:run320x160 nv5:run160x320 nv7: s0 s1 s2 s3 >> $O/convbench_perf.log 2>&1; done
for s in l23_3x3 l29_3x3; do timeout 300 build/convbench $s 20 nv5:run128x160 nv5:run320x160 nv7: s0 >> $O/convbench_perf.log 2>&1; done
timeout 600 python -m pytest tests/test_gpu_precision_x6.py -q -s --timeout 500 > $O/pytest_precision.log 2>&1; echo "pytest exit $?" >> $O/pytest_precision.log
timeout 900 python -m pytest tests/test_gpu_headline.py -q --timeout 800 -k "eight_wave" > $O/pytest_eight.log 2>&1; echo "pytest exit $?" >> $O/pytest_eight.log
ls -la $O > $O/ls.log
