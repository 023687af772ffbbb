#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s24
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_headline.py -q -x --timeout 800 -k "fused_bottleneck or strip_and_fused" > $O/pytest_c80.log 2>&1; echo "exit $?" >> $O/pytest_c80.log
