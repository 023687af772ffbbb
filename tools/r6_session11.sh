#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s11
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -k "preprocess or modern_mode" > $O/pytest_pre.log 2>&1; echo "exit $?" >> $O/pytest_pre.log
timeout 300 python tools/letterbox_bench.py > $O/letterbox_bench.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_headline.py -q -x --timeout 900 -k "real_letterbox" > $O/pytest_real.log 2>&1; echo "exit $?" >> $O/pytest_real.log
ls -la $O > $O/ls.log
