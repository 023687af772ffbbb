#!/bin/bash
# round 6, session 15: the four-row fused bottleneck of the 80-channel block: bit-identity tests, then the A/B inside whole forwards
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s15
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_headline.py -q -x --timeout 800 -k "fused_bottleneck or strip_and_fused" > $O/pytest_c80.log 2>&1; echo "exit $?" >> $O/pytest_c80.log
timeout 600 python tools/c80_ab.py --dtype bf16 > $O/c80_ab_bf16.txt 2>&1
timeout 600 python tools/c80_ab.py --dtype fp16 --rounds 1 > $O/c80_ab_fp16.txt 2>&1
timeout 600 python tools/c80_ab.py --dtype bf16 --shape 960x1280 --rounds 1 > $O/c80_ab_bf16_960.txt 2>&1
ls -la $O > $O/ls.log
