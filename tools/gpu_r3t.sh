#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
for sh in l26_1x1 l26_cv3 l23_1x1 l2_cv3; do
  echo "== $sh before"
  timeout 120 build/convbench_a $sh 20 nv2:160x160 nv2:128x160 2>&1 | grep -v "nan\|waves,\|HW_ID"
  echo "== $sh after (pointwise: no VALU instruction per DMA piece)"
  timeout 120 build/convbench $sh 20 nv2:160x160 nv2:128x160 2>&1 | grep -v "nan\|waves,\|HW_ID"
done
CONVBENCH_B=3 timeout 60 build/convbench l26_1x1 5 nv2:160x160 nv2:128x160 nv2:128x80 2>&1 | grep -v "nan\|waves,\|HW_ID"
CONVBENCH_B=1 timeout 60 build/convbench l23_1x1 5 nv2:160x160 nv2:96x160 2>&1 | grep -v "nan\|waves,\|HW_ID"
} > gpurun_out/convbench_pw_lean.txt 2>&1
cat gpurun_out/convbench_pw_lean.txt
timeout 600 python -m pytest tests/test_gpu_headline.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
