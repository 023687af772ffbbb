#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/lean2
mkdir -p $O
{
for sh in l6_3x3r l23_3x3 l2_3x3; do
  echo "== $sh before"
  timeout 120 build/convbench_a $sh 20 nv5:run128x160/2x2 nv5:run256x160 nv5:run128x80 nf8:run128x160 2>&1 | grep -v "nan\|waves,"
  echo "== $sh after (run pieces addressed from the tensor's first byte in every conv_v5 / conv_f8 tile, tap select pinned)"
  timeout 120 build/convbench $sh 20 nv5:run128x160/2x2 nv5:run256x160 nv5:run128x80 nf8:run128x160 2>&1 | grep -v "nan\|waves,"
done
} > $O/convbench_abs_run.txt 2>&1
cat $O/convbench_abs_run.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.log
