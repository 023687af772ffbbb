#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
for sh in l26_3x3 l23_3x3; do
  echo "== $sh  t3 = stamps, t7 = no LDS reads of the pixel fragments of taps 1, 2 (wrong results), t8 = same with stamps"
  timeout 120 build/convbench $sh 20 nv5:run320x160/4x2 t3 t7 t8 2>&1 | grep -v "nan"
done
} > gpurun_out/convbench_ldsbound.txt 2>&1
cat gpurun_out/convbench_ldsbound.txt
