#!/bin/bash
# round 6, session 1: the prefetch wave of the pointwise ring tiles (conv_v2 "a3p") against "a3" per shape and distance,
# the ring instantiation's ablations, and the bench line of this box before any table change
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s1
mkdir -p $O
export TMPDIR=/tmp
CB=build/convbench
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -4; nproc) > $O/env.txt 2>&1
for sh in l26_cv3 l19_cv12 l23_cv12 l20_1x1 l8_cv3 l15_cv12 l29_cv12 l4_cv3 l26_1x1 l10_cv3 l8_1x1 l23_1x1; do
  echo "== $sh" >> $O/pf_by_shape.txt
  timeout 120 $CB $sh 20 nv2:160x160/2x2 nv2:320x160/4x2/a3 nv2:256x160/4x2/a3 >> $O/pf_by_shape.txt 2>&1
done
for d in 1 2 4 6 8 12 16; do
  for sh in l26_cv3 l23_cv12 l15_cv12 l26_1x1; do
    echo "== $sh ahead=$d" >> $O/pf_distance.txt
    MDHIP_DEV_PARAM=$d timeout 120 $CB $sh 20 nv2:320x160/4x2/a3p >> $O/pf_distance.txt 2>&1
  done
done
for sh in l26_cv3 l23_cv12; do
  echo "== $sh" >> $O/pf_ablation.txt
  timeout 200 $CB $sh 20 p6 p7 p8 p9 p10 p11 p12 >> $O/pf_ablation.txt 2>&1
done
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra-configs --profile-out $O/ops_b32.json > $O/bench.log 2>&1
echo "bench exit $?" >> $O/bench.log
ls -la $O > $O/ls.log
