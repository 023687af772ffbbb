#!/bin/bash
# round 6, session 2: the pointwise ring tile's epilogue: non-temporal / write-through stores, M streams started out of phase
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s2
mkdir -p $O
export TMPDIR=/tmp
CB=build/convbench
for sh in l26_cv3 l26_1x1 l23_1x1 l23_cv12; do
  echo "== $sh (p17 = reference build of the variant list, p13 = nt stores, p14 = sc1 stores)" >> $O/pw_stores.txt
  timeout 200 $CB $sh 20 p17 p13 p14 p17 p13 p14 >> $O/pw_stores.txt 2>&1
  for d in 0 1 2 3 4 6 8; do
    echo "== $sh stagger dev_param=$d (p15: plain stores, p16: nt stores)" >> $O/pw_stagger.txt
    MDHIP_DEV_PARAM=$d timeout 200 $CB $sh 20 p15 p16 >> $O/pw_stagger.txt 2>&1
  done
done
ls -la $O > $O/ls.log
