#!/bin/bash
# round 6, session 31: the small-batch table refresh A/B'd on ONE box: old tables (build/old_tables) against the refreshed ones, alternating
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s31
mkdir -p $O $O/new
export TMPDIR=/tmp
cp megadetector_amd/tuned_cfgs.json megadetector_amd/tuned_cfgs_fp16.json $O/new/
run() {  # $1 = label, $2 = dtype, $3 = batch
  grep_line() { grep '^{' "$1" | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
  timeout 200 python bench.py --dtype $2 --batch $3 --steps 150 --warmup 10 --no-cpu-baseline --no-extra-configs --lean > $O/tmp.log 2>&1
  echo "$1 $2 batch $3: $(grep_line $O/tmp.log)" >> $O/ab.txt
}
for rep in 1 2; do
  for dt in bf16 fp16; do
    for b in 4 8 16; do
      cp build/old_tables/*.json megadetector_amd/; run old $dt $b
      cp $O/new/*.json megadetector_amd/;           run new $dt $b
    done
  done
done
cp $O/new/*.json megadetector_amd/
rm -f $O/tmp.log
