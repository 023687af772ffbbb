#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s12
mkdir -p $O
export TMPDIR=/tmp
CB=build/convbench
for sh in l1_s2 l3_s2 l5_s2 l7_s2 l24_s2 l27_s2 s2a s2b s2c s2d s2e s2f; do
  for dp in 67 0 67 0; do
    echo "== $sh MDHIP_DEV_PARAM=$dp (67 = common kernel-row order in every tile; 0 = alternating, the new default)" >> $O/v7_alternate.txt
    MDHIP_DEV_PARAM=$dp timeout 120 $CB $sh 20 nv7: nv2:160x160/2x2 >> $O/v7_alternate.txt 2>&1
  done
done
for sh in l1_s2 l3_s2 l5_s2; do for dp in 67 0; do
  (cd /tmp && MDHIP_DEV_PARAM=$dp timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OLDPWD/$O/pmc_${sh}_$dp" -o f -- "$OLDPWD/$CB" $sh 3 nv7: > "$OLDPWD/$O/pmc_${sh}_$dp.log" 2>&1)
  python3 - "$O/pmc_${sh}_$dp" >> $O/v7_alternate_fetch.txt <<'PY'
import csv, glob, sys, collections
d=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name']=='FETCH_SIZE' and 'conv_v7' in r['Kernel_Name']: d[r['Kernel_Name'][:70]].append(float(r['Counter_Value']))
for k,v in d.items(): print(sys.argv[1], k, 'dispatches', len(v), 'FETCH_SIZE x 2 = %.1f MB per dispatch' % (2*1024*sum(v)/len(v)/1e6))
PY
  find $O/pmc_${sh}_$dp -type f -size +1M -delete 2>/dev/null
done; done
timeout 900 python -m pytest tests -m gpu -q --timeout 900 -x -k "stride2 or row_run or v7 or headline or every_layer or tile_config or batch" > $O/pytest_v7.log 2>&1; echo "exit $?" >> $O/pytest_v7.log
ls -la $O > $O/ls.log
