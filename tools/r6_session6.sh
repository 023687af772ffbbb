#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s6
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/letterbox_bench.py > $O/letterbox_bench.txt 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$O/prof" -o lb -- python "$OLDPWD/tools/letterbox_bench.py" --src 1536x2048 --iters 20 > "$OLDPWD/$O/rocprof.log" 2>&1)
find $O/prof -name "*kernel_stats*" -exec cp {} $O/letterbox_kernel_stats.csv \;
find $O/prof -type f -size +1M -delete 2>/dev/null
CB=build/convbench
for sh in l23_3x3 l4_3x3r l26_3x3 l6_3x3r; do
  for dp in 0 55 0 55; do
    echo "== $sh MDHIP_DEV_PARAM=$dp (55 = a stream takes a contiguous chunk of tiles)" >> $O/v5_chunked.txt
    MDHIP_DEV_PARAM=$dp timeout 120 $CB $sh 20 nv5:run320x160 >> $O/v5_chunked.txt 2>&1
  done
done
for dp in 0 55; do
  (cd /tmp && MDHIP_DEV_PARAM=$dp timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OLDPWD/$O/pmc_$dp" -o f -- "$OLDPWD/$CB" l23_3x3 3 nv5:run320x160 > "$OLDPWD/$O/pmc_$dp.log" 2>&1)
  python3 - "$O/pmc_$dp" >> $O/v5_chunked_fetch.txt <<'PY'
import csv, glob, sys, collections
d=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name']=='FETCH_SIZE' and 'conv_v5' in r['Kernel_Name']: d[r['Kernel_Name'][:80]].append(float(r['Counter_Value']))
for k,v in d.items(): print(sys.argv[1], k, 'dispatches', len(v), 'FETCH_SIZE KiB per dispatch (x2 for bytes):', sum(v)/len(v))
PY
  find $O/pmc_$dp -type f -size +1M -delete 2>/dev/null
done
ls -la $O > $O/ls.log
