#!/bin/bash
# PMC counters (two SQ sets, separate passes, counters only) of every kernel of a short bench run, averaged per kernel
# name.  usage on the GPU box: bash tools/pmc_bench.sh <tag> [bench.py args]  ->  gpurun_out/pmc_<tag>.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG="$1"; shift
R="$PWD"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_INSTS_SALU"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $SET -d /tmp/pmc_${TAG}_$i -o p --output-format csv -- \
     python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --lean "$@" > "$R/gpurun_out/pmc_${TAG}_$i.log" 2>&1)
done
python - "$TAG" > "gpurun_out/pmc_${TAG}.txt" <<'PY'
import collections, csv, glob, re, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_%s_*/**/*counter_collection.csv' % tag, recursive=True):
    for row in csv.DictReader(open(f)):
        name = re.sub(r'\(.*', '', row.get('Kernel_Name', ''))
        acc[name][row['Counter_Name']].append(float(row['Counter_Value']))
for name in sorted(acc, key=lambda n: -sum(acc[n].get('SQ_BUSY_CYCLES', [0]))):
    c = {k: sum(v) / len(v) for k, v in acc[name].items()}
    n = len(next(iter(acc[name].values())))
    print('%s   (%d dispatches; per-dispatch means)' % (name, n))
    for k in sorted(c):
        print('    %-28s %.6g' % (k, c[k]))
    w = c.get('SQ_WAVE_CYCLES')
    if w:
        print('    -> wait %.1f %%, issue-stall %.1f %%, LDS-issue-stall %.1f %%, active %.1f %% of wave cycles; MFMA busy %.1f %% of SQ busy x4'
              % (100 * c.get('SQ_WAIT_ANY', 0) / w, 100 * c.get('SQ_WAIT_INST_ANY', 0) / w, 100 * c.get('SQ_WAIT_INST_LDS', 0) / w,
                 100 * c.get('SQ_ACTIVE_INST_ANY', 0) / w, 100 * c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(1.0, 4 * c.get('SQ_BUSY_CYCLES', 1))))
    if c.get('SQ_INSTS_MFMA'):
        print('    -> per MFMA: VALU %.2f, LDS %.2f, VMEM %.3f, SALU %.2f; LDS bank-conflict cycles / LDS active cycles %.3f'
              % (c.get('SQ_INSTS_VALU', 0) / c['SQ_INSTS_MFMA'], c.get('SQ_INSTS_LDS', 0) / c['SQ_INSTS_MFMA'],
                 c.get('SQ_INSTS_VMEM', 0) / c['SQ_INSTS_MFMA'], c.get('SQ_INSTS_SALU', 0) / c['SQ_INSTS_MFMA'],
                 c.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, c.get('SQ_LDS_IDX_ACTIVE', 1))))
PY
rm -rf /tmp/pmc_${TAG}_*
head -60 "gpurun_out/pmc_${TAG}.txt"
