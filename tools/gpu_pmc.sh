#!/bin/bash
# PMC passes (counters only, no tracing) on one conv op.  usage: gpu_pmc.sh "<op name>" <cfg> <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
OP="$1"; CFG="$2"; TAG="$3"
R="$PWD"
cd /tmp
rocprofv3 -L > "$R/gpurun_out/pmc/counters_list.txt" 2>&1
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET -d "$R/gpurun_out/pmc/${TAG}_p$i" -o p --output-format csv -- python "$R/tools/pmc_probe.py" "$OP" "$CFG" 3 > "$R/gpurun_out/pmc/${TAG}_p$i.log" 2>&1
done
cd "$R"
python tools/pmc_report.py "$TAG" > "gpurun_out/pmc/${TAG}_summary.txt"; cat "gpurun_out/pmc/${TAG}_summary.txt"; cat > /dev/null <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc/%s_p*/**/*counter_collection.csv' % tag, recursive=True):
    for row in csv.DictReader(open(f)):
        if 'conv_igemm' in row.get('Kernel_Name', ''):
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
with open('gpurun_out/pmc/%s_summary.txt' % tag, 'w') as out:
    for k in sorted(acc):
        v = acc[k]
        out.write('%-28s n=%d mean=%.6g min=%.6g max=%.6g\n' % (k, len(v), sum(v) / len(v), min(v), max(v)))
print(open('gpurun_out/pmc/%s_summary.txt' % tag).read())
PY
find gpurun_out/pmc -type f -size +1M -delete
