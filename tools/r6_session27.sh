#!/bin/bash
# round 6, session 27: why the general letterbox kernel measures 0.9 ms on 1080p sources the second time tools/letterbox_bench.py times it
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s27
mkdir -p $O
export TMPDIR=/tmp
timeout 200 python tools/letterbox_bench.py --src 1080x1920 > $O/lb_1080.txt 2>&1
timeout 200 python tools/letterbox_bench.py --src 1080x1920,1080x1920 --iters 20 >> $O/lb_1080.txt 2>&1
RP="$PWD"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$RP/$O/prof" -o lb -- python "$RP/tools/letterbox_bench.py" --src 1080x1920 --iters 10 > "$RP/$O/rocprof.log" 2>&1)
python3 - <<'PY' > $O/kernel_durations.txt 2>&1
import csv, glob, collections
f = glob.glob('gpurun_out/s27/prof/**/lb_kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
seq = [(r['Kernel_Name'][:40], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows if 'letterbox' in r['Kernel_Name']]
print(len(seq))
for i in range(0, len(seq), 13): print(i, seq[i])
PY
find $O/prof -type f -size +1M -delete 2>/dev/null
ls -la $O > $O/ls.log
