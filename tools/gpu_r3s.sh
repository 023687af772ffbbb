#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/lean
mkdir -p $O
{
for sh in l26_3x3 l6_3x3r l23_3x3 l29_3x3; do
  echo "== $sh before"
  timeout 120 build/convbench_a $sh 20 nv5:run320x160 nv5:run160x320 2>&1 | grep -v "nan\|waves,"
  echo "== $sh after (lean DMA issue + tap select pinned in the first half)"
  timeout 120 build/convbench $sh 20 nv5:run320x160 nv5:run160x320 2>&1 | grep -v "nan\|waves,"
done
} > $O/convbench_lean_dma2.txt 2>&1
cat $O/convbench_lean_dma2.txt
cp megadetector_amd/libmdhip.so $O/libmdhip_new.so
timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_parity.py -x -q 2>&1 | tail -4 | tee $O/pytest.log
for rep in 1 2; do
  cp $O/libmdhip_new.so megadetector_amd/libmdhip.so
  python bench.py --no-cpu-baseline --steps 60 2>&1 | tail -1 > $O/bench_new_$rep.json
  cp build/libmdhip_head.so megadetector_amd/libmdhip.so
  python bench.py --no-cpu-baseline --steps 60 2>&1 | tail -1 > $O/bench_head_$rep.json
done
cp $O/libmdhip_new.so megadetector_amd/libmdhip.so
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/lean/bench_*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
rm -f $O/libmdhip_new.so
