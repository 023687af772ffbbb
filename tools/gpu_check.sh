cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/check
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/check/build.log 2>&1; echo "build exit $?" >> gpurun_out/check/build.log
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee gpurun_out/check/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/check/smoke.log
timeout 600 python bench.py 2>&1 | tail -1 > gpurun_out/check/bench.json
python -c "
import json; d=json.load(open('gpurun_out/check/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])"
