"""
bench.py as a self-contained launcher (the reference's recipe is one command per GPU,
notebooks/manage_local_batch.py:617-621): `python bench.py --gpus N` without a launcher starts N ranks through
torch.distributed.run on 127.0.0.1 and prints ONE line with n_gpus == N; under a launcher the world size must equal
--gpus.  The rendezvous check stops before anything touches a GPU, so this runs in the CPU suite (gloo, world 2).
Also here: the cpu_baseline leg (thread sweep, forward / NMS split) on a small family member, and the forced
single-worker CPU pinning.
"""

import json
import os
import subprocess
import sys

import pytest

from conftest import REPO


def _env():
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env['OMP_NUM_THREADS'] = '1'
    return env


def test_bench_gpus_n_launches_its_own_ranks():
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--rendezvous-check'],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout                       # ONE JSON line, from rank 0
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and sorted(line['ranks']) == [0, 1]
    assert 'torch.distributed.run' in r.stderr and '--nproc-per-node 2' in r.stderr


def test_bench_refuses_a_world_size_that_is_not_gpus():
    env = _env()
    env.update(WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '4', '--rendezvous-check'],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'WORLD_SIZE=2' in r.stderr
    # and the driver's N = 1 command stays a plain one-process run
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '1', '--rendezvous-check'],
                       env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and json.loads(r.stdout.strip())['n_gpus'] == 1


def test_cpu_baseline_reports_the_best_thread_count_and_the_split():
    import torch
    import bench
    from megadetector_amd import weights_io, yolo_yaml
    w = weights_io.synthetic_weights(yolo_yaml.YOLOV5N6_TEST, seed=0)
    before = torch.get_num_threads()
    res = bench.cpu_baseline(w, 128, 0.2, budget_s=2.0, single_thread=True)
    assert torch.get_num_threads() == before
    assert res['kind'] == 'port' and res['unit'] == 'images/s' and res['value'] > 0
    assert str(res['cores']) in res['thread_sweep_images_per_s'] and res['cores'] <= res['host_threads_available']
    assert res['forward_s_per_image'] > 0 and res['nms_format_s_per_image'] >= 0
    assert res['single_thread']['cores'] == 1


@pytest.mark.skipif(not hasattr(os, 'sched_setaffinity'), reason='no affinity API')
def test_a_lone_worker_is_pinned_only_when_asked():
    from megadetector_amd import placement as P
    code = r'''
import os, sys, json
sys.path.insert(0, {repo!r})
from megadetector_amd import placement as P
allowed = sorted(os.sched_getaffinity(0))
topo = {{'gpu_node': [0], 'node_cpus': {{0: allowed[:max(1, len(allowed) // 2)]}}, 'allowed': allowed}}
a = P.pin_worker(0, 1, topology=topo, verbose=False)
b = P.pin_worker(0, 1, topology=topo, verbose=False, force=True)
print(json.dumps([a == allowed, b, sorted(os.sched_getaffinity(0)), allowed]))
'''.format(repo=REPO)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=60, env=_env())
    assert r.returncode == 0, r.stderr
    untouched, pinned, mask, allowed = json.loads(r.stdout.strip().splitlines()[-1])
    assert untouched
    assert pinned == allowed[:max(1, len(allowed) // 2)] == mask
