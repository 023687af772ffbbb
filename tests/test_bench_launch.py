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
    # the launcher picks its own rendezvous port (no bind-then-release probe that another process could win)
    assert '--standalone' in r.stderr and '--master-port' not in r.stderr


def test_bench_without_gpus_takes_the_launchers_world_size():
    # `torchrun --nproc-per-node 2 bench.py` (no --gpus): two ranks, one line; an explicit --gpus must still agree
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr', '127.0.0.1', '--nnodes=1',
                        '--nproc-per-node', '2', os.path.join(REPO, 'bench.py'), '--rendezvous-check'],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1 and json.loads(lines[0])['n_gpus'] == 2


_CONTROL_PLANE_RANK = r'''
import os, sys, json
sys.path.insert(0, {repo!r})
import torch, bench
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
if os.environ.get('FAIL_ON_RANK') == str(rank):
    # one rank alone cannot start the device plane: every rank has to end up on the host plane
    import torch.distributed as d
    real = d.new_group
    def broken(*a, **k):
        if k.get('backend') == 'nccl':
            raise RuntimeError('injected: no RCCL on this rank')
        return real(*a, **k)
    d.new_group = broken
dist, group, on_host, name, why = bench.init_control_plane(torch, rank, world, 0, os.environ.get('FORCE_GLOO') == '1',
                                                           timeout_s=120, nccl_timeout_s=20)
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
dist.barrier(group=group)
allr = [None] * world
dist.all_gather_object(allr, [name, on_host, bool(why)])
if rank == 0:
    print(json.dumps({{'planes': allr, 'max': float(t.item())}}))
dist.destroy_process_group()
'''


@pytest.mark.parametrize('mode', ['no_device_plane', 'one_rank_fails', 'forced'])
def test_control_plane_falls_back_to_gloo_on_every_rank(tmp_path, mode):
    """bench.py's N > 1 process group (init_control_plane): RCCL is tried as a second group and adopted only if EVERY rank
    got through; here no rank can (no GPU) / one rank is made to fail / the attempt is skipped -- the barriers and the MAX
    over ranks must work over gloo on all ranks either way, and every rank must report the same plane."""
    script = tmp_path / 'rank.py'
    script.write_text(_CONTROL_PLANE_RANK.format(repo=REPO))
    env = _env()
    if mode == 'one_rank_fails':
        env['FAIL_ON_RANK'] = '1'
    if mode == 'forced':
        env['FORCE_GLOO'] = '1'
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr', '127.0.0.1', '--nnodes=1',
                        '--nproc-per-node', '2', str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert out['max'] == 2.0
    assert [p[0] for p in out['planes']] == ['gloo', 'gloo'] and all(p[1] for p in out['planes'])
    assert all(p[2] for p in out['planes'])                 # each rank says why RCCL is not carrying the barriers


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


def test_cpu_baseline_thread_sweep_stops_past_the_optimum_and_inside_the_budget():
    """bench.thread_sweep: the thread-count sweep of the CPU baseline (north_star: the CPU path "timed on the same box's host
    cores in the same run") must not be what an N > 1 line dies of.  Round 6: on a 256-thread host one image at 256 threads
    took 117 s and the two-rank launch ran into its timeout.  With the timings of that host the sweep stops at 64 threads
    (1.5x off the best); with a slow host it stops when the budget is nearly used; and it always tries two counts."""
    import importlib
    bench = importlib.import_module('bench')
    host = {8: 1.02, 16: 0.87, 32: 0.99, 64: 1.58, 128: 3.3, 256: 117.0}
    tried = []
    now = [0.0]

    def time_at(c):
        tried.append(c)
        now[0] += host[c]
        return host[c]

    sweep = bench.thread_sweep(256, time_at, 30.0, clock=lambda: now[0])
    assert tried == [8, 16, 32, 64] and min(sweep, key=sweep.get) == 16, (tried, sweep)
    # a host where every count takes 9 s: after two counts 18 s + 2 x 9 s > 0.7 x 30 s -> no third count
    tried.clear()
    now[0] = 0.0
    slow = lambda c: (tried.append(c), now.__setitem__(0, now[0] + 9.0), 9.0)[2]
    sweep = bench.thread_sweep(128, slow, 30.0, clock=lambda: now[0])
    assert tried == [8, 16], tried
    # a small host: its only count
    tried.clear()
    assert list(bench.thread_sweep(4, lambda c: (tried.append(c), 1.0)[1], 30.0, clock=lambda: 0.0)) == [4]
