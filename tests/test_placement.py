"""
CPU / NUMA placement of the per-GPU workers (megadetector_amd/placement.py; SURVEY.md 8(e) "scaling limiter",
reference notebooks/manage_local_batch.py:619-621 pins by process): the policy on described topologies, and a
world_size-2 gloo run in which every rank pins itself and rank 0 checks that the masks are disjoint.
"""

import os
import socket
import sys

import pytest

from conftest import REPO
from megadetector_amd import placement as P


def test_parse_cpulist():
    assert P.parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]
    assert P.parse_cpulist('') == []


def _mi355x_host():
    """2 sockets x 64 cores x 2 threads, 8 GPUs, 4 per socket (what the bench box looks like)"""
    node0 = list(range(0, 64)) + list(range(128, 192))
    node1 = list(range(64, 128)) + list(range(192, 256))
    return {'gpu_node': [0, 0, 0, 0, 1, 1, 1, 1], 'node_cpus': {0: node0, 1: node1}, 'allowed': list(range(256))}


def test_plan_on_a_two_socket_eight_gpu_host():
    topo = _mi355x_host()
    plan = P.plan(8, topo)
    assert len(plan) == 8 and all(len(p) == 32 for p in plan)
    flat = [c for p in plan for c in p]
    assert len(flat) == len(set(flat)) == 256                                  # disjoint, nothing wasted
    for g, cpus in enumerate(plan):
        assert set(cpus) <= set(topo['node_cpus'][topo['gpu_node'][g]])        # node-local
    # 2 of the 8 GPUs in use: each still stays on its own node and takes the whole node
    two = P.plan(2, {'gpu_node': [0, 1], 'node_cpus': topo['node_cpus'], 'allowed': topo['allowed']})
    assert set(two[0]) == set(topo['node_cpus'][0]) and set(two[1]) == set(topo['node_cpus'][1])


def test_plan_respects_the_allowed_set_and_unknown_nodes():
    topo = _mi355x_host()
    topo['allowed'] = list(range(0, 16)) + list(range(64, 80))               # a cgroup-limited container
    plan = P.plan(8, topo)
    assert sorted(c for p in plan for c in p) == topo['allowed']
    assert all(len(p) == 4 for p in plan)
    # no NUMA information at all (numa_node = -1 everywhere): even split of the allowed CPUs
    flat = {'gpu_node': [-1] * 4, 'node_cpus': {}, 'allowed': list(range(10))}
    plan = P.plan(4, flat)
    assert [len(p) for p in plan] == [2, 3, 2, 3] and sorted(c for p in plan for c in p) == list(range(10))
    # mixed: GPU 1's node is unknown -> it gets what GPU 0's node left
    mixed = {'gpu_node': [0, -1], 'node_cpus': {0: [0, 1, 2, 3]}, 'allowed': list(range(8))}
    plan = P.plan(2, mixed)
    assert plan[0] == [0, 1, 2, 3] and plan[1] == [4, 5, 6, 7]
    # fewer CPUs than workers: nobody is pinned
    assert P.plan(4, {'gpu_node': [-1] * 4, 'node_cpus': {}, 'allowed': [0, 1]}) == [[0, 1]] * 4


def test_loader_workers_are_sized_to_the_workers_cpus():
    assert P.loader_workers_for(16, 32) == 16
    assert P.loader_workers_for(16, 8) == 7
    assert P.loader_workers_for(4, 1) == 1


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _pin_worker(rank, world, port, out_path):
    sys.path.insert(0, REPO)
    import json
    import torch.distributed as dist
    from megadetector_amd import placement
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    before = sorted(os.sched_getaffinity(0))
    cpus = placement.pin_worker(rank, world, verbose=False)
    mine = {'rank': rank, 'before': before, 'cpus': sorted(cpus), 'mask': sorted(os.sched_getaffinity(0))}
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(mine, gathered, dst=0)
    if rank == 0:
        with open(out_path, 'w') as f:
            json.dump(gathered, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(not hasattr(os, 'sched_setaffinity') or len(os.sched_getaffinity(0)) < 2,
                    reason='needs at least two CPUs to hand out disjoint masks')
def test_two_ranks_pin_themselves_to_disjoint_cpu_sets(tmp_path):
    import json
    import torch.multiprocessing as mp
    out = str(tmp_path / 'masks.json')
    mp.spawn(_pin_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = json.load(open(out))
    assert [g['rank'] for g in got] == [0, 1]
    a, b = set(got[0]['mask']), set(got[1]['mask'])
    assert a and b and not (a & b)                                             # disjoint
    assert a | b <= set(got[0]['before'])                                      # inside what the process was allowed
    assert got[0]['mask'] == got[0]['cpus'] and got[1]['mask'] == got[1]['cpus']
