"""
CPU / NUMA placement of the per-GPU worker processes (SURVEY.md section 8(e) "scaling limiter": host memory
placement and -- for real images -- JPEG decode; the reference pins work by process, one CUDA_VISIBLE_DEVICES per
command, notebooks/manage_local_batch.py:546-547,619-621, and leaves the CPUs to the OS).

One process per GPU; each gets a CPU set that is
  * on the NUMA node its GPU hangs off (pinned staging buffers, the shared-memory ring and the formatting thread are
    then node-local to the PCIe root the copies go through), and
  * disjoint from the sets of the other GPU workers (8 x 16 loader processes on 128 hardware threads otherwise
    migrate over each other the first time 8 GPUs are fed from JPEGs).
Loader processes are spawned by the GPU worker and inherit its mask; their number is sized to it
(`loader_workers_for`).

The topology comes from sysfs (`/sys/bus/pci/devices/<bus id>/numa_node`, `/sys/devices/system/node/node<N>/cpulist`)
with the GPU's PCI bus id from the HIP runtime; when any of that is missing (containers, single-node hosts) the allowed
CPUs of the process are split evenly.  `plan` is a pure function of a topology description, so the policy is testable
without the hardware.
"""

import ctypes
import os


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    cpus = []
    for part in text.strip().split(','):
        part = part.strip()
        if not part:
            continue
        if '-' in part:
            a, b = part.split('-', 1)
            cpus.extend(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return cpus


def gpu_pci_bus_id(ordinal):
    """'0000:c1:00.0' of HIP device `ordinal`, or None (no HIP runtime / no such device).  Initialises HIP in the
    calling process: call it in the (spawned) GPU worker, not in a parent that is about to fork."""
    try:
        hip = ctypes.CDLL('libamdhip64.so')
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(ordinal)) != 0:
            return None
        return buf.value.decode().lower()
    except Exception:
        return None


def gpu_numa_node(ordinal, bus_id=None):
    """NUMA node of the GPU's PCIe root, or -1 when the platform does not say"""
    bus_id = bus_id or gpu_pci_bus_id(ordinal)
    if not bus_id:
        return -1
    try:
        with open('/sys/bus/pci/devices/{}/numa_node'.format(bus_id)) as f:
            return int(f.read().strip())
    except Exception:
        return -1


def node_cpus(node):
    try:
        with open('/sys/devices/system/node/node{}/cpulist'.format(int(node))) as f:
            return parse_cpulist(f.read())
    except Exception:
        return []


def read_topology(n_gpus):
    """{'gpu_node': [node of GPU 0 .. n-1], 'node_cpus': {node: [cpus]}, 'allowed': sorted allowed CPUs}"""
    gpu_node = [gpu_numa_node(g) for g in range(n_gpus)]
    nodes = sorted(set(n for n in gpu_node if n >= 0))
    return {'gpu_node': gpu_node, 'node_cpus': {n: node_cpus(n) for n in nodes},
            'allowed': sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else list(range(os.cpu_count() or 1))}


def _split(cpus, k):
    """k contiguous, near-equal, disjoint slices of `cpus` (hyper-thread siblings are usually numbered cpu and
    cpu + n_cores: a contiguous slice of each half would be better still, but needs the sibling lists)"""
    n = len(cpus)
    return [cpus[(i * n) // k:((i + 1) * n) // k] for i in range(k)]


def plan(n_gpus, topology):
    """
    CPU list for each of the n_gpus workers.  GPUs that share a NUMA node split that node's (allowed) CPUs; GPUs
    whose node is unknown split whatever the known nodes left over.  Falls back to an even split of the allowed set
    when a worker would end up without a CPU.  With fewer allowed CPUs than workers nobody is pinned.
    """
    allowed = list(topology.get('allowed') or [])
    if n_gpus <= 0:
        return []
    if len(allowed) < n_gpus:
        return [list(allowed) for _ in range(n_gpus)]
    allowed_set = set(allowed)
    gpu_node = list(topology.get('gpu_node') or [-1] * n_gpus)
    out = [None] * n_gpus
    taken = set()
    by_node = {}
    for g, node in enumerate(gpu_node[:n_gpus]):
        by_node.setdefault(node, []).append(g)
    for node, gpus in sorted(by_node.items()):
        if node < 0:
            continue
        cpus = [c for c in topology['node_cpus'].get(node, []) if c in allowed_set]
        if len(cpus) < len(gpus):
            by_node.setdefault(-1, []).extend(gpus)           # nothing usable on that node: treat as unknown
            continue
        for g, part in zip(gpus, _split(cpus, len(gpus))):
            out[g] = part
            taken.update(part)
    unknown = sorted(g for g in by_node.get(-1, []) if out[g] is None)
    if unknown:
        rest = [c for c in allowed if c not in taken]
        if len(rest) < len(unknown):
            return _split(allowed, n_gpus)
        for g, part in zip(unknown, _split(rest, len(unknown))):
            out[g] = part
    if any(not p for p in out):
        return _split(allowed, n_gpus)
    return out


def pin_worker(gpu, n_gpus, topology=None, verbose=True, force=False):
    """Restricts the calling process (and everything it spawns later) to its share of the CPUs; returns the CPU list.
    MDHIP_NO_PINNING=1 switches it off.  A lone worker (n_gpus <= 1) keeps the whole machine unless `force` (or
    MDHIP_PIN_SINGLE=1) asks for the CPUs of its GPU's NUMA node: `bench.py --pin-cpus`, the detector option
    'pin_cpus' and the image / video drivers' single-GPU runs on a two-socket host."""
    force = force or os.environ.get('MDHIP_PIN_SINGLE') == '1'
    if os.environ.get('MDHIP_NO_PINNING') == '1' or not hasattr(os, 'sched_setaffinity') or (n_gpus <= 1 and not force):
        return sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else []
    n_gpus = max(1, n_gpus)
    if topology is None:
        # a lone worker on GPU g still needs the node of GPU g: read the topology up to and including that ordinal
        topology = read_topology(max(n_gpus, gpu + 1))
        if n_gpus == 1:
            topology = dict(topology, gpu_node=[topology['gpu_node'][gpu]])
    cpus = plan(n_gpus, topology)[gpu if n_gpus > 1 else 0]
    try:
        # every thread of the process, not only the caller: sched_setaffinity(0, ...) pins the calling THREAD on Linux,
        # and the runtime threads that torch / numpy / the HIP runtime started while importing would keep the full mask
        tids = [0]
        try:
            tids += [int(t) for t in os.listdir('/proc/self/task')]
        except OSError:
            pass
        for tid in tids:
            try:
                os.sched_setaffinity(tid, cpus)
            except ProcessLookupError:
                pass                                   # a thread that ended in between
    except OSError as e:
        print('Warning: could not pin the worker of GPU {} to CPUs {}: {}'.format(gpu, cpus, e))
        return sorted(os.sched_getaffinity(0))
    if verbose:
        node = (topology.get('gpu_node') or [-1] * n_gpus)[gpu if n_gpus > 1 else 0]
        print('GPU {} worker pinned to {} CPUs (NUMA node {}): {}..{}'.format(gpu, len(cpus), node, cpus[0], cpus[-1]))
    return cpus


def loader_workers_for(requested, n_cpus):
    """loader processes for one GPU worker that owns n_cpus CPUs: never more than the CPUs it has left after the
    GPU-feeding main thread (the reference's default of 4, run_detector_batch.py:86, assumes the whole machine)"""
    return max(1, min(int(requested), max(1, int(n_cpus) - 1)))
