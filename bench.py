#!/usr/bin/env python3
"""
bench.py -- images/s of the MDv5a batch-inference hot path on MI355X.

One "step" = one pass of the whole hot path over one batch of synthetic input that is already
resident in HBM:  letterbox/normalise (HIP) -> YOLOv5x6 conv stack (HIP, MFMA bf16) -> Detect
decode (HIP) -> per-image NMS (HIP) -> D2H of <=300 boxes/image -> host formatting to
MegaDetector detection dicts (the vectorised replacement of reference
pytorch_detector.py:1361-1422).  Workload = BASELINE.json configs[1]: MDv5a bf16, 1280 px
letterbox, batch 32 on one MI355X; with --gpus N every rank runs the same per-GPU workload on
its own GPU (the image queue shards embarrassingly, no collectives: SURVEY.md section 8(e)).

Launch (N > 1):  python -m torch.distributed.run --nnodes=1 --nproc-per-node N
                 --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W
"""

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_BF16_TFLOPS = 2500.0          # dense MFMA bf16 peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_FP8_TFLOPS = 5000.0           # dense MFMA fp8 peak (block-scaled K = 128), same guide
PEAK_HBM_GBPS = 8000.0             # HBM3E peak (vendor), same guide; measured copy rate 4.75-6.3 TB/s
GFLOP_PER_IMAGE_1280 = 831.64      # 415.82 GMAC over 163 convs
# SURVEY.md section 8(d), algorithmic bytes per image at 1280x1280 (NOT the padded figures of the implementation):
PRE_BYTES_PER_IMAGE = 14.75e6      # 4.92 MB uint8 read + 9.83 MB 16-bit NCHW-equivalent written
NMS_BYTES_PER_IMAGE = 3.26e6 + 7.2e3   # 102000 x 8 fp32 read + <= 300 x 6 fp32 written
DECODE_BYTES_PER_IMAGE = 2 * 3.26e6    # 102000 x 8 fp32 logits read + 102000 x 8 fp32 predictions written
# What each storage type is, against the reference's own definition of "same results" (md_tests.py:96-100,418-531:
# compare_detection_lists at |d conf| 0.005 / |d coord| 0.001) on the sparse x6 checkpoint fixture: enforced for fp16
# (tests/test_gpu_precision_x6.py::test_sparse_fixture_*), an expected failure for bf16 / fp8 (same test); the per-tensor study
# profiles/r6_bf16_storage_study.txt shows that no head-sized set of fp16 tensors brings bf16 inside on every conditioning.
PRECISION = {
    'fp16': {'parity_backed': True, 'mode': "the detector's default storage type",
             'sparse_fixture_list_level': {'conf': 0.0010, 'coord': 0.0004, 'bars': [0.005, 0.0018], 'enforced': True}},
    'bf16': {'parity_backed': False, 'mode': 'throughput storage type (BASELINE.json configs[1] names it): reduced precision',
             'sparse_fixture_list_level': {'conf': 0.203, 'coord': 0.0496, 'bars': [0.005, 0.0018], 'enforced': False,
                                           'max_abs_dconf_all_anchors': 0.031, 'detections_ours_vs_reference': [27, 16, 27, 20]},
             'parity_backed_alternative': '--dtype fp16 (extra_configs.fp16_default in the default run)'},
    'fp8': {'parity_backed': False, 'mode': 'throughput mode (BASELINE.json configs[4]): e4m3 bottleneck 3x3 convs over bf16 storage',
            'sparse_fixture_list_level': {'max_abs_dconf_all_anchors': 0.087, 'bars': [0.005, 0.0018], 'enforced': False},
            'parity_backed_alternative': '--dtype fp16'},
}
ACT_GB_PER_IMAGE = 1.99            # unfused activation traffic (every conv reads its input once, writes its output once)
WEIGHT_GB = 0.28


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=None,
                    help='ranks = GPUs of this node.  Default: WORLD_SIZE when a launcher set it, else 1; given explicitly it '
                         'must agree with the launcher')
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--size', type=int, default=1280)
    ap.add_argument('--model', default='YOLOV5X6_MD')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp16', 'fp8'],
                    help='storage type of activations and weights (bf16 = the BASELINE.json configuration configs[1]; fp8 = '
                         'configs[4]: bf16 storage with the bottleneck 3x3 convs on e4m3 operands, usually with --batch 64)')
    ap.add_argument('--no-table', action='store_true', help='ignore megadetector_amd/tuned_cfgs.json (heuristic tiles)')
    ap.add_argument('--src', default=None,
                    help='HxW of the source images (e.g. 1536x2048): the real-shape variant of SURVEY.md 8(d), '
                         'the letterbox kernel resizes to the --size long side; not the headline configuration')
    ap.add_argument('--threshold', type=float, default=1e-5,
                    help='detection threshold handed to NMS (batch mode default of the reference)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--host-fed', action='store_true',
                    help='inputs start in pinned host memory and cross PCIe inside the timed region (H2D on a copy '
                         'stream, overlapped with the previous step); reported as the PCIe-inclusive rate, NOT the headline value')
    ap.add_argument('--graph', default='off', choices=['auto', 'on', 'off'],
                    help="graph replay of the forward (mdhip_set_graph; 'auto' = batches <= 8).  Off like the detector's default: "
                         "measured, no gain (profiles/r3_graph_replay.txt)")
    ap.add_argument('--lean', action='store_true',
                    help='only warm-up + timed steps (no per-stage / per-op extras): for rocprofv3 runs')
    ap.add_argument('--cpu-seconds', type=float, default=24.0)
    ap.add_argument('--no-cpu-single-thread', action='store_true', help='skip the one-thread CPU row (one image, ~1 min)')
    ap.add_argument('--nms-inline', action='store_true',
                    help='NMS + D2H of step i on the compute stream behind the forward (rounds 2-3 default).  Default since '
                         'round 4: on their own stream next to the forward of step i+1 -- with the banded NMS sort the 32 '
                         'workgroups are gone before the next forward reaches its 8-wave kernels (+0.4 .. 0.7 %, '
                         'profiles/r4_bench_nms_stream.txt)')
    ap.add_argument('--nms-own-stream', action='store_true', help='(the default now; kept so that older command lines still parse)')
    ap.add_argument('--pre-own-stream', action='store_true',
                    help='letterbox of step i+1 on its own stream next to the forward of step i, behind that forward\'s stem '
                         '(mdhip_preprocess waits for it inside the library).  Measured in round 4: 1015.5 against 1021.5 '
                         'images/s with the letterbox on the compute stream (profiles/r4_bench_pre_stream.txt) -- its '
                         'workgroups keep the 8-wave conv workgroups off their CUs; off by default')
    ap.add_argument('--profile-out', default=None, help='write per-op timings (json) here')
    ap.add_argument('--no-extra-configs', action='store_true',
                    help='skip the short legs of the other BASELINE configurations (extra_configs: fp8 batch 64, 1080p video '
                         'frames, 4:3 real shape, fp16 storage); they only run with the default headline workload on one GPU')
    ap.add_argument('--extra-steps', type=int, default=10)
    ap.add_argument('--pin-cpus', action='store_true',
                    help='N = 1: pin this process to the CPUs of the GPU\'s NUMA node anyway (placement.pin_worker(force=True); '
                         'with N > 1 every rank is pinned by default, MDHIP_NO_PINNING=1 switches that off)')
    ap.add_argument('--rendezvous-check', action='store_true',
                    help='launch-contract check without a GPU: every rank joins a gloo group, rank 0 prints {"n_gpus": world, '
                         '"ranks": [...]} and exits (tests/test_bench_launch.py runs `bench.py --gpus 2 --rendezvous-check`)')
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: become the launcher the driver would be --
    one rank per GPU through torch.distributed.run on 127.0.0.1 (the reference's recipe is one command per GPU,
    notebooks/manage_local_batch.py:617-621) -- and hand its exit code on.  Rank 0 of that run prints the one JSON line.
    The rendezvous port is the launcher's own choice (`--standalone`: its c10d store binds port 0), not a port probed here
    and released again -- another process could take such a port before the launcher binds it."""
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr', '127.0.0.1', '--nnodes=1',
           '--nproc-per-node', str(args.gpus), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    print('bench.py: --gpus {} without WORLD_SIZE: launching {}'.format(args.gpus, ' '.join(cmd)), file=sys.stderr)
    return subprocess.call(cmd, env=env)


def init_control_plane(torch, rank, world, local_rank, force_gloo, timeout_s=300, nccl_timeout_s=120):
    """The process group of an N > 1 run.  The path has no data-path collective: the group only carries
    the barriers around the timed region and the MAX over ranks.  A gloo group on the host is created FIRST and always --
    it cannot fail for a reason that has to do with the GPUs -- then RCCL (backend 'nccl', one rank per GPU) is tried as
    a second group: created, exercised with one all-reduce, and adopted only if EVERY rank got through (agreed over the
    gloo group).  Returns (dist, group, on_host, name, why): `group` carries barriers / reductions (None = the default gloo
    group), `on_host` says where its tensors live, `name` ('rccl' | 'gloo') goes into the JSON line, with `why` when RCCL
    was not used."""
    import datetime
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
    if force_gloo:
        return dist, None, True, 'gloo', 'asked for (MDHIP_BENCH_BACKEND=gloo or the one-GPU test hook)'
    ok, why, group = 1, '', None
    try:
        group = dist.new_group(backend='nccl', timeout=datetime.timedelta(seconds=nccl_timeout_s),
                               device_id=torch.device('cuda', local_rank))
        t = torch.ones(1, device='cuda')
        dist.all_reduce(t, group=group)
        torch.cuda.synchronize()
        if int(t.item()) != world:
            raise RuntimeError('RCCL all-reduce returned {} for a world of {}'.format(int(t.item()), world))
    except Exception as e:                                   # RCCL missing / refusing this box: the host plane carries on
        ok, why = 0, '{}: {}'.format(type(e).__name__, str(e).splitlines()[0][:200] if str(e) else '')
    flag = torch.tensor([ok], dtype=torch.int64)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)               # over gloo: every rank or none
    if int(flag.item()) == 1:
        return dist, group, False, 'rccl', ''
    if ok:
        why = 'another rank could not initialise RCCL'
    print('rank {}: RCCL control plane not available ({}): barriers over gloo'.format(rank, why), file=sys.stderr)
    return dist, None, True, 'gloo', why


def thread_sweep(cores, time_at, budget_s, clock=time.perf_counter):
    """{thread count: seconds for one image} over {8, 16, 32, 64, cores} in rising order, `time_at(count)` doing the timing.
    Stops (a) before a count when the counts tried so far already used most of the budget -- allowing for the next one to
    take twice the last -- and (b) after the first count that is 1.5x slower than the best so far: past the optimum it only
    gets worse, and fast (one image on all 256 hardware threads of a bench host took 117 s in round 6 and cost the N = 2
    line its timeout).  At least two counts are tried when there are two."""
    counts = sorted(set(c for c in (8, 16, 32, 64) if c < cores) | {cores})
    t_all = clock()
    sweep = {}
    for c in counts:
        if len(sweep) >= 2 and clock() - t_all + 2 * sweep[max(sweep)] > 0.7 * budget_s:
            break
        sweep[c] = time_at(c)
        if len(sweep) >= 2 and sweep[c] > 1.5 * min(sweep.values()):
            break
    return sweep


def cpu_baseline(weights, size, threshold, budget_s, single_thread=True):
    """The oracle (CPU restatement of the reference's path) timed on this host's cores: the thread count that does best
    on this box (a sweep of one image each over {8, 16, 32, 64, all}, stopped at the first count 1.5x slower than the best:
    batch-1 convs on all threads of the bench host are oversubscribed), then the rest of the budget at that count; forward and NMS + formatting seconds apart."""
    import torch
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import parity_util as PU
    from oracle import yolov5 as Y
    cores = torch.get_num_threads()
    fw = Y.Forward(weights.yaml, weights.torch_state(), emulate_bf16=False)
    imgs = PU.random_images(2, size, size, seed=100)

    def one(img, split=None):
        t0 = time.perf_counter()
        x, infos = PU.oracle_input([img], size, weights.max_stride)
        with torch.no_grad():
            pred = fw(x)
        t1 = time.perf_counter()
        out = PU.oracle_detections(pred, infos, tuple(x.shape[2:]), threshold)
        if split is not None:
            split[0] += t1 - t0
            split[1] += time.perf_counter() - t1
        return out

    small = PU.random_images(1, 256, 256, seed=1)[0]
    one(small)                                   # warm-up (thread pools, allocator)
    t_all = time.perf_counter()

    def time_at(c):
        torch.set_num_threads(c)
        one(small)
        t0 = time.perf_counter()
        one(imgs[0])
        return time.perf_counter() - t0

    sweep = thread_sweep(cores, time_at, budget_s)
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    split = [0.0, 0.0]
    t0 = time.perf_counter()
    n = 0
    while True:
        one(imgs[n % 2], split)
        n += 1
        el = time.perf_counter() - t0
        if el >= max(0.4 * budget_s, budget_s - (t0 - t_all)) or n >= 16:
            break
    torch.set_num_threads(cores)
    res = {'value': n / el, 'unit': 'images/s', 'cores': int(best), 'kind': 'port',
           'host_threads_available': int(cores),
           'thread_sweep_images_per_s': {str(c): round(1.0 / t, 4) for c, t in sweep.items()},
           'forward_s_per_image': round(split[0] / n, 3), 'nms_format_s_per_image': round(split[1] / n, 3),
           'sample': '{} synthetic {}x{} images, batch 1 (CPU batch size is forced to 1 by the '
                     'reference), oracle fp32 torch-CPU forward + NMS + formatting, {:.1f} s at {} threads (the best of a '
                     'one-image sweep over {} threads)'.format(n, size, size, el, best, '/'.join(str(c) for c in sweep))}
    if single_thread:
        # one thread, comparable to the published single-core figures
        # (reference megadetector.md:358-359: 0.5-0.8 images/s per core class); ONE image bounds the cost
        torch.set_num_threads(1)
        try:
            t0 = time.perf_counter()
            one(imgs[0])
            el1 = time.perf_counter() - t0
        finally:
            torch.set_num_threads(cores)
        res['single_thread'] = {'value': 1.0 / el1, 'unit': 'images/s', 'cores': 1, 'kind': 'port',
                                'sample': '1 synthetic {}x{} image, torch.set_num_threads(1), {:.1f} s'.format(size, size, el1)}
    return res


class Workload:
    """One bench workload on one GPU: a context, synthetic uint8 batches resident in HBM, and the software pipeline of a
    step (letterbox -> conv stack -> decode -> NMS -> D2H -> host formatting of the previous step's detections)."""

    RING = 16

    def __init__(self, torch, weights, dtype, B, S, src, threshold, device, seed_base=0, n_batches=8, host_fed=False,
                 nms_own_stream=True, pre_own_stream=False, graph='off', no_table=False):
        from megadetector_amd.hip_backend import HipContext
        from megadetector_amd.postprocess import letterbox_geometry
        self.torch, self.B, self.S, self.threshold, self.dtype = torch, B, S, threshold, dtype
        self.host_fed, self.nms_own_stream = host_fed, nms_own_stream
        H0, W0 = (int(v) for v in src.lower().split('x')) if src else (S, S)
        lb = letterbox_geometry((H0, W0), new_shape=S, stride=64, auto=True, scaleup=True)
        self.H0, self.W0 = H0, W0
        self.Hn, self.Wn = lb['out_hw']             # network input (letterboxed) shape
        self.ctx = ctx = HipContext(weights, device=device, dtype=dtype, max_batch=B, max_h=S, max_w=S)
        ctx.set_graph(graph)
        # measured tile choices (tools/autotune.py -> megadetector_amd/tuned_cfgs.json) are loaded by HipContext
        if no_table:
            ctx.lib.mdhip_set_tuned(ctx.h, None, 0)
        # synthetic uint8 RGB batches, resident in HBM before the timed region (K >= 8 distinct, cycled)
        self.n_batches = n_batches
        gen = torch.Generator(device='cuda')
        batches = []
        for i in range(n_batches):
            gen.manual_seed(seed_base + i)
            batches.append(torch.randint(0, 256, (B, H0, W0, 3), dtype=torch.uint8, device='cuda', generator=gen))
        self.geoms = [(H0, W0, lb['new_unpad'][1], lb['new_unpad'][0], lb['top'], lb['left'])] * B
        self.ptr_lists = [[int(b[i].data_ptr()) for i in range(B)] for b in batches]
        # everything is enqueued on one non-blocking stream (the legacy null stream synchronises implicitly
        # with every other blocking stream and measured ~1.5 ms per step slower)
        self.comp_s = torch.cuda.Stream()
        self.compute_stream = self.comp_s.cuda_stream
        # NMS + D2H of step i on their own stream, next to the forward of step i+1 (the library alternates between two
        # prediction buffers); ordered with events.  --nms-inline puts them back on the compute stream
        self.nms_s = torch.cuda.Stream()
        # (--pre-own-stream: the letterbox of step i + 1 on its own stream too, next to the forward of step i: the library makes
        # it wait for that forward's stem -- the only reader of the network input -- and the forward of step i + 1 waits for
        # it here.  Measured slower than in line: off by default)
        self.pre_s = torch.cuda.Stream()
        self.pre_own_stream = pre_own_stream and not host_fed
        self.pre_done = {}
        self.fwd_done = [torch.cuda.Event() for _ in range(4)]
        self.nms_done = [None] * 4
        # live per-stage timing (roofline.stages): event pairs on the stream each stage is launched on
        self.ev_pre = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(self.RING)]
        self.ev_nms = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(self.RING)]
        self.stage_live = {'on': False, 'steps': []}
        self.dev_ptrs = None
        if host_fed:
            # PCIe-inclusive variant: the batches live in pinned host memory; every step copies its batch
            # into one of two device buffers on a copy stream while the previous step computes
            self.host_batches = [b.cpu().pin_memory() for b in batches]
            self.dev_in = [torch.empty_like(batches[0]) for _ in range(2)]
            self.dev_ptrs = [[int(d[i].data_ptr()) for i in range(B)] for d in self.dev_in]
            self.copy_s = torch.cuda.Stream()
            self.copied = [torch.cuda.Event() for _ in range(2)]
            self.consumed = [torch.cuda.Event() for _ in range(2)]
            batches = None
        self.batches = batches
        torch.cuda.synchronize()

    def prepare(self):
        """fp8 mode: static activation scales from the first synthetic batch (mdhip_calibrate), before anything is timed"""
        if self.dtype != 'fp8':
            return
        ctx = self.ctx
        if self.host_fed:
            self.dev_in[0].copy_(self.host_batches[0])
            self.torch.cuda.synchronize()
            ctx.preprocess(self.dev_ptrs[0], self.geoms, self.Hn, self.Wn, stream=self.compute_stream)
        else:
            ctx.preprocess(self.ptr_lists[0], self.geoms, self.Hn, self.Wn, stream=self.compute_stream)
        ctx.calibrate(self.B, self.Hn, self.Wn, stream=self.compute_stream)

    # Software pipeline: the GPU work of step i (preprocess -> forward -> NMS -> D2H into a pinned
    # slot) is enqueued asynchronously, then the host formats the detections of step i-1 while the
    # GPU runs step i.  Every step's results are fully formatted inside the timed region.
    def forward_and_nms(self, i):
        torch, ctx, comp_s = self.torch, self.ctx, self.comp_s
        k = i % 4
        if self.nms_done[(i - 2) % 4] is not None:
            comp_s.wait_event(self.nms_done[(i - 2) % 4])      # the prediction buffer this forward overwrites has been consumed
        ctx.forward(self.B, self.Hn, self.Wn, stream=self.compute_stream)
        ns = self.nms_s if self.nms_own_stream else comp_s
        if self.nms_own_stream:
            self.fwd_done[k].record(comp_s)
            self.nms_s.wait_event(self.fwd_done[k])
        if self.stage_live['on']:
            self.ev_nms[i % self.RING][0].record(ns)
        ctx.nms_enqueue(self.B, self.threshold, 0.45, 300, slot=k, stream=ns.cuda_stream)
        if self.stage_live['on']:
            self.ev_nms[i % self.RING][1].record(ns)
        ev = torch.cuda.Event()
        ev.record(ns)
        self.nms_done[k] = ev

    def enqueue(self, i):
        torch, ctx, comp_s = self.torch, self.ctx, self.comp_s
        if self.host_fed:
            k = i % 2
            with torch.cuda.stream(self.copy_s):
                if i >= 2:
                    self.copy_s.wait_event(self.consumed[k])          # the letterbox kernel of step i-2 has read this buffer
                self.dev_in[k].copy_(self.host_batches[i % self.n_batches], non_blocking=True)
                self.copied[k].record(self.copy_s)
            comp_s.wait_event(self.copied[k])
            ctx.preprocess(self.dev_ptrs[k], self.geoms, self.Hn, self.Wn, stream=self.compute_stream)
            self.consumed[k].record(comp_s)
            self.forward_and_nms(i)
            return
        if not self.pre_own_stream:
            self.do_preprocess(i, comp_s)
            self.forward_and_nms(i)
            return
        if i not in self.pre_done:                      # the first step of a run(): nothing enqueued it yet
            self.do_preprocess(i, self.pre_s)
        comp_s.wait_event(self.pre_done.pop(i))
        self.forward_and_nms(i)
        if i + 1 < self.run_steps:                      # exactly K letterbox launches for K steps
            self.do_preprocess(i + 1, self.pre_s)

    def do_preprocess(self, i, stream):
        torch = self.torch
        if self.stage_live['on']:
            self.ev_pre[i % self.RING][0].record(stream)
        self.ctx.preprocess(self.ptr_lists[i % self.n_batches], self.geoms, self.Hn, self.Wn, stream=stream.cuda_stream)
        if self.stage_live['on']:
            self.ev_pre[i % self.RING][1].record(stream)
            self.stage_live['steps'].append(i)
        ev = torch.cuda.Event()
        ev.record(stream)
        self.pre_done[i] = ev

    def collect(self, i):
        from megadetector_amd.postprocess import format_detections
        det, counts = self.ctx.nms_wait(slot=i % 4)
        return [format_detections(det[b, :counts[b]], (self.Hn, self.Wn), (self.H0, self.W0, 3), (self.H0, self.W0, 3),
                                  self.threshold) for b in range(self.B)]

    def run(self, n_steps):
        # two steps are kept queued on the GPU behind the running one, so that a slow moment of the host
        # thread (formatting, a descheduled process) does not leave the GPU idle
        last = None
        depth = 2
        self.run_steps = n_steps
        self.pre_done = {}
        for i in range(n_steps):
            self.enqueue(i)
            if i >= depth:
                last = self.collect(i - depth)
        for i in range(max(0, n_steps - depth), n_steps):
            last = self.collect(i)
        return last

    def conv_roofline(self, fwd_ms):
        """(achieved TFLOP/s, peak TFLOP/s, share of the FLOPs on e4m3 operands) of the conv stack for a forward of
        fwd_ms: algorithmic conv FLOPs of the ops of the last forward; every launch priced at its own MFMA peak"""
        ctx = self.ctx
        conv = [o for o in ctx.op_infos() if o['kind'] == 0]
        flops = sum(o['flops'] for o in conv)
        op_peak = lambda o: PEAK_FP8_TFLOPS if ctx.conv_cfg_name(o['cfg']).startswith('f8:') else PEAK_BF16_TFLOPS
        peak = flops / sum(o['flops'] / op_peak(o) for o in conv)
        f8 = sum(o['flops'] for o in conv if op_peak(o) == PEAK_FP8_TFLOPS) / flops
        return flops / (fwd_ms * 1e-3) / 1e12, peak, f8, flops

    def close(self):
        self.torch.cuda.synchronize()
        self.ctx.close()
        self.batches = None


# bench.py's default invocation also measures, after the timed region of the headline workload (BASELINE configs[1]),
# every other BASELINE configuration that fits one GPU -- short legs, reported under "extra_configs", never `value`
EXTRA_LEGS = [
    # key,            dtype,  batch, source HxW,  what
    ('fp8_b64',       'fp8',  64, None,        'BASELINE configs[4]: fp8 (bottleneck 3x3 convs on e4m3 operands), 1280x1280, batch 64'),
    ('video_1080p',   'bf16', 32, '1080x1920', 'BASELINE configs[3] on one GPU: 1080x1920 video frames -> 768x1280 letterbox, batch 32'),
    ('real_4x3',      'bf16', 32, '1536x2048', 'SURVEY 8(d) real-shape variant: 1536x2048 camera-trap images -> 960x1280 letterbox, batch 32'),
    ('fp16_default',  'fp16', 32, None,        "fp16 storage (the detector's default dtype), 1280x1280, batch 32"),
]


def extra_leg(torch, weights, key, dtype, B, src, what, S, threshold, device, steps, warmup):
    t_start = time.perf_counter()
    wl = Workload(torch, weights, dtype, B, S, src, threshold, device, seed_base=7000, n_batches=2)
    try:
        wl.prepare()
        wl.run(2)
        wl.run(warmup)
        wl.ctx.time_forwards(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        wl.run(steps)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        fwd = wl.ctx.forward_times(min(steps, 64))
        wl.ctx.time_forwards(False)
        fwd_ms = float(np.mean(fwd))
        achieved, peak, f8, flops = wl.conv_roofline(fwd_ms)
        return {
            'workload': '{} ({}x{} letterbox, {:.2f} GFLOP/image, uint8 inputs resident in HBM, NMS threshold {})'.format(
                what, wl.Hn, wl.Wn, flops / B / 1e9, threshold),
            'value': round(B * steps / elapsed, 2), 'unit': 'images/s', 'ms_per_step': round(elapsed / steps * 1e3, 3),
            'steps': steps, 'warmup': warmup, 'dtype': dtype, 'batch': B,
            'roofline': {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                         'frac': round(achieved / peak, 4), 'kernel_ms_per_step': round(fwd_ms, 3),
                         'e4m3_flop_share': round(f8, 4)},
            'leg_seconds': round(time.perf_counter() - t_start, 2),
        }
    finally:
        wl.close()


def pmc_row_for_cfg(cfg_name):
    """MFMA-pipe utilisation and shader clock of a tile configuration from the newest committed counter summary
    (tools/pmc_bench.sh -> profiles/r*_pmc_bench_kernels.json); None when there is none.  A configuration runs as several
    template instantiations (channel tail, aligned mode): their rows are combined weighted by dispatches x duration, and
    listed one by one under `instantiations`."""
    import glob
    import re
    m = re.match(r'(v2|v5:run|v5:strip|v7:s2run|f8:run)?:?(\d+)x(\d+)/(\d+)x(\d+)', cfg_name)
    if not m:
        return None
    sym = {'v2': 'conv_v2_kernel', 'v5:run': 'conv_v5_kernel', 'f8:run': 'conv_f8_kernel', 'v7:s2run': 'conv_v7_kernel'}.get(m.group(1))
    if sym is None:
        return None
    if sym == 'conv_v7_kernel':
        want = 'conv_v7_kernel<'                                          # one tile shape: <channel-tail mode, aligned>
        keep = lambda k: True
    else:
        want = '{}<{}, {}, {}, {}, 0'.format(sym, *m.groups()[1:])        # (PROF = 0: not the developer variants)
        # conv_v2_kernel<BM, BN, WM, WN, PROF, upsample-reading, tail mode, activation ring stages>: a '/a3' configuration
        # is the three-stage ring instantiation, every other v2 name the two-stage one
        ring3 = cfg_name.endswith('/a3')
        keep = (lambda k: k.rstrip().endswith(', 3>') == ring3) if sym == 'conv_v2_kernel' else (lambda k: True)
    for ppath in sorted(glob.glob(os.path.join(REPO, 'profiles', 'r*_pmc_bench_kernels.json')), reverse=True):
        try:
            rows = {k: r for k, r in json.load(open(ppath)).get('kernels', {}).items() if want in k and keep(k)}
        except Exception:
            continue
        if not rows:
            continue
        src = 'profiles/' + os.path.basename(ppath)
        if len(rows) == 1 or not all('dispatches' in r for r in rows.values()):
            return dict(next(iter(rows.values())), source=src)
        w = {k: r['dispatches'] * r['dispatch_us'] for k, r in rows.items()}
        tot = sum(w.values())
        clk = [(w[k], r['shader_clock_ghz']) for k, r in rows.items() if r.get('shader_clock_ghz')]
        out = {'dispatches': sum(r['dispatches'] for r in rows.values()),
               'dispatch_us': tot / sum(r['dispatches'] for r in rows.values()),
               'mfma_util_2p4ghz': sum(w[k] * r['mfma_util_2p4ghz'] for k, r in rows.items()) / tot,
               'shader_clock_ghz': (sum(a * b for a, b in clk) / sum(a for a, _ in clk)) if clk else None,
               'instantiations': {k[k.index(sym):]: r for k, r in rows.items()}, 'source': src}
        if clk and all(r.get('mfma_util_measured_clock') for r in rows.values()):
            out['mfma_util_measured_clock'] = sum(w[k] * r['mfma_util_measured_clock'] for k, r in rows.items()) / tot
        return out
    return None


def main():
    args = parse_args()
    if os.environ.get('MDHIP_BENCH_DUMP_AFTER'):
        # a rank that is still running after this many seconds writes every thread's Python stack to stderr (and goes on):
        # what a hung multi-rank launch was waiting for, without a debugger on the box
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ['MDHIP_BENCH_DUMP_AFTER']), repeat=False, exit=False)
    gpus_given = args.gpus is not None
    if not gpus_given:                                        # `torchrun --nproc-per-node N bench.py` without --gpus: N ranks
        args.gpus = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if gpus_given and world != args.gpus:
        raise SystemExit('bench.py: --gpus {} but WORLD_SIZE={}: launch with --nproc-per-node {} (or without a launcher: '
                         '`python bench.py --gpus N` starts the ranks itself)'.format(args.gpus, world, args.gpus))
    if args.rendezvous_check:
        import torch
        import torch.distributed as dist
        got = [rank]
        if world > 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            dist.init_process_group('gloo', rank=rank, world_size=world)
            allr = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(allr, torch.tensor([rank], dtype=torch.int64))
            got = [int(t.item()) for t in allr]
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({'n_gpus': world, 'ranks': got, 'rendezvous_check': True}))
        return
    import torch
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: no HIP device visible (there is no CPU path to time)')
    # test hook: MDHIP_BENCH_ONE_GPU=1 runs every rank on device 0 with the gloo backend, so that the multi-rank code
    # path can be exercised on a one-GPU box (RCCL refuses two ranks on one device); never set by the driver
    # ('try': every rank on device 0 AND the RCCL attempt made -- RCCL refuses a duplicate device, which exercises the
    # fall-back to gloo with a real RCCL failure on a real GPU box: profiles/r6_bench_2rank_rccl_refused.json)
    hook = os.environ.get('MDHIP_BENCH_ONE_GPU', '0')
    one_gpu = hook == '1'
    if hook in ('1', 'try'):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # one rank per GPU: CPUs of the GPU's NUMA node, disjoint from the other ranks' (the host thread formats 2.8 ms of
    # detections per step and keeps the queue fed; SURVEY.md 8(e) "scaling limiter")
    from megadetector_amd import placement
    pinned_cpus = placement.pin_worker(local_rank, 1 if hook in ('1', 'try') else int(os.environ.get('LOCAL_WORLD_SIZE', world)), verbose=False,
                                       force=args.pin_cpus)
    if world > 1 or args.pin_cpus:                             # stdout carries the one JSON line and nothing else
        print('rank {}: {} CPUs{}'.format(rank, len(pinned_cpus), ' ({}..{})'.format(pinned_cpus[0], pinned_cpus[-1])
                                          if pinned_cpus else ''), file=sys.stderr)
    dist, group, control_plane, control_why = None, None, None, ''
    if world > 1:
        # RCCL when every rank can start it, gloo otherwise (init_control_plane); MDHIP_BENCH_BACKEND=gloo skips the attempt
        force_gloo = one_gpu or os.environ.get('MDHIP_BENCH_BACKEND', 'nccl') == 'gloo'
        dist, group, one_gpu, control_plane, control_why = init_control_plane(torch, rank, world, local_rank, force_gloo)
        # (from here on one_gpu means "control tensors live on the host")

    from megadetector_amd import weights_io, yolo_yaml
    from megadetector_amd.postprocess import format_detections

    B, S = args.batch, args.size
    yaml = getattr(yolo_yaml, args.model)
    weights = weights_io.synthetic_weights(yaml, seed=0)
    wl = Workload(torch, weights, args.dtype, B, S, args.src, args.threshold, local_rank, seed_base=1000 * rank,
                  host_fed=args.host_fed, nms_own_stream=not args.nms_inline, pre_own_stream=args.pre_own_stream, graph=args.graph,
                  no_table=args.no_table)
    ctx, run, stage_live = wl.ctx, wl.run, wl.stage_live
    H0, W0, Hn, Wn = wl.H0, wl.W0, wl.Hn, wl.Wn
    geoms, ptr_lists, compute_stream = wl.geoms, wl.ptr_lists, wl.compute_stream
    dev_ptrs = wl.dev_ptrs
    ev_pre, ev_nms, RING = wl.ev_pre, wl.ev_nms, wl.RING

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier(group=group)
        torch.cuda.synchronize()

    wl.prepare()                # fp8: static activation scales from the first synthetic batch (mdhip_calibrate), before anything is timed
    run(2)                      # initialisation (first-touch of every buffer and code path), not a warm-up step
    run(args.warmup)
    # live roofline measurement: a HIP event pair on the launch stream around the conv stack of every
    # timed step (mdhip_forward = the 152 implicit-GEMM launches + 11 small pool/upsample/decode kernels)
    ctx.time_forwards(True)
    # Host side: formatting a step creates ~10^4 dicts, which triggers the cyclic collector many times per step, and
    # every full collection walks the start-up heap (torch, numpy, the context): 2.5 ms per step measured.  Freezing
    # that heap (gc stays enabled) removes it; the batch driver does the same after loading the model.
    import gc
    gc.collect()
    gc.freeze()
    stage_live['on'] = not args.host_fed
    barrier()
    t0 = time.perf_counter()
    out = run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    stage_live['on'] = False
    my_elapsed = elapsed
    live_pre = [ev_pre[i % RING][0].elapsed_time(ev_pre[i % RING][1]) for i in stage_live['steps'][-RING:]]
    live_nms = [ev_nms[i % RING][0].elapsed_time(ev_nms[i % RING][1]) for i in stage_live['steps'][-RING:]]
    fwd_ms_live = ctx.forward_times(min(args.steps, 64))
    ctx.time_forwards(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if one_gpu else 'cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        elapsed = float(t.item())
        # per-rank rates (outside the timed region): exposes stragglers / NUMA effects on the first multi-GPU run
        allt = [torch.zeros(1, dtype=torch.float64, device='cpu' if one_gpu else 'cuda') for _ in range(world)]
        dist.all_gather(allt, torch.tensor([my_elapsed], dtype=torch.float64, device='cpu' if one_gpu else 'cuda'), group=group)
        per_rank = [round(B * args.steps / float(x.item()), 2) for x in allt]
    else:
        per_rank = [round(B * args.steps / my_elapsed, 2)]

    # ---- per-stage and per-kernel timing (outside the timed region) --------------------
    stages = {}
    roof = None
    if rank == 0:
        def timed(fn, reps=3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / reps * 1e3
        ms = np.zeros(ctx.num_ops(), dtype=np.float64)
        if not args.lean:
            stages['preprocess_ms'] = timed(lambda: ctx.preprocess(dev_ptrs[0] if args.host_fed else ptr_lists[0], geoms, Hn, Wn))
            stages['forward_ms'] = timed(lambda: ctx.forward(B, Hn, Wn))
            stages['nms_d2h_ms'] = timed(lambda: ctx.nms(B, args.threshold, 0.45, 300))
            det, cnt = ctx.nms(B, args.threshold, 0.45, 300)
            t = time.perf_counter()
            for b in range(B):
                format_detections(det[b, :cnt[b]], (Hn, Wn), (H0, W0, 3), (H0, W0, 3), args.threshold)
            stages['host_format_ms'] = (time.perf_counter() - t) * 1e3
            stages['mean_detections_per_image'] = float(np.mean(cnt))
            reps = 3
            for _ in range(reps):
                ms += ctx.forward_timed(B, Hn, Wn)
            ms /= reps
        infos = ctx.op_infos()
        conv = [(o, ms[o['op']]) for o in infos if o['kind'] == 0]
        conv_flops = sum(o['flops'] for o, _ in conv)
        # the MFMA peak a conv is priced against: 5 PFLOP/s for the launches on e4m3 operands, 2.5 for 16-bit ones;
        # the stack's peak is the rate at which it would finish with every launch at its own peak
        op_peak = lambda o: PEAK_FP8_TFLOPS if ctx.conv_cfg_name(o['cfg']).startswith('f8:') else PEAK_BF16_TFLOPS
        stack_peak = conv_flops / sum(o['flops'] / op_peak(o) for o, _ in conv)
        f8_share = sum(o['flops'] for o, _ in conv if op_peak(o) == PEAK_FP8_TFLOPS) / conv_flops
        conv_ms = sum(t for _, t in conv)
        other_ms = float(ms.sum() - conv_ms)
        # `achieved`: algorithmic conv FLOPs of one step / duration of the conv-stack launch sequence
        # measured live (HIP events, timed region).  That duration includes the 11 non-conv kernels
        # of the forward (`other_kernels_ms_per_step`, measured per op outside the timed region), so
        # the figure is slightly conservative; `per_op_conv_tflops` is the per-op-event figure.
        if conv_ms <= 0:
            conv_ms = float('nan')
        fwd_ms = float(np.mean(fwd_ms_live)) if len(fwd_ms_live) else float(ms.sum())
        achieved = conv_flops / (fwd_ms * 1e-3) / 1e12
        # HBM bytes per step from the PMC counters: two separate `rocprofv3 --pmc` passes of THIS command
        # (FETCH_SIZE, WRITE_SIZE; tools/gpu_session.sh traffic -> tools/hbm_traffic.py), committed under profiles/.
        # It is read from the newest committed measurement of this workload, not measured in this run (counters
        # cannot be collected from inside the process); `traffic_source` says which file.
        traffic, traffic_source = None, None
        if not args.src and args.dtype == 'bf16':
            import glob
            for tpath in sorted(glob.glob(os.path.join(REPO, 'profiles', 'r*_hbm_traffic.json')), reverse=True):
                try:
                    tj = json.load(open(tpath))
                    if tj.get('key') == '{}:{}:{}'.format(args.model, B, S):
                        traffic = tj.get('hbm_bytes_per_step')
                        traffic_source = 'profiles/{} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; ' \
                                         'FETCH_SIZE doubled per the guide\'s gfx950 note; not re-measured in this run)'.format(
                                             os.path.basename(tpath))
                        break
                except Exception:
                    pass
        algorithmic_bytes = (ACT_GB_PER_IMAGE * B + WEIGHT_GB) * 1e9 * (Hn * Wn) / (1280.0 * 1280.0)
        roof = {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': round(stack_peak, 1), 'unit': 'TFLOP/s',
                'frac': round(achieved / stack_peak, 4),
                'peak_note': '2500 (dense bf16 MFMA) for 16-bit launches, 5000 (dense fp8 MFMA) for launches on e4m3 operands: '
                             '{:.1f} % of the FLOPs of this step run on e4m3 operands'.format(100 * f8_share),
                'traffic': traffic, 'traffic_source': traffic_source,
                'algorithmic_bytes_per_step': algorithmic_bytes,
                'traffic_over_algorithmic': None if not traffic else round(traffic / algorithmic_bytes, 3),
                'kernel': 'conv stack of one step = {} conv launches (conv_igemm_kernel / conv_v2_kernel / conv_v7_kernel / '
                          'conv_v5_kernel / conv_v5s_kernel / conv_stem_kernel / conv_f8_kernel instantiations), HIP events on the launch stream around '
                          'mdhip_forward in the timed region, mean of {} steps'.format(len(conv), len(fwd_ms_live)),
                'flops_per_step': conv_flops, 'kernel_ms_per_step': round(fwd_ms, 3),
                'per_op_conv_ms_per_step': None if args.lean else round(conv_ms, 3),
                'per_op_conv_tflops': None if args.lean else round(conv_flops / (conv_ms * 1e-3) / 1e12, 2),
                'other_kernels_ms_per_step': None if args.lean else round(other_ms, 3)}
        if not args.lean:
            # the dominant kernel instantiation (by time) and its own rate, from the per-op events
            by_cfg = {}
            for o, t in conv:
                e = by_cfg.setdefault(o['cfg'], [0, 0.0, 0.0])
                e[0] += 1
                e[1] += t
                e[2] += o['flops']
            top = max(by_cfg.items(), key=lambda kv: kv[1][1])

            pmc_row = pmc_row_for_cfg
            roof['dominant_kernel'] = {
                'name': ctx.conv_cfg_name(top[0]), 'launches_per_step': top[1][0],
                'ms_per_step': round(top[1][1], 3), 'avg_launch_us': round(top[1][1] / top[1][0] * 1e3, 2),
                'flops_per_step': top[1][2], 'achieved_tflops': round(top[1][2] / (top[1][1] * 1e-3) / 1e12, 2),
                'peak': PEAK_FP8_TFLOPS if ctx.conv_cfg_name(top[0]).startswith('f8:') else PEAK_BF16_TFLOPS,
                'frac': round(top[1][2] / (top[1][1] * 1e-3) / 1e12 /
                              (PEAK_FP8_TFLOPS if ctx.conv_cfg_name(top[0]).startswith('f8:') else PEAK_BF16_TFLOPS), 4)}
            # `frac` above IS the MFMA-pipe utilisation against the 2.4 GHz the vendor peak assumes (live events); the
            # counter summary adds the utilisation against the clock the kernel really ran at
            roof['dominant_kernel']['mfma_util'] = roof['dominant_kernel']['frac']
            roof['dominant_kernel']['pmc'] = pmc_row(ctx.conv_cfg_name(top[0]))
            # the most matrix-bound instantiation: highest FLOP rate among those with >= 5 % of the conv time
            heavy = [kv for kv in by_cfg.items() if kv[1][1] >= 0.05 * conv_ms]
            best = max(heavy, key=lambda kv: kv[1][2] / kv[1][1])
            roof['fastest_heavy_kernel'] = {
                'name': ctx.conv_cfg_name(best[0]), 'launches_per_step': best[1][0], 'ms_per_step': round(best[1][1], 3),
                'achieved_tflops': round(best[1][2] / (best[1][1] * 1e-3) / 1e12, 2),
                'mfma_util': round(best[1][2] / (best[1][1] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4), 'pmc': pmc_row(ctx.conv_cfg_name(best[0]))}
        # per-stage HBM rooflines (north_star: "rocprof HBM GB/s for preprocess/NMS"): algorithmic bytes of SURVEY.md
        # section 8(d) / duration of the stage's kernels measured live (HIP events on the stream the stage is launched
        # on, timed region; the NMS pair also covers the D2H of the <= 300 x 6 results) -- decode from the per-op events
        def stage(bytes_per_image, ms_list, note):
            if not len(ms_list):
                return None
            ms_ = float(np.mean(ms_list))
            gbps = bytes_per_image * B / (ms_ * 1e-3) / 1e9
            return {'bound': 'hbm', 'bytes': bytes_per_image * B, 'ms': round(ms_, 4), 'achieved': round(gbps, 1),
                    'peak': PEAK_HBM_GBPS, 'unit': 'GB/s', 'frac': round(gbps / PEAK_HBM_GBPS, 4), 'how': note}
        scale_px = (Hn * Wn) / (1280.0 * 1280.0)
        src_bytes = H0 * W0 * 3 + Hn * Wn * 3 * 2           # uint8 source read + 16-bit network input written
        roof['stages'] = {
            'preprocess': stage(src_bytes, live_pre, 'letterbox kernel of the batch (streaming copy: no resampling; streaming bilinear: INTER_LINEAR; general otherwise), live events, {} steps'.format(len(live_pre))),
            'nms': stage(NMS_BYTES_PER_IMAGE * scale_px, live_nms,
                         'nms kernels + D2H of the results, live events on the NMS stream, {} steps'.format(len(live_nms))),
            'decode': None,
        }
        if not args.lean:
            dec_ops = [o for o in infos if o['kind'] == 3]
            folded = [o for o in dec_ops if o['cfg'] == -2]          # decoded in the epilogue of the conv in front (mdhip.h)
            if folded and len(folded) == len(dec_ops):
                # no decode launch exists: the stage IS the Detect 1x1 convs (read the level's feature map, write the fp32
                # predictions); bytes = those convs' algorithmic bytes, per batch
                det = [infos[o['op'] - 1] for o in dec_ops]
                roof['stages']['decode'] = stage(
                    sum(o['bytes'] for o in det) / B, [float(sum(ms[o['op']] for o in det))],
                    'Detect 1x1 conv + decode in one launch x{} levels (no detect_decode_kernel launch, no fp32 logits tensor): '
                    'feature map read + predictions written, per-op events outside the timed region'.format(len(det)))
            else:
                roof['stages']['decode'] = stage(
                    DECODE_BYTES_PER_IMAGE * scale_px, [float(sum(ms[o['op']] for o in dec_ops if o['cfg'] != -2))],
                    'detect_decode_kernel x{} levels, per-op events outside the timed region'.format(
                        sum(1 for o in dec_ops if o['cfg'] != -2)))
        if args.model == 'YOLOV5X6_MD' and (Hn, Wn) == (1280, 1280):
            assert abs(conv_flops / B / 1e9 - GFLOP_PER_IMAGE_1280) < 0.05, conv_flops / B / 1e9
        if args.profile_out and not args.lean:
            os.makedirs(os.path.dirname(os.path.abspath(args.profile_out)), exist_ok=True)
            with open(args.profile_out, 'w') as f:
                json.dump([dict(o, ms=float(ms[o['op']]),
                                tflops=(o['flops'] / (ms[o['op']] * 1e-3) / 1e12 if ms[o['op']] > 0 else 0.0),
                                gbps=(o['bytes'] / (ms[o['op']] * 1e-3) / 1e9 if ms[o['op']] > 0 else 0.0))
                           for o in infos], f, indent=1)

    extra = None
    default_workload = (args.dtype == 'bf16' and B == 32 and not args.src and not args.host_fed and args.model == 'YOLOV5X6_MD'
                        and S == 1280 and not args.no_table)
    if rank == 0 and world == 1 and default_workload and not args.lean and not args.no_extra_configs:
        wl.close()                                   # the headline context's arena goes back before the legs allocate theirs
        extra = {}
        for key, dt, b, src, what in EXTRA_LEGS:
            try:
                extra[key] = extra_leg(torch, weights, key, dt, b, src, what, S, args.threshold, local_rank,
                                       steps=args.extra_steps, warmup=3)
            except Exception as e:                   # a failing leg must not cost the headline line
                extra[key] = {'error': '{}: {}'.format(type(e).__name__, e)}

    # N > 1: the ranks part here -- the process group goes (nothing below is collective), every rank but 0 is done; rank 0
    # still owes the CPU baseline ("timed on the same box's host cores in the same run", BASELINE.json north_star), which
    # it runs AFTER the group is gone so that no rank waits in a barrier for 30 s of host work
    if dist is not None:
        dist.barrier(group=group)
        dist.destroy_process_group()
        dist = None
    wl.close()
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        if world > 1 and hasattr(os, 'sched_setaffinity'):
            # the rank was pinned to its GPU's share of the host; the baseline is the HOST's best, as in the N = 1 line
            try:
                os.sched_setaffinity(0, range(os.cpu_count() or 1))
                # (the launcher exported OMP_NUM_THREADS=1; never more than 128 threads: batch-1 convs on every hardware
                # thread of a 256-thread host are pathologically oversubscribed)
                torch.set_num_threads(max(1, min(128, len(os.sched_getaffinity(0)))))
            except OSError:
                pass
        cpu = cpu_baseline(weights, S, args.threshold, args.cpu_seconds, single_thread=not args.no_cpu_single_thread)
        if world > 1:
            cpu['note'] = 'rank 0, after the timed region and after the process group was destroyed (the other ranks have exited)'

    if rank == 0:
        total_images = world * B * args.steps
        line = {
            'metric': 'images/sec (whole node) MDv5a @1280px batch inference' +
                      (' [host-fed: PCIe-inclusive, not the headline value]' if args.host_fed else '') +
                      (' [real-shape variant {}x{} -> {}x{}, not the headline configuration]'.format(H0, W0, Hn, Wn)
                       if args.src else ''),
            'value': round(total_images / elapsed, 2),
            'unit': 'images/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': args.dtype,
            'precision': dict(PRECISION[args.dtype], evidence='tests/test_gpu_precision_x6.py (sparse fixture), '
                                                               'profiles/r6_bf16_storage_study.txt'),
            'data': 'synthetic',
            'config': {
                'workload': ('MDv5a topology (YOLOv5x6, nc=3, 163 convs, {3:.2f} GFLOP/image) {5}, '
                             '{0}x{4} letterbox, batch {1} per GPU, uint8 RGB inputs resident in HBM, seeded '
                             'synthetic weights (no checkpoint available offline), NMS threshold {2}').format(
                                 Hn, B, args.threshold, GFLOP_PER_IMAGE_1280 * Hn * Wn / (1280.0 * 1280.0), Wn, args.dtype),
                'parallelism': 'image queue sharded over {} GPU(s), one process per GPU, no collectives'.format(world),
            },
            'roofline': roof,
            'cpu_baseline': cpu,
            'stages': stages,
            'extra_configs': extra,
            'per_rank_images_per_s': per_rank,
            'pinned_cpus': len(pinned_cpus) if (world > 1 or args.pin_cpus) else None,
            'control_plane': control_plane,
        }
        if control_why:
            line['control_plane_note'] = control_why
        print(json.dumps(line))


if __name__ == '__main__':
    main()
