#!/usr/bin/env python3
"""
Measures every implicit-GEMM tile configuration on every conv op of a model at a given
(batch, size) on the GPU and records the fastest per GEMM shape.  Output (json):
  { "entries": [ {"m","n","k","ntaps","stride","has_res","cfg","batch","ms","tflops"}, ... ] }
      -> megadetector_amd/tuned_cfgs.json (loaded by HipContext, matched on the exact shape)
plus a human-readable table (per op: ms and TFLOP/s per configuration).

Run on the GPU box:  python tools/autotune.py --out gpurun_out/tuned_cfgs.json
"""

import argparse
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='YOLOV5X6_MD')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--size', type=int, default=1280)
    ap.add_argument('--hw', default=None, help='HxW of the letterboxed input when it is not square (e.g. 960x1280)')
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--reps', type=int, default=2)
    ap.add_argument('--out', default=os.path.join(REPO, 'gpurun_out', 'tuned_cfgs.json'))
    ap.add_argument('--table', default=None)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp16', 'fp8'])
    ap.add_argument('--family-from', default=None,
                    help='existing table (measured at the canonical batch size): only configurations of the same kernel '
                         'family (same fp32 summation order) as its entry for the layer are tried, so that an '
                         'image\'s result does not depend on the batch size it is processed at')
    ap.add_argument('--only', default=None,
                    help='comma-separated name prefixes (e.g. "v6:,v5:"): only these configurations are measured against '
                         'the configuration the current table selects (a short re-tune after adding a kernel family)')
    args = ap.parse_args()

    import torch
    from megadetector_amd import weights_io, yolo_yaml
    from megadetector_amd.hip_backend import HipContext
    B, S = args.batch, args.size
    HH, WW = (int(v) for v in args.hw.lower().split('x')) if args.hw else (S, S)
    S = max(HH, WW)
    W = weights_io.synthetic_weights(getattr(yolo_yaml, args.model), seed=0)
    ctx = HipContext(W, device=0, dtype=args.dtype, max_batch=B, max_h=S, max_w=S)
    only = None
    if args.only:
        pre = tuple(p for p in args.only.split(',') if p)
        only = {c for c in range(ctx.num_conv_cfgs()) if ctx.conv_cfg_name(c).startswith(pre)}
    else:
        ctx.load_tuned('/nonexistent')      # measure against the heuristic, not an older table
        ctx.lib.mdhip_set_tuned(ctx.h, None, 0)
    x = torch.randint(0, 256, (B, HH, WW, 3), dtype=torch.uint8, device='cuda')
    ctx.preprocess([int(x[i].data_ptr()) for i in range(B)], [(HH, WW, HH, WW, 0, 0)] * B, HH, WW)
    if args.dtype == 'fp8':
        ctx.calibrate(B, HH, WW)
    ctx.forward(B, HH, WW)                     # real activations in every buffer
    infos = ctx.op_infos()
    ncfg = ctx.num_conv_cfgs()
    lines = []
    cache = {}
    entries = {}
    canon = {}
    if args.family_from and os.path.exists(args.family_from):
        for e in json.load(open(args.family_from)).get('entries', []):
            key = (e['n'], e['k'], e['ntaps'], e['stride'], e['has_res'], e['m'] // max(1, e.get('batch', 32)))
            if key not in canon or e.get('batch', 32) > canon[key][1]:
                canon[key] = (bool(ctx.cfg_is_bitwise(e['cfg'])), e.get('batch', 32))
    for oi, o in enumerate(infos):
        if o['kind'] != 0:
            continue
        # A 1x1 conv right behind an upsample reads the low-resolution tensor in place when its configuration can (conv_v2's
        # 160x160); with any other configuration the upsample runs as its own launch.  mdhip_time_op times an op in
        # isolation, WITHOUT that absorption, so it cannot price the difference: a short re-tune (--only) leaves these
        # layers with the entry they have (round 4: three of them taken to a ring configuration cost 0.25 ms of upsample
        # launches and 2.5 GB per step for 0.15 ms of conv time).
        if only is not None and oi > 0 and infos[oi - 1]['kind'] == 2 and o['ntaps'] == 1:
            continue
        sig = (o['m'], o['n'], o['k'], o['ntaps'], o['stride'], o['has_res'])
        if sig in cache:
            ms = cache[sig]
        else:
            ms = []
            fam = canon.get((o['n'], o['k'], o['ntaps'], o['stride'], o['has_res'], o['m'] // B))
            for cfg in range(ncfg):
                if only is not None and cfg not in only and cfg != o['cfg']:
                    ms.append(float('inf'))
                    continue
                if fam is not None and bool(ctx.cfg_is_bitwise(cfg)) != fam[0]:
                    ms.append(float('inf'))
                    continue
                try:
                    ctx.set_op_cfg(o['op'], cfg)
                    ms.append(min(ctx.time_op(o['op'], B, HH, WW, iters=args.iters) for _ in range(args.reps)))
                except Exception:
                    ms.append(float('inf'))
            ctx.set_op_cfg(o['op'], -1)
            cache[sig] = ms
        b = int(np.argmin(ms))
        if not np.isfinite(ms[b]):          # nothing could be timed (e.g. an op that runs inside a fused launch): the table keeps what it has
            continue
        tf = [o['flops'] / (t * 1e-3) / 1e12 if np.isfinite(t) else 0.0 for t in ms]
        entries[sig] = dict(m=sig[0], n=sig[1], k=sig[2], ntaps=sig[3], stride=sig[4], has_res=sig[5], cfg=b,
                            batch=B, ms=round(ms[b], 5), tflops=round(tf[b], 1), name=ctx.conv_cfg_name(b))
        lines.append('{:34s} M={:8d} N={:5d} K={:6d} default={:2d} best={:2d} {:8.3f} ms {:7.1f} TF/s | '.format(
            o['name'], o['m'], o['n'], o['k'], o['cfg'], b, ms[b], tf[b]) +
            ' '.join('{:6.1f}'.format(t) for t in tf))
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    data = {'entries': []}
    if os.path.exists(args.out):
        try:
            data = json.load(open(args.out))
        except Exception:
            data = {'entries': []}
    keep = [e for e in data.get('entries', [])
            if e.get('batch', 32) != B or (e['m'], e['n'], e['k'], e['ntaps'], e['stride'], e['has_res']) not in entries]
    data = {'entries': keep + list(entries.values())}
    json.dump(data, open(args.out, 'w'), indent=1, sort_keys=True)
    table = args.table or (os.path.splitext(args.out)[0] + '_{}_{}_{}x{}.txt'.format(args.model, B, HH, WW))
    with open(table, 'w') as f:
        f.write('TFLOP/s per configuration (columns = cfg 0..{})\n'.format(ncfg - 1))
        f.write('\n'.join(lines) + '\n')
    print('\n'.join(lines))
    ctx.close()


if __name__ == '__main__':
    main()
