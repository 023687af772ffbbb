#!/usr/bin/env python3
"""
A/B of the strip configurations on the 80-channel C3 block (layer 2 of the x6 stack) INSIDE whole forwards: every
'v5:strip*' configuration is forced onto the block's four bottleneck 3x3s in turn (fused launches: 1x1 + 3x3 + residual),
the per-op events of the forward are averaged over --reps forwards, and the predictions are compared bit for bit with
the first configuration's.  GPU box:  python tools/c80_ab.py --dtype bf16 --shape 1280x1280 --batch 32
"""
import argparse
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--shape', default='1280x1280')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--rounds', type=int, default=2)
    args = ap.parse_args()
    import parity_util as PU
    from megadetector_amd import weights_io, yolo_yaml
    from megadetector_amd.hip_backend import HipContext
    h, w = (int(v) for v in args.shape.split('x'))
    B = args.batch
    W = weights_io.synthetic_weights(yolo_yaml.YOLOV5X6_MD, seed=0)
    ctx = HipContext(W, device=0, dtype=args.dtype, max_batch=B, max_h=h, max_w=w)
    imgs = PU.random_images(B, h, w, seed=5)
    ctx.preprocess(imgs, [(h, w, h, w, 0, 0)] * B, h, w)
    infos = ctx.op_infos()
    strips = [c for c in range(ctx.num_conv_cfgs()) if ctx.conv_cfg_name(c).startswith('v5:strip')]
    ops = [o['op'] for o in infos if o['kind'] == 0 and ctx.op_supports_cfg(o['op'], strips[0])]
    print('{} {}x{} batch {}: ops {}'.format(args.dtype, h, w, B, [infos[o]['name'] for o in ops]))
    ref = None
    pre = [o - 1 for o in ops]                                   # the bottlenecks' 1x1 convs (absorbed when the block runs fused)
    for rnd in range(args.rounds):
        for cfg in [-1] + strips:                                # -1: whatever the tile table picks at this batch size
            for op in ops:
                ctx.set_op_cfg(op, cfg)
            ctx.forward(B, h, w)
            pred = ctx.read_predictions(B).copy()
            if ref is None:
                ref = pred
            same = bool(np.array_equal(pred, ref))
            ms = np.zeros(len(infos))
            for _ in range(args.reps):
                ms += ctx.forward_timed(B, h, w)
            ms /= args.reps
            skipped = sum(1 for o in ctx.op_infos() if o['kind'] == 0 and o['cfg'] < 0)
            name = ctx.conv_cfg_name(cfg) if cfg >= 0 else 'table: ' + ctx.conv_cfg_name(ctx.op_infos()[ops[0]]['cfg'])
            print('  {:<34} bottlenecks (1x1 + 3x3) {}  = {:.4f} ms   forward (sum of op events) {:.3f} ms   1x1s absorbed {}   bits {}'.format(
                name, ' '.join('{:.4f}'.format(ms[o] + (ms[q] if skipped == 0 else 0.0)) for o, q in zip(ops, pre)),
                float(sum(ms[o] for o in ops) + (sum(ms[q] for q in pre) if skipped == 0 else 0.0)), float(ms.sum()), skipped,
                'identical' if same else 'DIFFERENT'), flush=True)
    ctx.close()


if __name__ == '__main__':
    main()
