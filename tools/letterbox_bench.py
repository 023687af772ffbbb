#!/usr/bin/env python3
"""Times mdhip_preprocess on device-resident sources: the streaming kernels against the general one (mdhip_set_option
"letterbox_general"), per source shape.  GPU box:  python tools/letterbox_bench.py [--src 1536x2048] [--batch 32]"""
import argparse
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--src', default='1536x2048,1080x1920,1600x2400,1280x1280,480x640')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--dtype', default='bf16')
    args = ap.parse_args()
    import torch
    from megadetector_amd import weights_io, yolo_yaml
    from megadetector_amd.hip_backend import HipContext
    from megadetector_amd.postprocess import letterbox_geometry
    W = weights_io.synthetic_weights(yolo_yaml.YOLOV5N6_TEST, seed=1)
    B = args.batch
    ctx = HipContext(W, device=0, dtype=args.dtype, max_batch=B, max_h=1280, max_w=1280)
    for src in args.src.split(','):
        h0, w0 = (int(v) for v in src.split('x'))
        g = letterbox_geometry((h0, w0), new_shape=1280, stride=64)
        h, w = g['out_hw']
        geoms = [(h0, w0, g['new_unpad'][1], g['new_unpad'][0], g['top'], g['left'])] * B
        imgs = [torch.randint(0, 256, (h0, w0, 3), dtype=torch.uint8, device='cuda') for _ in range(B)]
        ptrs = [int(t.data_ptr()) for t in imgs]
        nbytes = B * (h0 * w0 * 3 + h * w * 3 * 2)
        row = []
        for general in (1, 0, 1, 0):
            ctx.set_option('letterbox_general', general)
            for _ in range(3):
                ctx.preprocess(ptrs, geoms, h, w)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                ctx.preprocess(ptrs, geoms, h, w)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            row.append('{} {:.4f} ms = {:.2f} TB/s'.format('general  ' if general else 'streaming', ms, nbytes / ms / 1e9))
        print('{} -> {}x{} x{}: '.format(src, h, w, B) + ' | '.join(row))
    ctx.close()


if __name__ == '__main__':
    main()
