#!/usr/bin/env python3
"""Runs ONE conv op (by name, with a given tile configuration) a few times so that a
`rocprofv3 --pmc ...` pass sees only that kernel.  usage: pmc_probe.py "<op name>" <cfg> [iters]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch
    from megadetector_amd import weights_io, yolo_yaml
    from megadetector_amd.hip_backend import HipContext
    name, cfg = sys.argv[1], int(sys.argv[2])
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    B, S = 32, 1280
    W = weights_io.synthetic_weights(yolo_yaml.YOLOV5X6_MD, seed=0)
    ctx = HipContext(W, device=0, max_batch=B, max_h=S, max_w=S)
    x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, device='cuda')
    ctx.preprocess([int(x[i].data_ptr()) for i in range(B)], [(S, S, S, S, 0, 0)] * B, S, S)
    ctx.forward(B, S, S)
    op = [o for o in ctx.op_infos() if o['name'] == name][0]
    ctx.set_op_cfg(op['op'], cfg)
    ms = ctx.time_op(op['op'], B, S, S, iters=iters)
    print('{} cfg {}: {:.4f} ms, {:.1f} TF/s'.format(name, cfg, ms, op['flops'] / ms / 1e9))
    ctx.close()


if __name__ == '__main__':
    main()
