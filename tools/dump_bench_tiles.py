#!/usr/bin/env python3
"""
Records, ON THE GPU, the tile configuration every conv op of a benchmarked forward launches
(`mdhip_get_op_info` after one forward of the bench workload) -> tests/golden/bench_tiles.json:
  { "<dtype>:<batch>x<H>x<W>": [name per conv op, in op order; "fused" = ran inside the next 3x3's launch], ... }
tests/test_gpu_headline.py and tests/test_gpu_fp8.py compare the tiles they force from the table (exact layer + M
match) with these lists, so the per-layer parity tests provably run the kernels bench.py's step runs.
Re-run after every re-tune of megadetector_amd/tuned_cfgs*.json:   python tools/dump_bench_tiles.py
"""

import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

WORKLOADS = [                      # (dtype, batch, letterboxed H, W): bench.py's default line and its extra_configs legs
    ('bf16', 32, 1280, 1280),      # BASELINE configs[1]
    ('bf16', 32, 768, 1280),       # configs[3]: 1080x1920 video frames
    ('bf16', 32, 960, 1280),       # SURVEY 8(d) real-shape: 1536x2048
    ('bf16', 32, 896, 1280),       # 3:2 frames: 1600x2400
    ('fp16', 32, 1280, 1280),      # the detector's default storage type
    ('fp16', 32, 768, 1280),
    ('fp16', 32, 960, 1280),
    ('fp16', 32, 896, 1280),
    ('fp8', 64, 1280, 1280),       # configs[4]
]


def main():
    import torch
    from megadetector_amd import weights_io, yolo_yaml
    from megadetector_amd.hip_backend import HipContext
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, 'tests', 'golden', 'bench_tiles.json')
    W = weights_io.synthetic_weights(yolo_yaml.YOLOV5X6_MD, seed=0)
    out = {}
    for dtype, B, H, Wd in WORKLOADS:
        ctx = HipContext(W, device=0, dtype=dtype, max_batch=B, max_h=1280, max_w=1280)
        x = torch.randint(0, 256, (B, H, Wd, 3), dtype=torch.uint8, device='cuda')
        ctx.preprocess([int(x[i].data_ptr()) for i in range(B)], [(H, Wd, H, Wd, 0, 0)] * B, H, Wd)
        if dtype == 'fp8':
            ctx.calibrate(B, H, Wd)
        ctx.forward(B, H, Wd)
        torch.cuda.synchronize()
        names = ['fused' if o['cfg'] < 0 else ctx.conv_cfg_name(o['cfg']) for o in ctx.op_infos() if o['kind'] == 0]
        out['{}:{}x{}x{}'.format(dtype, B, H, Wd)] = names
        print('{}:{}x{}x{}: {} conv ops, {} distinct configurations'.format(dtype, B, H, Wd, len(names), len(set(names))))
        ctx.close()
        del x
    with open(out_path, 'w') as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print('wrote', out_path)


if __name__ == '__main__':
    main()
