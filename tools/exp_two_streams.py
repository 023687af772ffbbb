#!/usr/bin/env python3
"""Experiment: does running two half-batches on two streams (two contexts) fill the kernels' tail rounds?
Prints forward-only images/s for one context at batch B and for two contexts at batch B/2 on two streams."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from megadetector_amd import weights_io, yolo_yaml
from megadetector_amd.hip_backend import HipContext

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    S, steps = 1280, 20
    W = weights_io.synthetic_weights(yolo_yaml.YOLOV5X6_MD, seed=0)
    x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, device='cuda')
    def run(nctx):
        b = B // nctx
        ctxs = [HipContext(W, device=0, max_batch=b, max_h=S, max_w=S) for _ in range(nctx)]
        streams = [torch.cuda.Stream() for _ in range(nctx)]
        ptrs = [[int(x[k * b + i].data_ptr()) for i in range(b)] for k in range(nctx)]
        for k, c in enumerate(ctxs):
            c.preprocess(ptrs[k], [(S, S, S, S, 0, 0)] * b, S, S, stream=streams[k].cuda_stream)
        def step():
            for k, c in enumerate(ctxs):
                c.forward(b, S, S, stream=streams[k].cuda_stream)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        for c in ctxs:
            c.close()
        return B * steps / dt
    print('one context, batch {}: {:.1f} images/s (forward only)'.format(B, run(1)))
    print('{} contexts, batch {} each, {} streams: {:.1f} images/s (forward only)'.format(parts, B // parts, parts, run(parts)))

if __name__ == '__main__':
    main()
