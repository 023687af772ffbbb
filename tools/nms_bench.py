"""
NMS stress measurement: synthetic (B, 102000, 8) fp32 predictions with
obj ~ Beta(0.05, 1), boxes jittered around 50 cluster centres, seed 0; thresholds chosen so that
about 1 % and 10 % of the anchors pass, plus the batch-mode threshold 1e-5.  Times
mdhip_nms (candidate compaction + per-class greedy NMS + top-300 + D2H) on the device-resident tensor.

(Exactness of the NMS on this kind of input is the business of tests/test_gpu_parity.py, not of this tool.)

Usage (GPU box): python tools/nms_bench.py [--batch 32] [--out gpurun_out/nms_stress.json]
"""

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def stress_predictions(b, a, seed=0):
    rng = np.random.default_rng(seed)
    centres = rng.uniform(100, 1180, size=(50, 2)).astype(np.float32)
    which = rng.integers(0, 50, size=(b, a))
    pred = np.empty((b, a, 8), dtype=np.float32)
    pred[..., 0:2] = centres[which] + rng.normal(0, 25, size=(b, a, 2)).astype(np.float32)
    pred[..., 2:4] = rng.uniform(20, 300, size=(b, a, 2)).astype(np.float32)
    pred[..., 4] = rng.beta(0.05, 1.0, size=(b, a)).astype(np.float32)
    pred[..., 5:8] = rng.uniform(0, 1, size=(b, a, 3)).astype(np.float32)
    return pred


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    import torch
    from megadetector_amd import weights_io, yolo_yaml
    from megadetector_amd.hip_backend import HipContext

    B, S = args.batch, 1280
    W = weights_io.synthetic_weights(yolo_yaml.YOLOV5N6_TEST, seed=0)
    ctx = HipContext(W, device=0, max_batch=B, max_h=S, max_w=S)
    A = ctx.num_anchors(S, S)
    imgs = [np.zeros((S, S, 3), dtype=np.uint8)] * B
    ctx.preprocess(imgs, [(S, S, S, S, 0, 0)] * B, S, S)
    ctx.forward(B, S, S)                       # sets the context's current shape (A anchors per image)
    pred = stress_predictions(B, A)
    score = pred[..., 4] * pred[..., 5:8].max(-1)
    cases = [('1e-5 (batch mode)', 1e-5), ('~10 % pass', float(np.quantile(score, 0.90))),
             ('~1 % pass', float(np.quantile(score, 0.99)))]
    rows = []
    for name, thr in cases:
        out, counts = ctx.nms_on(pred, thr, 0.45, 300)          # uploads the tensor into the context's buffer
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            ctx.nms(B, thr, 0.45, 300)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.reps * 1e3
        passing = float(((pred[..., 4] > thr) & (score > thr)).mean())
        rows.append({'case': name, 'conf_thres': thr, 'candidates_frac': passing,
                     'candidates_per_image': passing * A, 'ms_per_batch': ms, 'us_per_image': ms / B * 1e3,
                     'read_GBps': B * A * 8 * 4 / (ms * 1e-3) / 1e9, 'kept_mean': float(counts.mean())})
        print('{:20s} thr {:.3g}: {:6.0f} candidates/image, {:7.3f} ms per batch of {} ({:6.1f} us/image, {:6.1f} GB/s of '
              'prediction reads), kept {:.1f}'.format(
                  name, thr, passing * A, ms, B, ms / B * 1e3, rows[-1]['read_GBps'], counts.mean()))
    if args.out:
        with open(args.out, 'w') as f:
            json.dump({'batch': B, 'anchors': A, 'rows': rows}, f, indent=1)
    ctx.close()


if __name__ == '__main__':
    main()
