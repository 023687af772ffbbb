#!/usr/bin/env python3
"""First-contact diagnostics on the GPU box: per-layer error of the HIP conv stack against the
bf16-emulating oracle (prints a table instead of asserting, so that one run shows where a bug
starts)."""

import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))


def main():
    import parity_util as PU
    from megadetector_amd import weights_io, yolo_yaml
    from megadetector_amd.hip_backend import HipContext
    model = sys.argv[1] if len(sys.argv) > 1 else 'YOLOV5N6_TEST'
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    W = weights_io.synthetic_weights(getattr(yolo_yaml, model), seed=1)
    ctx = HipContext(W, device=0, max_batch=n, max_h=size, max_w=size)
    imgs = PU.structured_images(n, size, size, seed=5)
    ctx.preprocess(imgs, [(size, size, size, size, 0, 0)] * n, size, size)
    got_in = ctx.read_input(n, size, size)
    x, _ = PU.oracle_input(imgs, size, W.max_stride)
    print('input exact:', np.array_equal(got_in, PU.bf16_round_np(x.numpy())))
    ctx.forward(n, size, size)
    keep = {}
    pred_ref, _ = PU.oracle_forward(W, x, emulate_bf16=True, keep=keep)
    for i in sorted(keep):
        got = ctx.read_layer(i, n)
        emax, emean = PU.rel_err(got, keep[i].numpy())
        print('layer {:2d} {:>18s} max {:.3e} mean {:.3e} {}'.format(
            i, str(tuple(got.shape)), emax, emean, 'BAD' if (emax > 3e-2 or emean > 4e-3 or not np.isfinite(emax)) else ''))
    pred = ctx.read_predictions(n, size, size)
    print('pred box rel', PU.rel_err(pred[..., :4], pred_ref[..., :4].numpy()),
          'conf abs', float(np.abs(pred[..., 4:] - pred_ref[..., 4:].numpy()).max()))
    pred32, _ = PU.oracle_forward(W, x, emulate_bf16=False)
    print('vs fp32 oracle: box rel', PU.rel_err(pred[..., :4], pred32[..., :4].numpy()),
          'conf abs', float(np.abs(pred[..., 4:] - pred32[..., 4:].numpy()).max()))
    for o in ctx.op_infos():
        print(o)
    ctx.close()


if __name__ == '__main__':
    main()
