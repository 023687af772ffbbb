"""Helpers shared by the GPU parity tests, tests/gpu_diag.py and bench.py's checker leg."""

import numpy as np
import torch

from oracle import pre_post as O
from oracle import yolov5 as Y


def bf16_round_np(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.bfloat16).float().numpy()


def random_images(n, h, w, seed=0):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(n)]


def structured_images(n, h, w, seed=0):
    """Smooth blobs + noise: gives the resize kernel real gradients to interpolate."""
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    out = []
    for _ in range(n):
        img = np.zeros((h, w, 3), dtype=np.float32)
        for _ in range(6):
            cy, cx = rng.random() * h, rng.random() * w
            s = 20 + rng.random() * 0.3 * max(h, w)
            col = rng.random(3) * 255
            img += np.exp(-((ys - cy) ** 2 + (xs - cx) ** 2) / (2 * s * s))[..., None] * col
        img += rng.normal(0, 12, img.shape)
        out.append(np.clip(img, 0, 255).astype(np.uint8))
    return out


def oracle_input(images, image_size, stride):
    """letterbox every image with the oracle and stack to the NCHW fp32 batch the reference builds."""
    infos = [O.preprocess_image_classic(im, image_size=image_size, stride=stride) for im in images]
    shapes = {i['img_processed'].shape for i in infos}
    assert len(shapes) == 1, 'images of one test batch must letterbox to one shape'
    return O.to_batch_tensor([i['img_processed'] for i in infos]), infos


def fp8_scale_map(ctx):
    """{(model layer, bottleneck index): scale} of an fp8 HipContext, for the oracle's emulate_bf16='fp8' mode"""
    out, seen = {}, {}
    for scale, layer, _op in ctx.fp8_scales():
        j = seen.get(layer, 0)
        seen[layer] = j + 1
        out[(layer, j)] = scale
    return out


def oracle_forward(weights, x, emulate_bf16, keep=None, augment=False, fp8_scales=None):
    fw = Y.Forward(weights.yaml, weights.torch_state(), emulate_bf16=emulate_bf16, keep=keep, fp8_scales=fp8_scales)
    with torch.no_grad():
        return (fw.forward_augment(x) if augment else fw(x)), fw


def rel_err(a, b):
    """(max abs err / max abs ref, mean abs err / mean abs ref)"""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.abs(a - b)
    return float(d.max() / max(np.abs(b).max(), 1e-30)), float(d.mean() / max(np.abs(b).mean(), 1e-30))


def oracle_detections(pred, infos, batch_hw, threshold, iou=0.45):
    dets = O.nms(pred, conf_thres=threshold, iou_thres=iou)
    out = []
    for d, info in zip(dets, infos):
        lst, mx = O.format_detections(d, batch_hw, info['img_original'].shape, info['scaling_shape'], threshold)
        out.append({'detections': lst, 'max_detection_conf': mx})
    return out


def random_predictions(seed, batch, n, n_clusters=12, img=1280.0):
    """Clustered boxes so that suppression actually happens; obj skewed towards 0."""
    g = torch.Generator().manual_seed(seed)
    centres = torch.rand(n_clusters, 2, generator=g) * img
    sizes = 40 + torch.rand(n_clusters, 2, generator=g) * 300
    which = torch.randint(0, n_clusters, (batch, n), generator=g)
    xy = centres[which] + torch.randn(batch, n, 2, generator=g) * 12
    wh = sizes[which] * (1 + 0.15 * torch.randn(batch, n, 2, generator=g)).clamp(0.3, 2)
    obj = torch.rand(batch, n, 1, generator=g) ** 6
    cls = torch.rand(batch, n, 3, generator=g)
    return torch.cat([xy, wh, obj, cls], 2).float()
