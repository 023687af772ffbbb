"""
compatibility_mode 'modern' (reference pytorch_detector.py:1036-1109,:1318-1321,:1369-1397) on the CPU: the
oracle's INTER_AREA restatement against the exact area average it approximates, and the host-side geometry /
box formatting of the product path (megadetector_amd/postprocess.py) against the oracle.  cv2 itself is not
available offline: parity with OpenCV's INTER_AREA is unpinned (oracle/pre_post.py).
"""

import numpy as np
import pytest
import torch

from oracle import pre_post as O
from megadetector_amd import postprocess as P


def _exact_area(img, dw, dh):
    h, w = img.shape[:2]
    sx, sy = w / dw, h / dh
    out = np.zeros((dh, dw, 3))
    for y in range(dh):
        wy = np.clip(np.minimum(np.arange(h) + 1, (y + 1) * sy) - np.maximum(np.arange(h), y * sy), 0, 1)
        for x in range(dw):
            wx = np.clip(np.minimum(np.arange(w) + 1, (x + 1) * sx) - np.maximum(np.arange(w), x * sx), 0, 1)
            out[y, x] = (wy[:, None, None] * wx[None, :, None] * img).sum((0, 1)) / (sx * sy)
    return out


@pytest.mark.parametrize('dst', [(50, 37), (40, 30), (20, 15), (27, 20), (79, 59), (80, 60)])
def test_area_resize_is_the_rounded_area_average(dst):
    rng = np.random.default_rng(dst[0])
    img = rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)
    out = O.resize_area_u8(img, dst[0], dst[1])
    assert out.shape == (dst[1], dst[0], 3) and out.dtype == np.uint8
    assert np.abs(out.astype(np.float64) - _exact_area(img, dst[0], dst[1])).max() <= 0.5 + 1e-3
    const = np.full((60, 80, 3), 137, np.uint8)
    assert (O.resize_area_u8(const, dst[0], dst[1]) == 137).all()


@pytest.mark.parametrize('shape', [(1536, 2048), (2048, 1536), (480, 640), (1280, 1280), (1080, 1920), (3000, 4000),
                                   (97, 211), (1281, 640), (720, 2560)])
@pytest.mark.parametrize('use_ceil', [False, True])
def test_modern_geometry_host_equals_oracle(shape, use_ceil):
    a = P.modern_geometry(shape, 1280, 64, use_ceil)
    b = O.modern_geometry(shape, 1280, 64, use_ceil)
    assert a['resized_hw'] == b['resized_hw'] and a['target_shape'] == b['target_shape']
    assert a['interp'] == {None: 0, 'linear': 0, 'area': 1}[b['interp']]
    for k in ('top', 'left', 'out_hw', 'new_unpad', 'pad', 'ratio'):
        assert a['letterbox'][k] == b['letterbox'][k]
    assert max(a['resized_hw']) in (1280, 1281)        # long side -> image size (ceil can overshoot by one)
    assert a['letterbox']['out_hw'] == a['target_shape'] and a['target_shape'][0] % 64 == 0


def test_modern_box_formatting_host_equals_oracle():
    rng = np.random.default_rng(4)
    scaling_shape = (1536, 2048, 3)
    m = P.modern_geometry(scaling_shape[:2], 1280, 64)
    hh, ww = m['target_shape']
    k = 60
    det = np.zeros((k, 6), np.float32)
    det[:, 0] = rng.uniform(-20, ww - 100, k)
    det[:, 1] = rng.uniform(-20, hh - 100, k)
    det[:, 2] = det[:, 0] + rng.uniform(5, 400, k)
    det[:, 3] = det[:, 1] + rng.uniform(5, 400, k)
    det[:, 4] = np.sort(rng.uniform(0, 1, k))[::-1]
    det[:, 5] = rng.integers(0, 3, k)
    resized = (m['resized_hw'][0], m['resized_hw'][1], 3)
    pad = m['letterbox']['pad']
    got, gmax = P.format_detections(det, (hh, ww), resized, scaling_shape, 0.05, modern=True, letterbox_pad=pad)
    want, wmax = O.format_detections(torch.from_numpy(det), (hh, ww), resized, scaling_shape, 0.05, modern=True,
                                     letterbox_pad=pad)
    assert got == want and gmax == wmax and len(got) > 10
    classic, _ = P.format_detections(det, (hh, ww), scaling_shape, scaling_shape, 0.05)
    assert classic != got
