"""
Second-source checks of the oracle's restatements of OpenCV's resize (oracle/pre_post.py:82-190).

cv2 is not installed here and the reference calls it as a third-party dependency (pytorch_detector.py:231-343 ->
yolov5 letterbox -> cv2.resize), so these two functions are "parity unpinned" against OpenCV itself (oracle header,
DESIGN.md section 3).  What CAN be checked offline is that they compute the mathematical operation OpenCV documents,
against an implementation nobody here wrote:

  * INTER_LINEAR = 2-tap bilinear interpolation at half-pixel centres without antialiasing =
    torch.nn.functional.interpolate(mode='bilinear', align_corners=False, antialias=False) in float arithmetic;
    OpenCV evaluates it for uint8 with 11-bit fixed-point weights, so the two agree to within one grey level
    (rounding of the weights and of the two passes);
  * INTER_AREA with an integer scale factor = the mean over each k x k block = torch avg_pool2d, rounded.

A transcription error in the coefficient tables, the border clamps or the pass order shows as differences of many grey
levels on the structured images used here.
"""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import pre_post as O
import parity_util as PU


def _torch_bilinear(img, dst_w, dst_h):
    x = torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None]
    y = F.interpolate(x, size=(dst_h, dst_w), mode='bilinear', align_corners=False, antialias=False)
    return y[0].permute(1, 2, 0).numpy()


@pytest.mark.parametrize('src_hw,dst_hw', [
    ((480, 640), (960, 1280)),       # upscale x2 (scaleup=True in classic mode)
    ((1536, 2048), (960, 1280)),     # the 3-MP camera-trap frame of bench.py --src
    ((1080, 1920), (720, 1280)),     # x2/3
    ((333, 517), (824, 1280)),       # nothing divides anything
    ((1200, 900), (1280, 960)),      # portrait, slight upscale
    ((97, 41), (13, 7)),             # strong reduction (bilinear skips source pixels, as OpenCV does)
])
def test_linear_resize_is_half_pixel_bilinear(src_hw, dst_hw):
    img = PU.structured_images(1, src_hw[0], src_hw[1], seed=src_hw[0] + dst_hw[1])[0]
    got = O.resize_linear_u8(img, dst_hw[1], dst_hw[0]).astype(np.float64)
    ref = _torch_bilinear(img, dst_hw[1], dst_hw[0]).astype(np.float64)
    d = np.abs(got - ref)
    # fixed point vs float: measured max 0.63 .. 0.76 of a grey level, 89 .. 98 % of the pixels within rounding (0.5),
    # mean signed difference -0.02 .. -0.11 (the two truncating shifts of the fixed-point passes); a half-pixel error
    # in the coordinate map or a swapped weight pair is several grey levels on this content
    assert d.max() <= 0.8, d.max()
    assert (d <= 0.5 + 1e-3).mean() > 0.88, (d <= 0.5 + 1e-3).mean()
    assert abs((got - ref).mean()) < 0.15


def test_linear_resize_on_a_ramp_has_no_phase_error():
    """a horizontal ramp resized by 2 must stay the same ramp sampled at half-pixel centres (an off-by-half in the
    coordinate map shifts it by a quarter of a source step)"""
    w = 64
    ramp = np.repeat((np.arange(w, dtype=np.float32) * 4)[None, :, None], 8, 0).repeat(3, 2).astype(np.uint8)
    out = O.resize_linear_u8(ramp, 2 * w, 16)[4, :, 0].astype(np.float64)
    expect = np.clip(((np.arange(2 * w) + 0.5) / 2 - 0.5), 0, w - 1) * 4
    assert np.abs(out - expect).max() <= 1.0


@pytest.mark.parametrize('k', [2, 3, 4])
def test_area_resize_with_integer_factor_is_the_block_mean(k):
    h, w = 24 * k, 40 * k
    img = PU.structured_images(1, h, w, seed=90 + k)[0]
    got = O.resize_area_u8(img, w // k, h // k).astype(np.float64)
    x = torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None]
    ref = F.avg_pool2d(x, k)[0].permute(1, 2, 0).numpy().astype(np.float64)
    d = np.abs(got - ref)
    assert d.max() <= 0.5 + 1e-3, d.max()          # = rounding of the exact mean


def test_area_resize_fractional_factor_conserves_the_mean_and_bounds():
    """fractional INTER_AREA weights every source pixel by its overlap with the destination pixel: the image mean is
    conserved (to rounding) and every output lies within the range of the source pixels it overlaps"""
    img = PU.structured_images(1, 300, 500, seed=5)[0]
    out = O.resize_area_u8(img, 333, 200)
    assert abs(out.astype(np.float64).mean() - img.astype(np.float64).mean()) < 0.3
    # against torch's antialiased bilinear (a different low-pass kernel): same image to a few grey levels on this content
    x = torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None]
    ref = F.interpolate(x, size=(200, 333), mode='bilinear', align_corners=False, antialias=True)[0].permute(1, 2, 0).numpy()
    assert np.abs(out.astype(np.float64) - ref).mean() < 2.5
