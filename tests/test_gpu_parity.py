"""
GPU parity tests (run with -m gpu on an MI355X): every stage of the HIP hot path, called
through the C ABI (include/mdhip.h via megadetector_amd.hip_backend), against the CPU oracle
on the same seeded inputs, against the committed golden fixtures, and -- at full size --
through size-independent properties.

Tolerances (stated once, used below):
  * integer / byte / index work (letterbox + resize, NMS selection and order, categories):
    bit-exact.
  * conv stack vs the bf16-emulating oracle (same storage rounding, different fp32 summation
    order, so single-ulp bf16 flips (2^-8 relative) appear and propagate through ~100 layers):
    max|err| <= 3e-2 * max|ref| and mean|err| <= 8e-3 * mean|ref| per layer.
  * predictions vs the fp32 oracle (what the reference computes): reported; end-to-end
    every confident detection must have a same-class candidate at IoU >= 0.85 (md_tests.py:124)
    in the oracle's predictions with |dconf| <= 0.04 (bf16-emulating oracle) and <= 0.08 (fp32
    oracle = what the reference computes): the price of bf16 activation storage on the seeded-weight
    test model (see the comment at E2E_CONF_TOL_*).
"""

import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN
import parity_util as PU
from oracle import pre_post as O

pytestmark = pytest.mark.gpu

LAYER_MAX_TOL = 3e-2
LAYER_MEAN_TOL = 8e-3
# end to end, per confident (conf >= 0.1) detection, against its best same-class IoU >= 0.85 candidate:
# Measured on MI355X with the seeded-weight test model: 0.028 / 0.056.  bf16 keeps 8 significant bits, every
# conv output is rounded to it, and two evaluations that only differ in fp32 summation order already flip
# single bf16 ulps that then propagate through ~100 layers into Detect logits of magnitude ~5 (the
# synthetic Detect gain makes confident detections exist at all).  The reference's own bar between
# environments is 0.005-0.01 (md_tests.py:96-100,1779); it is not reachable against an fp32 evaluation
# with bf16 activation storage on these weights -- DESIGN.md section 6 discusses it (and the fp16 option).
E2E_CONF_TOL_BF16_ORACLE = 0.04    # oracle with the same bf16 storage rounding, different summation order
E2E_CONF_TOL_FP32_ORACLE = 0.08    # fp32 oracle (= what the reference computes)
# The detector's DEFAULT storage type is fp16 (megadetector_amd/detector.py DEFAULT_DTYPE): there the bar is the
# reference's own (md_tests.py:96-100 max_conf_error 0.005, CI 0.01 at :1779), against the fp32 oracle, on the same
# ill-conditioned seeded weights:
E2E_CONF_TOL_FP16 = {True: 0.005, False: 0.01}      # keyed like the loop below: storage-emulating oracle / fp32 oracle


@pytest.fixture(scope='module')
def n6():
    from megadetector_amd import weights_io, yolo_yaml
    from megadetector_amd.hip_backend import HipContext
    W = weights_io.synthetic_weights(yolo_yaml.YOLOV5N6_TEST, seed=1)
    ctx = HipContext(W, device=0, max_batch=4, max_h=320, max_w=320)
    yield W, ctx
    ctx.close()


def _identity_geoms(images):
    return [(im.shape[0], im.shape[1], im.shape[0], im.shape[1], 0, 0) for im in images]


# ---------------------------------------------------------------------------------------
# preprocess: bit-exact
# ---------------------------------------------------------------------------------------
def test_preprocess_identity_bit_exact(n6):
    W, ctx = n6
    imgs = PU.random_images(3, 256, 192, seed=3)
    ctx.preprocess(imgs, _identity_geoms(imgs), 256, 192)
    got = ctx.read_input(3, 256, 192)
    ref = PU.bf16_round_np(O.to_batch_tensor(imgs).numpy())
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize('shape', [(300, 400), (97, 211), (640, 360), (1000, 750), (128, 128)])
def test_preprocess_letterbox_resize_bit_exact(n6, shape):
    from megadetector_amd.postprocess import letterbox_geometry
    W, ctx = n6
    imgs = PU.structured_images(2, shape[0], shape[1], seed=shape[0])
    g = letterbox_geometry(shape, new_shape=256, stride=64)
    geoms = [(shape[0], shape[1], g['new_unpad'][1], g['new_unpad'][0], g['top'], g['left'])] * 2
    h, w = g['out_hw']
    ctx.preprocess(imgs, geoms, h, w)
    got = ctx.read_input(2, h, w)
    x, infos = PU.oracle_input(imgs, 256, 64)
    assert tuple(x.shape[2:]) == (h, w)
    np.testing.assert_array_equal(got, PU.bf16_round_np(x.numpy()))


@pytest.mark.parametrize('shape', [(256, 191), (191, 256), (256, 130), (130, 256), (256, 256), (255, 256)])
def test_preprocess_streaming_copy_kernel_bit_exact(n6, shape, monkeypatch):
    """[r5] batches without resampling (long side already at the network size) take the streaming-copy letterbox kernel
    (aligned dword loads + a u8 -> storage-type table in LDS + 16-byte stores): bit-exact against the oracle's letterbox
    (reference pytorch_detector.py:1104-1109, :1283-1306) for row lengths that are not multiples of four bytes (every row
    starts at another alignment), padding on the left / right / top / bottom, device pointers at odd addresses, and the
    same bits as the general kernel (mdhip_set_option "letterbox_general")."""
    from megadetector_amd.postprocess import letterbox_geometry
    W, ctx = n6
    g = letterbox_geometry(shape, new_shape=256, stride=64)
    assert (g['new_unpad'][1], g['new_unpad'][0]) == shape          # no resampling
    imgs = PU.random_images(3, shape[0], shape[1], seed=shape[0] + 7 * shape[1])
    geoms = [(shape[0], shape[1], shape[0], shape[1], g['top'], g['left'])] * 3
    h, w = g['out_hw']
    x, _ = PU.oracle_input(imgs, 256, 64)
    assert tuple(x.shape[2:]) == (h, w)
    want = PU.bf16_round_np(x.numpy())
    ctx.preprocess(imgs, geoms, h, w)                                # host arrays: staged at aligned addresses
    np.testing.assert_array_equal(ctx.read_input(3, h, w), want)
    # device-resident sources at odd addresses (a view one / two / three bytes into a buffer)
    n_bytes = shape[0] * shape[1] * 3
    bufs = [torch.zeros(n_bytes + 8, dtype=torch.uint8, device='cuda') for _ in range(3)]
    ptrs = []
    for k, (b, im) in enumerate(zip(bufs, imgs)):
        b[k + 1:k + 1 + n_bytes] = torch.from_numpy(im.reshape(-1)).cuda()
        ptrs.append(int(b.data_ptr()) + k + 1)
    torch.cuda.synchronize()
    ctx.preprocess(ptrs, geoms, h, w)
    np.testing.assert_array_equal(ctx.read_input(3, h, w), want)
    ctx.set_option('letterbox_general', 1)
    try:
        ctx.preprocess(ptrs, geoms, h, w)
        np.testing.assert_array_equal(ctx.read_input(3, h, w), want)
    finally:
        ctx.set_option('letterbox_general', 0)


@pytest.mark.parametrize('case', ['down_4x3', 'up', 'mixed_sources', 'tiny', 'one_axis'])
def test_preprocess_streaming_bilinear_kernel_bit_exact(n6, case):
    """[r6] batches whose images are resampled with cv2.INTER_LINEAR (every real camera image: reference
    pytorch_detector.py:1104-1109) take the streaming bilinear letterbox kernel (source rows staged in LDS as aligned
    dwords, column weights once per column pair, OpenCV's two fixed-point passes, the u8 -> storage-type table, 16-byte
    stores): bit-exact against the oracle's letterbox AND against the general kernel (mdhip_set_option
    "letterbox_general"), for shrinking and enlarging, sources of different sizes in one batch (one of them not resampled at
    all), images of a few pixels, a resize along one axis only, and device pointers at odd addresses."""
    from megadetector_amd.postprocess import letterbox_geometry
    W, ctx = n6
    shapes = {'down_4x3': [(384, 512)] * 2, 'up': [(60, 100), (120, 200)], 'mixed_sources': [(192, 256), (600, 800), (96, 128)],
              'tiny': [(3, 5), (6, 10)], 'one_axis': [(255, 100)] * 2}[case]
    imgs = [PU.structured_images(1, hh, ww, seed=hh + 3 * ww)[0] for hh, ww in shapes]
    geoms, out = [], None
    for (hh, ww) in shapes:
        g = letterbox_geometry((hh, ww), new_shape=256, stride=64)
        geoms.append((hh, ww, g['new_unpad'][1], g['new_unpad'][0], g['top'], g['left']))
        assert out in (None, g['out_hw']), 'the images of one batch letterbox to one shape'
        out = g['out_hw']
    h, w = out
    n = len(imgs)
    want = np.concatenate([PU.bf16_round_np(PU.oracle_input([im], 256, 64)[0].numpy()) for im in imgs], 0)
    assert want.shape[2:] == (h, w)
    ctx.preprocess(imgs, geoms, h, w)
    np.testing.assert_array_equal(ctx.read_input(n, h, w), want)
    # device-resident sources at odd addresses
    bufs, ptrs = [], []
    for k, im in enumerate(imgs):
        b = torch.zeros(im.size + 8, dtype=torch.uint8, device='cuda')
        b[k + 1:k + 1 + im.size] = torch.from_numpy(im.reshape(-1)).cuda()
        bufs.append(b)
        ptrs.append(int(b.data_ptr()) + k + 1)
    torch.cuda.synchronize()
    ctx.preprocess(ptrs, geoms, h, w)
    np.testing.assert_array_equal(ctx.read_input(n, h, w), want)
    ctx.set_option('letterbox_general', 1)
    try:
        ctx.preprocess(ptrs, geoms, h, w)
        np.testing.assert_array_equal(ctx.read_input(n, h, w), want)
    finally:
        ctx.set_option('letterbox_general', 0)


# ---------------------------------------------------------------------------------------
# conv stack: layer by layer against the bf16-emulating oracle
# ---------------------------------------------------------------------------------------
def test_every_layer_matches_bf16_oracle(n6):
    W, ctx = n6
    imgs = PU.structured_images(2, 256, 256, seed=5)
    ctx.preprocess(imgs, _identity_geoms(imgs), 256, 256)
    ctx.forward(2, 256, 256)
    x, _ = PU.oracle_input(imgs, 256, 64)
    keep = {}
    pred_ref, _ = PU.oracle_forward(W, x, emulate_bf16=True, keep=keep)
    worst = []
    for i in sorted(keep):
        got = ctx.read_layer(i, 2)
        emax, emean = PU.rel_err(got, keep[i].numpy())
        worst.append((i, emax, emean))
    bad = [t for t in worst if t[1] > LAYER_MAX_TOL or t[2] > LAYER_MEAN_TOL]
    assert not bad, 'layers out of tolerance (layer, max, mean): {}'.format(bad)
    pred = ctx.read_predictions(2, 256, 256)
    assert pred.shape == tuple(pred_ref.shape)
    emax, emean = PU.rel_err(pred[..., :4], pred_ref[..., :4].numpy())
    assert emax < 3e-2 and emean < 8e-3
    assert np.abs(pred[..., 4:] - pred_ref[..., 4:].numpy()).max() < 2e-2


# fp16 storage (MDHIP_DTYPE_FP16): 11 significant bits instead of 8
F16_LAYER_MAX_TOL = 4e-3
F16_LAYER_MEAN_TOL = 1e-3
F16_CONF_TOL_FP32_ORACLE = 0.01     # the reference's own bar between environments (md_tests.py:96-100)


def test_fp16_storage_mode_layers_and_end_to_end():
    """
    dtype='fp16': the same kernels built for fp16 storage.  Preprocess bit-exact, every layer against the
    fp16-emulating oracle (tolerances 8x tighter than bf16), both summation-order families, batch
    invariance bitwise, and the decoded confidences against the *fp32* oracle (= what the reference
    computes) within the reference's own cross-environment tolerance of 0.01.
    """
    from megadetector_amd import weights_io, yolo_yaml
    from megadetector_amd.hip_backend import HipContext
    W = weights_io.synthetic_weights(yolo_yaml.YOLOV5S6_TEST, seed=3)
    HH, WW = 384, 640
    ctx = HipContext(W, device=0, dtype='fp16', max_batch=2, max_h=HH, max_w=WW)
    try:
        imgs = PU.structured_images(2, HH, WW, seed=61)
        ctx.preprocess(imgs, _identity_geoms(imgs), HH, WW)
        x, _ = PU.oracle_input(imgs, WW, 64)
        assert tuple(x.shape[2:]) == (HH, WW)
        np.testing.assert_array_equal(ctx.read_input(2, HH, WW), x.half().float().numpy())
        keep = {}
        pred16, _ = PU.oracle_forward(W, x, emulate_bf16='fp16', keep=keep)
        pred32, _ = PU.oracle_forward(W, x, emulate_bf16=False)
        convs = [o['op'] for o in ctx.op_infos() if o['kind'] == 0]
        families = {}
        for cfg in [-1] + [c for c in range(ctx.num_conv_cfgs()) if not ctx.cfg_is_bitwise(c)]:
            switched = [op for op in convs if cfg >= 0 and ctx.op_supports_cfg(op, cfg)]
            if cfg >= 0 and not switched:
                continue
            for op in convs:
                ctx.set_op_cfg(op, cfg if op in switched else -1)
            ctx.forward(2, HH, WW)
            bad = []
            for i in sorted(keep):
                e = PU.rel_err(ctx.read_layer(i, 2), keep[i].numpy())
                if e[0] > F16_LAYER_MAX_TOL or e[1] > F16_LAYER_MEAN_TOL:
                    bad.append((i,) + e)
            assert not bad, (cfg, bad)
            got = ctx.read_predictions(2, HH, WW).copy()
            families[cfg] = got
            emax, emean = PU.rel_err(got[..., :4], pred16[..., :4].numpy())
            assert emax < 4e-3 and emean < 1e-3, (cfg, emax, emean)
            d32 = float(np.abs(got[..., 4:] - pred32[..., 4:].numpy()).max())
            assert d32 < F16_CONF_TOL_FP32_ORACLE, (cfg, d32)
            # batch composition invariance: bitwise
            ctx.preprocess([imgs[1]], _identity_geoms([imgs[1]]), HH, WW)
            ctx.forward(1, HH, WW)
            np.testing.assert_array_equal(ctx.read_predictions(1, HH, WW)[0], got[1])
            ctx.preprocess(imgs, _identity_geoms(imgs), HH, WW)
        assert len(families) >= 2, 'no row-segment / row-patch kernel took part'
        # all v1 tile configurations stay bitwise-identical in the fp16 build as well
        for op in convs:
            ctx.set_op_cfg(op, -1)
        base = families[-1]
        for cfg in (0, 5, 15):
            for op in convs:
                ctx.set_op_cfg(op, cfg)
            ctx.forward(2, HH, WW)
            np.testing.assert_array_equal(ctx.read_predictions(2, HH, WW), base)
    finally:
        ctx.close()


def test_p5_family_member_with_other_class_count():
    """
    SURVEY.md 8(f) N4: the same kernels on a P5 network (3 Detect levels, max stride 32, as MDv1000-spruce =
    YOLOv5s) with nc = 5 (no = 10, Detect N = 30) at a non-square 320x224 input: every layer against the
    bf16-emulating oracle, the decoded predictions, and NMS bit-exact on those predictions.
    """
    from megadetector_amd import weights_io, yolo_yaml
    from megadetector_amd.hip_backend import HipContext
    W = weights_io.synthetic_weights(yolo_yaml.YOLOV5N_P5_TEST, seed=2)
    assert W.max_stride == 32
    ctx = HipContext(W, device=0, max_batch=3, max_h=320, max_w=320)
    try:
        hh, ww = 320, 224
        imgs = PU.structured_images(3, hh, ww, seed=15)
        ctx.preprocess(imgs, _identity_geoms(imgs), hh, ww)
        ctx.forward(3, hh, ww)
        x, _ = PU.oracle_input(imgs, 320, 32)
        assert tuple(x.shape[2:]) == (hh, ww)
        keep = {}
        pred_ref, _ = PU.oracle_forward(W, x, emulate_bf16=True, keep=keep)
        bad = []
        for i in sorted(keep):
            e = PU.rel_err(ctx.read_layer(i, 3), keep[i].numpy())
            if e[0] > LAYER_MAX_TOL or e[1] > LAYER_MEAN_TOL:
                bad.append((i,) + e)
        assert not bad, bad
        pred = ctx.read_predictions(3, hh, ww)
        assert pred.shape == tuple(pred_ref.shape) == (3, 3 * (40 * 28 + 20 * 14 + 10 * 7), 10)
        emax, emean = PU.rel_err(pred[..., :4], pred_ref[..., :4].numpy())
        assert emax < 3e-2 and emean < 8e-3
        assert np.abs(pred[..., 4:] - pred_ref[..., 4:].numpy()).max() < 2e-2
        out, counts = ctx.nms(3, 1e-5, 0.45, 300)
        ref = O.nms(torch.from_numpy(pred), conf_thres=1e-5, iou_thres=0.45, max_det=300)
        for i in range(3):
            assert counts[i] == ref[i].shape[0]
            np.testing.assert_array_equal(out[i, :counts[i]], ref[i].numpy())
        assert counts.sum() > 0
        assert set(np.unique(out[0, :counts[0], 5]).astype(int)) <= set(range(5))
    finally:
        ctx.close()


def test_tile_configurations_agree_bitwise(n6):
    """
    Every implicit-GEMM tile configuration accumulates K in the same order: outputs must be
    identical.  The row-patch kernel sums in (channel group, tap) order instead: same result up to
    fp32 summation order, i.e. within the layer tolerance of the bf16-emulating oracle.
    """
    W, ctx = n6
    imgs = PU.random_images(2, 192, 256, seed=8)
    ctx.preprocess(imgs, _identity_geoms(imgs), 192, 256)
    ctx.forward(2, 192, 256)
    base = ctx.read_predictions(2, 192, 256).copy()
    convs = [o['op'] for o in ctx.op_infos() if o['kind'] == 0]
    try:
        for cfg in range(ctx.num_conv_cfgs()):
            switched = 0
            for op in convs:
                if ctx.op_supports_cfg(op, cfg):       # the later main loops do not take the stem
                    ctx.set_op_cfg(op, cfg)
                    switched += 1
                else:
                    ctx.set_op_cfg(op, -1)
            if switched == 0:
                # only the row-patch / row-segment / e4m3 kernels and the dedicated stem kernel (x6 stem only:
                # tests/test_gpu_headline.py) may find nothing in this network
                assert not ctx.cfg_is_bitwise(cfg) or ctx.conv_cfg_name(cfg).startswith('stem:'), cfg
                continue
            ctx.forward(2, 192, 256)
            got = ctx.read_predictions(2, 192, 256)
            if ctx.cfg_is_bitwise(cfg):
                np.testing.assert_array_equal(got, base, err_msg='cfg {}'.format(cfg))
            else:
                emax, emean = PU.rel_err(got[..., :4], base[..., :4])
                assert emax < LAYER_MAX_TOL and emean < LAYER_MEAN_TOL, (cfg, emax, emean)
                assert np.abs(got[..., 4:] - base[..., 4:]).max() < E2E_CONF_TOL_BF16_ORACLE, cfg
    finally:
        for op in convs:
            ctx.set_op_cfg(op, -1)


# one-workgroup-per-CU tiles with 80x80 wave tiles: they take layers whose channel count is a multiple of their BN only
# (160 / 320: the x6 widths) and are covered on that topology by tests/test_gpu_headline.py
EIGHT_WAVE_TILES = ('v5:run160x320', 'v5:run320x160')


def test_row_patch_conv_matches_implicit_gemm_and_oracle():
    """
    The row-patch direct convolution (conv_v4.cpp) and the row-segment kernels (conv_v5.cpp), each on
    every op it supports of a wider test network at 640x1280 (80x160 and 40x80 maps, 64..128 input channels incl. a half-full channel group):
    against the implicit-GEMM result (same arithmetic, different fp32 summation order) and, layer by
    layer, against the bf16-emulating oracle.
    """
    from megadetector_amd import weights_io, yolo_yaml
    from megadetector_amd.hip_backend import HipContext
    W = weights_io.synthetic_weights(yolo_yaml.YOLOV5S6_TEST, seed=3)
    HH, WW = 640, 1280          # 80x160 and 40x80 maps: divisible by both patch tile widths (40, 32) where C >= 64
    ctx = HipContext(W, device=0, max_batch=2, max_h=HH, max_w=WW)
    try:
        imgs = PU.structured_images(2, HH, WW, seed=21)
        ctx.preprocess(imgs, _identity_geoms(imgs), HH, WW)
        ctx.forward(2, HH, WW)
        base = ctx.read_predictions(2, HH, WW).copy()
        convs = [o['op'] for o in ctx.op_infos() if o['kind'] == 0]
        # (the e4m3 family f8:* takes fp8 operands only: tests/test_gpu_fp8.py; the strip kernel v5:strip* takes the
        # 80 -> 80 channel layers of the x6 stack only, the stride-2 row-run kernel v7:* layers with a multiple of 160
        # output channels only: tests/test_gpu_headline.py; dev:* are stamped developer variants that run only from tools/convbench)
        patch_cfgs = [c for c in range(ctx.num_conv_cfgs())
                      if not ctx.cfg_is_bitwise(c) and not ctx.conv_cfg_name(c).startswith(('f8:', 'v5:strip', 'dev:', 'v7:') + EIGHT_WAVE_TILES)]
        assert patch_cfgs, 'no row-patch configuration in this build'
        x, _ = PU.oracle_input(imgs, WW, 64)
        assert tuple(x.shape[2:]) == (HH, WW)
        keep = {}
        PU.oracle_forward(W, x, emulate_bf16=True, keep=keep)
        for cfg in patch_cfgs:
            switched = [op for op in convs if ctx.op_supports_cfg(op, cfg)]
            assert len(switched) >= 2, (cfg, len(switched))
            for op in convs:
                ctx.set_op_cfg(op, cfg if op in switched else -1)
            ctx.forward(2, HH, WW)
            infos = ctx.op_infos()
            assert all(infos[op]['cfg'] == cfg for op in switched)
            got = ctx.read_predictions(2, HH, WW)
            emax, emean = PU.rel_err(got[..., :4], base[..., :4])
            assert emax < LAYER_MAX_TOL and emean < LAYER_MEAN_TOL, (cfg, emax, emean)
            assert np.abs(got[..., 4:] - base[..., 4:]).max() < E2E_CONF_TOL_BF16_ORACLE, cfg
            bad = []
            for i in sorted(keep):
                g = ctx.read_layer(i, 2)
                e = PU.rel_err(g, keep[i].numpy())
                if e[0] > LAYER_MAX_TOL or e[1] > LAYER_MEAN_TOL:
                    bad.append((i,) + e)
            assert not bad, bad
            # batch-composition invariance stays bitwise with the patch kernel
            ctx.preprocess([imgs[1]], _identity_geoms([imgs[1]]), HH, WW)
            ctx.forward(1, HH, WW)
            np.testing.assert_array_equal(ctx.read_predictions(1, HH, WW)[0], got[1])
            ctx.preprocess(imgs, _identity_geoms(imgs), HH, WW)
    finally:
        ctx.close()


def test_batch_composition_invariance(n6):
    """reference md_tests.py:1235-1238: batched == unbatched (here: bit-identical)."""
    W, ctx = n6
    imgs = PU.random_images(4, 256, 256, seed=11)
    ctx.preprocess(imgs, _identity_geoms(imgs), 256, 256)
    ctx.forward(4, 256, 256)
    full = ctx.read_predictions(4, 256, 256).copy()
    for i in (0, 3):
        ctx.preprocess([imgs[i]], _identity_geoms([imgs[i]]), 256, 256)
        ctx.forward(1, 256, 256)
        np.testing.assert_array_equal(ctx.read_predictions(1, 256, 256)[0], full[i])


def test_batch_composition_invariance_with_measured_table():
    """The shipped tile table (megadetector_amd/tuned_cfgs.json, measured at batch 32 / 1280x1280) applied to
    the MDv5 topology at 640x640 and 384x640: kernels of both summation-order families are in use, and an
    image's predictions are bit-identical whether it travels alone or in a batch (the table is looked up
    per image, never per call)."""
    from megadetector_amd import weights_io, yolo_yaml
    from megadetector_amd.hip_backend import HipContext
    W = weights_io.synthetic_weights(yolo_yaml.YOLOV5X6_MD, seed=0)
    ctx = HipContext(W, device=0, max_batch=3, max_h=640, max_w=640)
    try:
        for (hh, ww) in ((640, 640), (384, 640)):
            imgs = PU.structured_images(3, hh, ww, seed=41)
            ctx.preprocess(imgs, _identity_geoms(imgs), hh, ww)
            ctx.forward(3, hh, ww)
            full = ctx.read_predictions(3, hh, ww).copy()
            cfgs3 = [o['cfg'] for o in ctx.op_infos() if o['kind'] == 0]
            if (hh, ww) == (640, 640):
                assert any(not ctx.cfg_is_bitwise(c) for c in cfgs3), 'the table selects no row-segment/row-patch kernel'
                assert any(ctx.cfg_is_bitwise(c) for c in cfgs3)
            for i in (0, 2):
                ctx.preprocess([imgs[i]], _identity_geoms([imgs[i]]), hh, ww)
                ctx.forward(1, hh, ww)
                cfgs1 = [o['cfg'] for o in ctx.op_infos() if o['kind'] == 0]
                assert [ctx.cfg_is_bitwise(c) for c in cfgs1] == [ctx.cfg_is_bitwise(c) for c in cfgs3]
                np.testing.assert_array_equal(ctx.read_predictions(1, hh, ww)[0], full[i])
    finally:
        ctx.close()


def test_cabi_error_returns_do_not_poison_the_context(n6):
    """C ABI conventions (include/mdhip.h): bad arguments return an error code with a message, nothing
    throws or exits, and the context keeps working afterwards."""
    from megadetector_amd._lib import HipError
    W, ctx = n6
    imgs = PU.random_images(2, 256, 256, seed=77)
    ctx.preprocess(imgs, _identity_geoms(imgs), 256, 256)
    ctx.forward(2, 256, 256)
    good = ctx.read_predictions(2, 256, 256).copy()
    with pytest.raises(HipError, match='batch'):
        ctx.forward(ctx.max_batch + 1, 256, 256)
    with pytest.raises(HipError, match='multiple of the model stride'):
        ctx.forward(1, 250, 256)
    with pytest.raises(HipError, match='exceeds the planned'):
        ctx.forward(1, 640, 640)
    with pytest.raises(HipError, match='does not fit'):
        ctx.preprocess(imgs, [(256, 256, 300, 256, 0, 0)] * 2, 256, 256)
    with pytest.raises(HipError):
        ctx.set_op_cfg(0, 10 ** 6)
    with pytest.raises(HipError):
        ctx.nms_on(np.zeros((1, 10, 8), np.float32), 0.1, 0.45, max_det=10 ** 6)
    ctx.preprocess(imgs, _identity_geoms(imgs), 256, 256)
    ctx.forward(2, 256, 256)
    np.testing.assert_array_equal(ctx.read_predictions(2, 256, 256), good)


# ---------------------------------------------------------------------------------------
# NMS: bit-exact against the reference fixtures and the oracle
# ---------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def nms_ctx():
    """a context whose NMS capacity is the full 1280x1280 anchor count (102000)"""
    from megadetector_amd import weights_io, yolo_yaml
    from megadetector_amd.hip_backend import HipContext
    W = weights_io.synthetic_weights(yolo_yaml.YOLOV5N6_TEST, seed=1)
    ctx = HipContext(W, device=0, max_batch=4, max_h=1280, max_w=1280)
    yield ctx
    ctx.close()


def _canon(a):
    a = np.asarray(a, dtype=np.float32)
    if a.shape[0] == 0:
        return a
    return a[np.lexsort((a[:, 5], a[:, 3], a[:, 2], a[:, 1], a[:, 0], -a[:, 4]))]


@pytest.mark.parametrize('case', ['synthetic', 'identical', 'rand_a', 'rand_b', 'rand_c', 'empty'])
def test_nms_matches_reference_fixture(n6, case):
    W, ctx = n6
    blob = np.load(os.path.join(GOLDEN, 'nms_reference.npz'))
    pred = blob[case + '/pred']
    ct, it, md = blob[case + '/params']
    out, counts = ctx.nms_on(pred, float(ct), float(it), int(md))
    for i in range(pred.shape[0]):
        ref = blob['{}/out{}'.format(case, i)]
        assert counts[i] == ref.shape[0]
        np.testing.assert_array_equal(_canon(out[i, :counts[i]]), _canon(ref))


@pytest.mark.parametrize('seed,n,a,ct', [(21, 2, 4000, 0.02), (22, 1, 15000, 1e-5), (23, 3, 9000, 0.3)])
def test_nms_matches_oracle_order_exact(nms_ctx, seed, n, a, ct):
    from parity_util import random_predictions
    ctx = nms_ctx
    pred = random_predictions(seed, n, a, n_clusters=25)
    out, counts = ctx.nms_on(pred.numpy(), ct, 0.45, 300)
    ref = O.nms(pred, conf_thres=ct, iou_thres=0.45, max_det=300)
    for i in range(n):
        assert counts[i] == ref[i].shape[0]
        np.testing.assert_array_equal(out[i, :counts[i]], ref[i].numpy())


@pytest.mark.parametrize('q', [0.99, 0.90])
def test_nms_stress_input_full_size_exact(nms_ctx, q):
    """SURVEY.md 8(d)'s NMS stress tensor (tools/nms_bench.py: 102000 anchors, obj ~ Beta(0.05, 1), 50 box
    clusters) at thresholds that let 1 % / 10 % of the anchors through: exact against the oracle."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    from nms_bench import stress_predictions
    ctx = nms_ctx
    a = ctx.num_anchors(1280, 1280)
    pred = stress_predictions(1, a, seed=0)
    thr = float(np.quantile(pred[..., 4] * pred[..., 5:8].max(-1), q))
    out, counts = ctx.nms_on(pred, thr, 0.45, 300)
    ref = O.nms(torch.from_numpy(pred), conf_thres=thr, iou_thres=0.45, max_det=300)[0].numpy()
    assert counts[0] == ref.shape[0] > 0
    np.testing.assert_array_equal(out[0, :counts[0]], ref)


def test_nms_properties_full_size(nms_ctx):
    """BASELINE size (102000 anchors/image, every one a candidate): sortedness, class-wise
    non-overlap and idempotence"""
    ctx = nms_ctx
    a = ctx.num_anchors(1280, 1280)
    assert a == 102000
    rng = np.random.default_rng(5)
    pred = np.zeros((2, a, 8), dtype=np.float32)
    pred[..., 0:2] = rng.random((2, a, 2)) * 1280
    pred[..., 2:4] = 30 + rng.random((2, a, 2)) * 400
    pred[..., 4] = rng.random((2, a)) * 0.9 + 0.05
    pred[..., 5:] = rng.random((2, a, 3))
    out, counts = ctx.nms_on(pred, 1e-5, 0.45, 300)
    for i in range(2):
        k = counts[i]
        assert 0 < k <= 300
        d = out[i, :k]
        assert np.all(np.diff(d[:, 4]) <= 0)
        for c in range(3):
            b = d[d[:, 5] == c][:, :4]
            for p in range(len(b)):
                for q in range(p + 1, len(b)):
                    iou = O.get_iou([b[p, 0], b[p, 1], b[p, 2] - b[p, 0], b[p, 3] - b[p, 1]],
                                    [b[q, 0], b[q, 1], b[q, 2] - b[q, 0], b[q, 3] - b[q, 1]])
                    assert iou <= 0.45 + 1e-6
        # idempotence: feeding the survivors back keeps all of them in the same order
        again = np.zeros((1, k, 8), dtype=np.float32)
        again[0, :, 0] = (d[:, 0] + d[:, 2]) / 2
        again[0, :, 1] = (d[:, 1] + d[:, 3]) / 2
        again[0, :, 2] = d[:, 2] - d[:, 0]
        again[0, :, 3] = d[:, 3] - d[:, 1]
        again[0, :, 4] = 1.0
        again[0, np.arange(k), 5 + d[:, 5].astype(int)] = d[:, 4]
        out2, c2 = ctx.nms_on(again, 1e-6, 0.46, 300)
        assert c2[0] == k
        np.testing.assert_array_equal(out2[0, :k, 5], d[:, 5])


# ---------------------------------------------------------------------------------------
# end to end through the detector seam
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', [None, 'bf16'])
def test_detector_end_to_end_vs_oracle(dtype):
    """dtype None = the detector's default storage type (fp16), 'bf16' = the benchmarked throughput mode"""
    from megadetector_amd import weights_io, yolo_yaml
    from megadetector_amd.detector import HIPDetector, DEFAULT_DTYPE
    W = weights_io.synthetic_weights(yolo_yaml.YOLOV5N6_TEST, seed=1)
    opts = {'batch_size': 4, 'max_image_size': 320}
    if dtype:
        opts['dtype'] = dtype
    det = HIPDetector(W, opts)
    storage = dtype or DEFAULT_DTYPE
    assert det._ctx.dtype == storage and DEFAULT_DTYPE == 'fp16'
    emulation = {True: True if storage == 'bf16' else 'fp16', False: False}
    conf_tol = {True: E2E_CONF_TOL_BF16_ORACLE, False: E2E_CONF_TOL_FP32_ORACLE} if storage == 'bf16' else E2E_CONF_TOL_FP16
    det.default_image_size = 320
    det.letterbox_stride = 64
    imgs = PU.structured_images(3, 240, 320, seed=31) + PU.structured_images(1, 300, 200, seed=32)
    ids = ['a.jpg', 'b.jpg', 'c.jpg', 'd.jpg']
    thr = 1e-5
    res = det.generate_detections_one_batch(imgs, ids, detection_threshold=thr)
    assert [r['file'] for r in res] == ids
    assert all('failure' not in r for r in res)
    ctx = det._ctx
    n_hi = {True: 0, False: 0}
    worst = {True: 0.0, False: 0.0}
    for im, r in zip(imgs, res):
        x, infos = PU.oracle_input([im], 320, 64)
        h, w = x.shape[2:]
        # (1) exact: NMS + rescale + formatting of the HIP path == the reference's statements
        #     (oracle) applied to the very predictions the HIP conv stack produced
        one = det.generate_detections_one_image(im, 'x.jpg', detection_threshold=thr)
        assert one['detections'] == r['detections']          # batch == single image, bit for bit
        pred_hip = torch.from_numpy(ctx.read_predictions(1, h, w))
        ref_same = PU.oracle_detections(pred_hip, infos, (h, w), thr)[0]
        assert r['detections'] == ref_same['detections']
        assert r['max_detection_conf'] == ref_same['max_detection_conf']
        # (2) tolerance: against the oracle's own forward (bf16-emulating and fp32 = reference)
        for emulate in (True, False):
            pred, _ = PU.oracle_forward(W, x, emulate_bf16=emulation[emulate])
            e_box = PU.rel_err(pred_hip[..., :4].numpy(), pred[..., :4].numpy())
            e_conf = float(np.abs(pred_hip[..., 4:].numpy() - pred[..., 4:].numpy()).max())
            if storage == 'bf16':
                assert e_box[0] < 5e-2 and e_box[1] < 1e-2 and e_conf < 6e-2, (emulate, e_box, e_conf)
            else:
                assert e_box[0] < 6e-3 and e_box[1] < 1.5e-3 and e_conf < F16_CONF_TOL_FP32_ORACLE, (emulate, e_box, e_conf)
            # Greedy NMS is discontinuous at near-ties, so survivors are not compared one to one;
            # instead every confident HIP survivor must be a legitimate candidate in the oracle's
            # own predictions: same class, IoU >= 0.85 (md_tests.py:124) and |dconf| within the
            # tolerance stated at the top of this file for that oracle
            det_hip, cnt = ctx.nms(1, thr, 0.45, 300)
            d = det_hip[0, :cnt[0]]
            d = d[d[:, 4] >= 0.1]
            pp = pred[0].numpy()
            cconf = pp[:, 5:] * pp[:, 4:5]
            ccls = cconf.argmax(1)
            cbest = cconf.max(1)
            cx1, cy1 = pp[:, 0] - pp[:, 2] / 2, pp[:, 1] - pp[:, 3] / 2
            cx2, cy2 = pp[:, 0] + pp[:, 2] / 2, pp[:, 1] + pp[:, 3] / 2
            for row in d:
                n_hi[emulate] += 1
                iw = np.clip(np.minimum(cx2, row[2]) - np.maximum(cx1, row[0]), 0, None)
                ih = np.clip(np.minimum(cy2, row[3]) - np.maximum(cy1, row[1]), 0, None)
                inter = iw * ih
                iou = inter / ((cx2 - cx1) * (cy2 - cy1) + (row[2] - row[0]) * (row[3] - row[1]) - inter)
                cand = (ccls == int(row[5])) & (iou >= 0.85)
                assert cand.any(), ('no oracle candidate for a confident HIP detection', emulate, row)
                worst[emulate] = max(worst[emulate], float(np.abs(cbest[cand] - row[4]).min()))
    print('end-to-end ({}): confident survivors {} ; worst |dconf| vs storage-emulating oracle {:.4f}, vs fp32 oracle {:.4f}'.format(
        storage, n_hi, worst[True], worst[False]))
    assert n_hi[True] > 0
    assert worst[True] <= conf_tol[True], worst
    assert worst[False] <= conf_tol[False], worst
    # a broken image must not kill the batch (reference pytorch_detector.py:1212-1222)
    res = det.generate_detections_one_batch([imgs[0], np.zeros((4, 4), np.uint8)], ['ok.jpg', 'bad.jpg'])
    assert res[1]['failure'] == 'image access failure' and res[1]['detections'] is None
    assert res[0]['detections'] is not None


def test_checkpoint_file_through_load_detector(tmp_path):
    """
    A .pt file with the pickle layout of md_v5a.0.0.pt (tests/fake_yolov5.py: whole fp16 module, BatchNorm
    not fused, `models.*` classes that are not importable here) through the reference's entry point
    `run_detector.load_detector(path)`: the HIP predictions against the file's own nn.Module forward
    (fp32, = what the reference computes from this file), and detections == the reference's
    post-processing statements applied to the HIP predictions.
    """
    import fake_yolov5 as FY
    from megadetector_amd import run_detector, yolo_yaml
    model = FY.build_model(yolo_yaml.YOLOV5N6_TEST, seed=5)
    path = str(tmp_path / 'md_fake.pt')
    FY.save_checkpoint(model, path)
    ref_model = model.half().float()
    imgs = PU.structured_images(2, 256, 320, seed=51)
    x, infos = PU.oracle_input(imgs, 320, 64)
    with torch.no_grad():
        ref_pred = ref_model(x).numpy()
    FY.uninstall()
    det = run_detector.load_detector(path, detector_options={'batch_size': 2, 'max_image_size': 320})
    det.default_image_size = 320
    res = det.generate_detections_one_batch(imgs, ['a.jpg', 'b.jpg'], detection_threshold=1e-5)
    assert all('failure' not in r for r in res), res
    h, w = x.shape[2:]
    got = det._ctx.read_predictions(2, h, w)
    assert got.shape == ref_pred.shape
    e_box = PU.rel_err(got[..., :4], ref_pred[..., :4])
    e_conf = float(np.abs(got[..., 4:] - ref_pred[..., 4:]).max())
    # default storage type (fp16): the reference's own bar against the module's fp32 forward
    assert det._ctx.dtype == 'fp16'
    print('checkpoint through load_detector: box {:.2e}/{:.2e} conf {:.2e}'.format(e_box[0], e_box[1], e_conf))
    assert e_box[0] < 6e-3 and e_box[1] < 1.5e-3 and e_conf < 0.005, (e_box, e_conf)
    ref_same = PU.oracle_detections(torch.from_numpy(got), infos, (h, w), 1e-5)
    for r, q in zip(res, ref_same):
        assert r['detections'] == q['detections']
    assert any(len(r['detections']) > 0 for r in res)


# ---------------------------------------------------------------------------------------
# the batch driver on the real HIP detector: orchestration modes, JSON, failures
# ---------------------------------------------------------------------------------------
def test_batch_driver_modes_identical_on_gpu(tmp_path):
    """
    reference md_tests.py:1251,1267,1283 (queue / preprocess-queue / checkpoint runs must give the same
    JSON as the plain run) and :1235-1238 (batched vs unbatched within 0.01 -- here: identical) with the
    HIP detector behind load_detector, on JPEGs of two different shapes plus one unreadable file.
    """
    import json
    from PIL import Image
    from megadetector_amd import run_detector_batch as RDB
    rng = np.random.default_rng(5)
    names = []
    for i in range(7):
        shape = (120, 160, 3) if i % 2 else (150, 100, 3)
        p = tmp_path / ('img_%02d.jpg' % i)
        base = rng.integers(0, 256, (shape[0] // 10 + 1, shape[1] // 10 + 1, 3), dtype=np.uint8)
        img = np.kron(base, np.ones((10, 10, 1), dtype=np.uint8))[:shape[0], :shape[1]]
        Image.fromarray(img).save(p, quality=95)
        names.append(str(p))
    bad = tmp_path / 'broken.jpg'
    bad.write_bytes(b'not a jpeg')
    names.append(str(bad))
    model = 'synthetic:YOLOV5N6_TEST:1'
    opts = {'batch_size': 4}      # default 1280 px letterbox, as in the reference's batch mode

    def run(**kw):
        res = RDB.load_and_run_detector_batch(model, names, quiet=True, detector_options=dict(opts), **kw)
        return json.loads(json.dumps(sorted(res, key=lambda r: r['file'])))

    plain = run()
    assert len(plain) == len(names)
    assert [r for r in plain if r['file'].endswith('broken.jpg')][0].get('failure') == 'image access failure'
    ok = [r for r in plain if 'failure' not in r]
    assert len(ok) == 7 and all(isinstance(r['detections'], list) for r in ok)
    for r in ok:
        for d in r['detections']:
            assert d['category'] in ('1', '2', '3') and 0.0 <= d['conf'] <= 1.0 and len(d['bbox']) == 4
    for kw in (dict(batch_size=4), dict(use_image_queue=True), dict(use_image_queue=True, batch_size=3),
               dict(use_image_queue=True, batch_size=3, preprocess_on_image_queue=True, loader_workers=2),
               # spawned loader processes + page-locked shared-memory ring + pipelined detector
               dict(use_image_queue=True, use_threads_for_queue=False, batch_size=3, loader_workers=2),
               dict(use_image_queue=True, use_threads_for_queue=False, batch_size=1, loader_workers=2)):
        assert run(**kw) == plain, kw
    out = tmp_path / 'out.json'
    RDB.write_results_to_file(plain, str(out), detector_file=model)
    j = json.load(open(out))
    assert j['info']['format_version'] == '1.6' and len(j['images']) == len(names)
    assert j['detection_categories'] == {'1': 'animal', '2': 'person', '3': 'vehicle'}


def test_video_frames_batched_equal_frame_by_frame_on_gpu():
    """SURVEY.md 8(f) N2 (megadetector_amd/process_video.py): 11 frames of one shape through the HIP detector,
    batched and pipelined, give exactly what the reference's one-frame-at-a-time callback loop gives."""
    import json
    from megadetector_amd import process_video as PV, run_detector
    det = run_detector.load_detector('synthetic:YOLOV5N6_TEST:1', detector_options={'batch_size': 4, 'max_image_size': 320})
    det.default_image_size = 320
    frames = PU.structured_images(11, 180, 320, seed=91)
    one = [det.generate_detections_one_image(f, PV.frame_number_to_filename(i), detection_threshold=1e-5)
           for i, f in enumerate(frames) if i % 2 == 0]
    got = PV.run_detector_on_frames(det, PV.ArrayFrameSource(frames, frame_rate=15.0), every_n_frames=2, batch_size=4,
                                    detection_threshold=1e-5)
    assert got['frame_filenames'] == [PV.frame_number_to_filename(i) for i in range(0, 11, 2)]
    assert json.loads(json.dumps(got['results'])) == json.loads(json.dumps(one))
    assert any(r['detections'] for r in got['results'])
    md = PV.run_detector_on_videos(det, [('cam/clip.mp4', frames)], open_source=lambda fr: PV.ArrayFrameSource(fr, 15.0),
                                   every_n_frames=2, batch_size=4, detection_threshold=1e-5)
    im = PV.video_results_to_md_format(md)[0]
    assert im['frames_processed'] == [0, 2, 4, 6, 8, 10] and im['frame_rate'] == 15.0
    assert sum(len(r['detections']) for r in one) == len(im['detections'])


def test_test_time_augmentation_matches_oracle(n6):
    """
    SURVEY.md 8(f) N3, `augment=True` (reference md_tests.py:917-930 exercises it): mdhip_forward_tta against the
    oracle's restatement of yolov5 `_forward_augment` (bf16-emulating; parity unpinned, oracle/yolov5.py): anchor
    count, the first pass bit-identical to the plain forward, de-scaled / un-flipped boxes and confidences of the
    scaled passes within the layer tolerances, NMS on the concatenated predictions exact, and the detector
    seam (`generate_detections_one_image(..., augment=True)`).
    """
    W, ctx = n6
    hh, ww = 256, 320
    imgs = PU.structured_images(2, hh, ww, seed=33)
    ctx.preprocess(imgs, _identity_geoms(imgs), hh, ww)
    ctx.forward(2, hh, ww)
    plain = ctx.read_predictions(2).copy()
    ctx.forward_tta(2, hh, ww)
    got = ctx.read_predictions(2).copy()
    x, infos = PU.oracle_input(imgs, 320, 64)
    assert tuple(x.shape[2:]) == (hh, ww)
    ref, _ = PU.oracle_forward(W, x, emulate_bf16=True, augment=True)
    ref = ref.numpy()
    assert got.shape == ref.shape and got.shape[1] == ctx.last_num_anchors() > plain.shape[1]
    a1 = plain.shape[1] - plain.shape[1] // 85
    np.testing.assert_array_equal(got[:, :a1], plain[:, :a1])                 # pass 1 = the plain forward
    emax, emean = PU.rel_err(got[..., :4], ref[..., :4])
    assert emax < 3e-2 and emean < 8e-3, (emax, emean)
    assert np.abs(got[..., 4:] - ref[..., 4:]).max() < 3e-2
    # flipped pass: box centres land inside the image again
    assert got[..., 0].min() > -128 and got[..., 0].max() < ww + 128
    out, counts = ctx.nms(2, 1e-5, 0.45, 300)
    want = O.nms(torch.from_numpy(got), conf_thres=1e-5, iou_thres=0.45, max_det=300)
    for i in range(2):
        assert counts[i] == want[i].shape[0] > 0
        np.testing.assert_array_equal(out[i, :counts[i]], want[i].numpy())
    # the context still serves the plain path afterwards
    ctx.forward(2, hh, ww)
    np.testing.assert_array_equal(ctx.read_predictions(2), plain)
    # through the detector seam
    from megadetector_amd.detector import HIPDetector
    det = HIPDetector(W, {'batch_size': 2, 'max_image_size': 320})
    det.default_image_size = 320
    r0 = det.generate_detections_one_image(imgs[0], 'a.jpg', detection_threshold=1e-5)
    r1 = det.generate_detections_one_image(imgs[0], 'a.jpg', detection_threshold=1e-5, augment=True)
    assert 'failure' not in r1 and r1['detections'] and r1 != r0
    want1 = PU.oracle_detections(torch.from_numpy(det._ctx.read_predictions(1)), infos[:1], (hh, ww), 1e-5)[0]
    assert r1['detections'] == want1['detections']
    # ... and through the pipelined interface of the batch driver (start_batch / finish_batch take `augment` since
    # round 3): the same detections as the synchronous call, plain and augmented
    both = det.generate_detections_one_batch(list(imgs), ['a.jpg', 'b.jpg'], detection_threshold=1e-5, augment=True)
    t = det.start_batch(list(imgs), ['a.jpg', 'b.jpg'], detection_threshold=1e-5, augment=True)
    assert det.finish_batch(t) == both and both[0]['detections'] == r1['detections']
    t = det.start_batch(list(imgs), ['a.jpg', 'b.jpg'], detection_threshold=1e-5)
    assert det.finish_batch(t)[0]['detections'] == r0['detections']


@pytest.mark.parametrize('shape', [(300, 400), (512, 384), (97, 211), (640, 640), (1000, 750), (256, 256), (150, 260)])
def test_modern_mode_preprocess_bit_exact(n6, shape):
    """compatibility_mode 'modern' (reference pytorch_detector.py:1036-1109): INTER_AREA / INTER_LINEAR resize to the
    long side + centred padding into the modern target shape, on the device, against the oracle's restatement
    (integer and fractional shrink factors, growing, no resize)."""
    from megadetector_amd.postprocess import modern_geometry
    W, ctx = n6
    size = 256
    imgs = PU.structured_images(2, shape[0], shape[1], seed=shape[1])
    m = modern_geometry(shape, image_size=size, stride=64)
    g = m['letterbox']
    h, w = g['out_hw']
    assert (h, w) == m['target_shape']
    geoms = [(shape[0], shape[1], m['resized_hw'][0], m['resized_hw'][1], g['top'], g['left'], m['interp'])] * 2
    ctx.preprocess(imgs, geoms, h, w)
    got = ctx.read_input(2, h, w)
    ref = [O.preprocess_image_modern(im, image_size=size, stride=64)['img_processed'] for im in imgs]
    assert ref[0].shape[:2] == (h, w)
    np.testing.assert_array_equal(got, PU.bf16_round_np(O.to_batch_tensor(ref).numpy()))


def test_modern_mode_through_the_detector():
    """detector with compatibility_mode='modern': NMS IoU 0.6, ratio_pad rescale, rounding -- detections equal the
    oracle's modern post-processing applied to the HIP predictions; and they differ from classic (md_tests.py:1333)."""
    from megadetector_amd import weights_io, yolo_yaml
    from megadetector_amd.detector import HIPDetector
    W = weights_io.synthetic_weights(yolo_yaml.YOLOV5N6_TEST, seed=1)
    img = PU.structured_images(1, 450, 600, seed=77)[0]
    res = {}
    for mode in ('classic', 'modern'):
        det = HIPDetector(W, {'batch_size': 2, 'max_image_size': 320, 'compatibility_mode': mode})
        det.default_image_size = 320
        res[mode] = det.generate_detections_one_image(img, 'a.jpg', detection_threshold=1e-5)
        assert 'failure' not in res[mode] and res[mode]['detections']
        if mode == 'modern':
            info = O.preprocess_image_modern(img, image_size=320, stride=64)
            hh, ww = info['img_processed'].shape[:2]
            assert (hh, ww) == (320, 384)
            pred = torch.from_numpy(det._ctx.read_predictions(1))
            d = O.nms(pred, conf_thres=1e-5, iou_thres=0.6, max_det=300)[0]
            want, want_max = O.format_detections(d, (hh, ww), info['img_original'].shape, info['scaling_shape'], 1e-5,
                                                 modern=True, letterbox_pad=info['letterbox_pad'])
            assert res[mode]['detections'] == want and res[mode]['max_detection_conf'] == want_max
        det._ctx.close()
    assert res['classic']['detections'] != res['modern']['detections']


def test_threshold_one_gives_empty_detection_lists_and_ragged_batches():
    """nothing above the threshold is an empty list with max_detection_conf 0.0 (not a failure); a batch larger than
    the context's batch size and of three different shapes is split and regrouped transparently"""
    from megadetector_amd import weights_io, yolo_yaml
    from megadetector_amd.detector import HIPDetector
    W = weights_io.synthetic_weights(yolo_yaml.YOLOV5N6_TEST, seed=1)
    det = HIPDetector(W, {'batch_size': 2, 'max_image_size': 320})
    det.default_image_size = 320
    imgs = (PU.structured_images(3, 240, 320, seed=1) + PU.structured_images(2, 320, 200, seed=2) +
            PU.structured_images(2, 100, 100, seed=3))
    ids = ['i%d.jpg' % i for i in range(len(imgs))]
    res = det.generate_detections_one_batch(imgs, ids, detection_threshold=1.0)
    assert [r['file'] for r in res] == ids
    assert all(r['detections'] == [] and r['max_detection_conf'] == 0.0 and 'failure' not in r for r in res)
    full = det.generate_detections_one_batch(imgs, ids, detection_threshold=1e-5)
    one_by_one = [det.generate_detections_one_image(im, i, detection_threshold=1e-5) for im, i in zip(imgs, ids)]
    assert full == one_by_one and any(r['detections'] for r in full)


def test_full_size_configuration_batch_invariance():
    """BASELINE.json configs[1] at full size -- MDv5 topology, batch 32, 1280x1280, the shipped tile table: every
    value finite, and images 0 / 17 / 31 of the batch bit-identical to their single-image forward (a size-
    independent property that catches 32-bit offset overflows and tile-edge bugs the small tests cannot)."""
    from megadetector_amd import weights_io, yolo_yaml
    from megadetector_amd.hip_backend import HipContext
    W = weights_io.synthetic_weights(yolo_yaml.YOLOV5X6_MD, seed=0)
    B, S = 32, 1280
    ctx = HipContext(W, device=0, max_batch=B, max_h=S, max_w=S)
    try:
        g = torch.Generator(device='cuda')
        g.manual_seed(1)
        x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, device='cuda', generator=g)
        ptrs = [int(x[i].data_ptr()) for i in range(B)]
        ctx.preprocess(ptrs, [(S, S, S, S, 0, 0)] * B, S, S)
        ctx.forward(B, S, S)
        full = ctx.read_predictions(B)
        assert full.shape == (B, 102000, 8) and np.isfinite(full).all()
        for i in (0, 17, 31):
            ctx.preprocess([ptrs[i]], [(S, S, S, S, 0, 0)], S, S)
            ctx.forward(1, S, S)
            np.testing.assert_array_equal(ctx.read_predictions(1)[0], full[i])
    finally:
        ctx.close()


def test_two_contexts_interleaved_and_recreated(n6):
    """One context per (process, GPU) is the contract, but nothing in the library may be process-global: a second
    context with other weights / storage type, used in alternation with the first and then destroyed, must not
    change the first one's results; a context created afterwards reproduces them."""
    from megadetector_amd import weights_io, yolo_yaml
    from megadetector_amd.hip_backend import HipContext
    W, ctx = n6
    imgs = PU.random_images(2, 192, 256, seed=8)
    geoms = _identity_geoms(imgs)
    ctx.preprocess(imgs, geoms, 192, 256)
    ctx.forward(2, 192, 256)
    base = ctx.read_predictions(2).copy()
    W2 = weights_io.synthetic_weights(yolo_yaml.YOLOV5S6_TEST, seed=9)
    other = HipContext(W2, device=0, dtype='fp16', max_batch=2, max_h=256, max_w=256)
    other.preprocess(imgs, geoms, 192, 256)
    other.forward(2, 192, 256)
    o1 = other.read_predictions(2).copy()
    ctx.forward(2, 192, 256)                      # the first context still holds its own input and plan
    np.testing.assert_array_equal(ctx.read_predictions(2), base)
    other.forward(2, 192, 256)
    np.testing.assert_array_equal(other.read_predictions(2), o1)
    other.close()
    ctx.preprocess(imgs, geoms, 192, 256)
    ctx.forward(2, 192, 256)
    np.testing.assert_array_equal(ctx.read_predictions(2), base)
    again = HipContext(W, device=0, max_batch=4, max_h=320, max_w=320)
    try:
        again.preprocess(imgs, geoms, 192, 256)
        again.forward(2, 192, 256)
        np.testing.assert_array_equal(again.read_predictions(2), base)
    finally:
        again.close()


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
def test_results_do_not_depend_on_unwritten_memory(dtype, monkeypatch):
    """ADVICE r3 (conv_v2 pointwise loader): K-slab tails of 80- / 160- / 480-channel tensors, pad channels and halo reads
    must never see memory nobody wrote.  MDHIP_ARENA_POISON=1 fills the arena with 0xFF bytes (NaN in every storage type)
    at mdhip_create; the x6 topology (channel tails 80 = 64 + 16, 160 = 128 + 32, 480 = 448 + 32), plain and
    augmented, must give finite predictions, bit-identical to a context whose arena started as zeros."""
    from megadetector_amd import weights_io, yolo_yaml
    from megadetector_amd.hip_backend import HipContext
    W = weights_io.synthetic_weights(yolo_yaml.YOLOV5X6_MD, seed=0)
    n, hh, ww = 2, 256, 384
    imgs = PU.structured_images(n, hh, ww, seed=5)
    geoms = [(hh, ww, hh, ww, 0, 0)] * n
    out = {}
    for poison in ('0', '1'):
        monkeypatch.setenv('MDHIP_ARENA_POISON', poison)
        ctx = HipContext(W, device=0, dtype=dtype, max_batch=n, max_h=hh, max_w=ww)
        try:
            ctx.preprocess(imgs, geoms, hh, ww)
            ctx.forward(n, hh, ww)
            plain = ctx.read_predictions(n).copy()
            ctx.forward_tta(n, hh, ww)
            tta = ctx.read_predictions(n).copy()
            ctx.preprocess(imgs[:1], geoms[:1], hh, ww)          # a smaller batch in the same arena: the rest stays poisoned
            ctx.forward(1, hh, ww)
            one = ctx.read_predictions(1).copy()
        finally:
            ctx.close()
        assert np.isfinite(plain).all() and np.isfinite(tta).all() and np.isfinite(one).all(), 'NaN: a kernel read unwritten memory'
        out[poison] = (plain, tta, one)
    for a, b in zip(out['0'], out['1']):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(out['1'][2][0], out['1'][0][0])
