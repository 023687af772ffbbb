"""
The precision claim as a test, on the benchmarked topology.

A checkpoint FILE with the pickle layout of md_v5a.0.0.pt at the x6 widths (tests/fake_yolov5.py: whole fp16 module,
BatchNorm not fused with non-trivial running statistics, per-layer gain 1.3 -- contractive through the head like a
trained network, so that thousands of anchors are confident) goes through the reference's entry point
`run_detector.load_detector(path)` for every storage type; the HIP predictions are compared with the file's own
`nn.Module` forward in fp32 on the CPU (= what the reference computes from this file, pytorch_detector.py:957,1313),
and the FORMATTED detections (the reference's post-processing statements) of the 300 most confident anchors of every
image on both sides, at the reference's bars (md_tests.py:96-100: conf 0.005, coords 0.001).

  fp16 (the detector's default)   |d conf| <= 0.005 over ALL anchors, formatted detections within (0.005, 0.002)
  bf16 (BASELINE.json configs[1]) |d conf| <= 0.005 over ALL anchors, formatted detections within (0.005, 0.004)
  fp8  (BASELINE.json configs[4]) with scales calibrated on OTHER images and saved: |d conf| <= 0.006 (measured
                                  0.0048-0.0053: ON the reference's 0.005 bar, not inside it), formatted detections
                                  within (0.01, 0.008)
The originals are 2560 pixels wide, like camera-trap images (the device letterboxes them to 640): the reference rounds
every box to integer pixels of the ORIGINAL (pytorch_detector.py:1379), so a sub-pixel difference can flip a rounded
corner by 1 / 2560 = 0.0004 and a width or height by two of them; the coordinate bars are the reference's 0.001 plus
those two flips for fp16 and proportionally wider for the 8-bit types.
"""

import numpy as np
import pytest
import torch

import parity_util as PU
from oracle import pre_post as O

pytestmark = pytest.mark.gpu

SIZE = 640
ORIG = 2560                   # camera-trap sized originals, letterboxed to SIZE on the device
BARS = {                      # dtype: (max |d conf| over all anchors, detection conf bar, detection coord bar)
    'fp16': (0.005, 0.005, 0.002),
    'bf16': (0.005, 0.005, 0.004),
    'fp8': (0.006, 0.01, 0.008),       # measured 0.0048-0.0053: fp8 sits ON the reference's 0.005, it does not clear it (README)
}


@pytest.fixture(scope='module')
def x6_checkpoint(tmp_path_factory):
    import fake_yolov5 as FY
    from megadetector_amd import yolo_yaml
    model = FY.build_model(yolo_yaml.YOLOV5X6_MD, seed=7, gain=1.3)
    path = str(tmp_path_factory.mktemp('x6') / 'md_v5a.0.0.pt')          # the name the reference's tables know
    FY.save_checkpoint(model, path)
    ref_model = model.half().float()                                        # what .float() of the file holds
    imgs = PU.random_images(2, ORIG, ORIG, seed=71)
    x, infos = PU.oracle_input(imgs, SIZE, 64)
    with torch.no_grad():
        ref_pred = ref_model(x)
    del model, ref_model
    FY.uninstall()
    return path, imgs, infos, ref_pred


def _detector(path, dtype, extra=None):
    from megadetector_amd import run_detector
    opts = {'batch_size': 2, 'max_image_size': SIZE}
    if dtype is not None:
        opts['dtype'] = dtype
    opts.update(extra or {})
    det = run_detector.load_detector(path, detector_options=opts)
    det.default_image_size = SIZE
    return det


def _saved_fp8_scales(path, tmp_path):
    """calibration on OTHER images than the evaluated ones, persisted the way a user would (fp8_scales_file)"""
    import os
    f = str(tmp_path / 'scales.json')
    det = _detector(path, 'fp8', {'fp8_calibrate_on_first_batch': True, 'fp8_scales_file': f})
    calib = PU.random_images(2, ORIG, ORIG, seed=1234)
    det.generate_detections_one_batch(calib, ['c0.jpg', 'c1.jpg'], detection_threshold=0.1)
    assert os.path.isfile(f)
    det._ctx.close()
    return f


def _match(da, dst):
    """the partner compare_detection_lists would pick: same category, highest IoU >= 0.85 (md_tests.py:440-470)"""
    best, best_iou = None, -1.0
    for db in dst:
        if db['category'] != da['category']:
            continue
        iou = O.get_iou(da['bbox'], db['bbox'])
        if iou >= 0.85 and iou > best_iou:
            best, best_iou = db, iou
    return best


@pytest.mark.parametrize('dtype', ['fp16', 'bf16', 'fp8'])
def test_x6_checkpoint_meets_the_reference_bars_on_confident_detections(x6_checkpoint, dtype, tmp_path):
    path, imgs, infos, ref_pred = x6_checkpoint
    ref = ref_pred.numpy()
    score = (ref[..., 4:5] * ref[..., 5:]).max(-1)
    n_confident = int((score > 0.1).sum())
    assert n_confident > 1000, 'the fixture must have confident anchors'
    extra = {'fp8_scales_file': _saved_fp8_scales(path, tmp_path)} if dtype == 'fp8' else None
    det = _detector(path, None if dtype == 'fp16' else dtype, extra)
    assert det._ctx.dtype == dtype                                   # fp16 is what a user gets without asking
    res = det.generate_detections_one_batch(imgs, ['a.jpg', 'b.jpg'], detection_threshold=0.005)
    assert all('failure' not in r for r in res), res
    got = det._ctx.read_predictions(2, SIZE, SIZE)
    conf_bar, det_conf_bar, det_coord_bar = BARS[dtype]
    d_conf = float(np.abs(got[..., 4:] - ref[..., 4:]).max())
    d_score = float(np.abs((got[..., 4:5] * got[..., 5:]).max(-1) - score)[score > 0.005].max())
    e_box = PU.rel_err(got[..., :4], ref[..., :4])
    print('x6 checkpoint {}: {} anchors above 0.1; |d conf| {:.5f} (bar {}), |d score| on anchors > 0.005 {:.5f}, '
          'box rel {:.2e} / {:.2e}'.format(dtype, n_confident, d_conf, conf_bar, d_score, e_box[0], e_box[1]))
    assert d_conf <= conf_bar, (dtype, d_conf)
    # Detection level, same anchors.  The reference's post-processing statements (scale_coords, the integer-pixel
    # rounding of pytorch_detector.py:1379, xyxy -> normalised xywh, truncation to 4 / 3 decimals; oracle.pre_post
    # restates them, pinned by tests/golden/*) applied to the 300 most confident anchors of every image: once to the
    # file's fp32 predictions, once to the HIP predictions of the SAME anchors.  Category indices must be bit-exact,
    # confidences and coordinates within the bars (+ one unit of the 3-decimal truncation on the confidence).
    worst = [0.0, 0.0]
    n_formatted = 0
    for b in range(2):
        top = np.argsort(-score[b], kind='stable')[:300]
        cls_ref = ref[b, top, 5:].argmax(-1)
        cls_got = got[b, top, 5:].argmax(-1)
        assert np.array_equal(cls_ref, cls_got), (dtype, 'category index differs on a confident anchor')

        def rows(p, cls):
            xywh = p[top, :4]
            xyxy = np.stack([xywh[:, 0] - xywh[:, 2] / 2, xywh[:, 1] - xywh[:, 3] / 2,
                             xywh[:, 0] + xywh[:, 2] / 2, xywh[:, 1] + xywh[:, 3] / 2], 1)
            conf = p[top, 4] * p[top, 5 + cls]
            return torch.from_numpy(np.concatenate([xyxy, conf[:, None], cls[:, None].astype(np.float32)], 1).astype(np.float32))
        fa, _ = O.format_detections(rows(ref[b], cls_ref), (SIZE, SIZE), infos[b]['img_original'].shape, infos[b]['scaling_shape'], 0.0)
        fb, _ = O.format_detections(rows(got[b], cls_got), (SIZE, SIZE), infos[b]['img_original'].shape, infos[b]['scaling_shape'], 0.0)
        assert len(fa) == len(fb) == 300
        for da, db in zip(fa, fb):
            assert da['category'] == db['category']
            worst[0] = max(worst[0], abs(da['conf'] - db['conf']))
            worst[1] = max(worst[1], max(abs(u - v) for u, v in zip(da['bbox'], db['bbox'])))
            n_formatted += 1
    print('x6 checkpoint {}: {} formatted detections on the same anchors: conf {:.4f} (bar {} + 0.001) coord {:.4f} (bar {})'.format(
        dtype, n_formatted, worst[0], det_conf_bar, worst[1], det_coord_bar))
    assert worst[0] <= det_conf_bar + 0.001 + 1e-9 and worst[1] <= det_coord_bar, (dtype, worst)
    # Detection lists after each side's own greedy NMS (what compare_detection_lists would see), informational: this
    # fixture is far denser than a camera-trap image (three quarters of all anchors are confident, both lists are cut at
    # max_det = 300), so the two NMS runs pick different representatives of near-tied clusters; no bar is put on it.
    want = PU.oracle_detections(ref_pred, infos, (SIZE, SIZE), 0.005)
    unmatched = total = 0
    for r, q in zip(res, want):
        assert len(r['detections']) > 50 and len(q['detections']) > 50
        for src, dst in ((r['detections'], q['detections']), (q['detections'], r['detections'])):
            for da in src:
                total += 1
                unmatched += _match(da, dst) is None
    print('x6 checkpoint {}: after each side\'s own NMS {} of {} detections have no IoU >= 0.85 partner in the other list '
          '(dense synthetic fixture)'.format(dtype, unmatched, total))
    det._ctx.close()
