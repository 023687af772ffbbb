"""
Pins the oracle (oracle/) against every golden vector / known-answer test the reference
tree holds for this path, and against fixtures generated from the reference's own code
(tests/golden/gen_golden_from_reference.py).
"""

import json
import os

import numpy as np
import pytest
import torch

from oracle import pre_post as O
from oracle import yolov5 as Y
from megadetector_amd import yolo_yaml
from conftest import GOLDEN


# ---- reference megadetector/utils/ct_utils.py:1332-1346 ------------------------------
def test_ct_utils_known_answers_literal():
    assert O.truncate_float_array([0.12345, 0.67890], precision=3) == [0.123, 0.678]
    assert O.truncate_float_array([1.0, 2.0], precision=2) == [1.0, 2.0]
    assert O.truncate_float(0.12345, precision=3) == 0.123
    assert O.truncate_float(1.999, precision=2) == 1.99
    assert O.truncate_float(0.0003214884, precision=6) == 0.000321
    assert O.truncate_float(1.0003214884, precision=6) == 1.000321
    assert O.round_float_array([0.12345, 0.67890], precision=3) == [0.123, 0.679]
    assert O.round_float(0.12345, precision=3) == 0.123
    assert O.round_float(0.12378, precision=3) == 0.124
    assert O.round_float(1.999, precision=2) == 2.00


# ---- reference ct_utils.py:1467-1493 ---------------------------------------------------
def test_bbox_known_answers_literal():
    assert np.allclose(O.convert_yolo_to_xywh([0.5, 0.5, 0.2, 0.2]), [0.4, 0.4, 0.2, 0.2])
    assert abs(O.get_iou([0, 0, 0.5, 0.5], [0.25, 0.25, 0.5, 0.5]) - 0.142857) < 1e-5
    assert abs(O.get_iou([0, 0, 1, 1], [0.5, 0.5, 1, 1]) - (0.25 / 1.75)) < 1e-5
    assert O.get_iou([0, 0, 1, 1], [1, 1, 1, 1]) == 0.0


def test_ct_utils_fixture_from_reference():
    kat = json.load(open(os.path.join(GOLDEN, 'ct_utils_kat.json')))
    xs = kat['x']
    for prec in (2, 3, 4, 6):
        assert [O.truncate_float(x, prec) for x in xs] == kat['truncate_{}'.format(prec)]
        assert [O.round_float(x, prec) for x in xs] == kat['round_{}'.format(prec)]
    assert [O.convert_yolo_to_xywh(b) for b in kat['yolo_boxes']] == kat['yolo_to_xywh']
    assert [O.get_iou(a, b) for a, b in zip(kat['iou_a'], kat['iou_b'])] == kat['iou']


# ---- reference megadetector/tests/test_nms_synthetic.py --------------------------------
def _centres(det):
    return [((float(d[0] + d[2]) / 2), (float(d[1] + d[3]) / 2), float(d[4]), int(d[5])) for d in det]


def test_nms_synthetic_expectations():
    blob = np.load(os.path.join(GOLDEN, 'nms_reference.npz'))
    pred = torch.from_numpy(blob['synthetic/pred'])
    det = O.nms(pred, conf_thres=0.3, iou_thres=0.5, max_det=300)[0]
    assert det.shape[0] != 0
    c = _centres(det)
    s1 = [x for x in c if 80 <= x[0] <= 130 and 80 <= x[1] <= 130 and x[3] == 0]
    s1b = [x for x in c if 180 <= x[0] <= 220 and 80 <= x[1] <= 120 and x[3] == 0]
    assert len(s1) == 1 and s1[0][2] >= 0.7            # test_nms_synthetic.py:188-208
    assert len(s1b) == 1 and s1b[0][2] >= 0.8
    s2 = [x for x in c if 270 <= x[0] <= 410 and 70 <= x[1] <= 130 and x[3] == 0]
    assert len(s2) == 2                                  # :247
    s3 = [x for x in c if 65 <= x[0] <= 135 and 265 <= x[1] <= 335]
    assert len(s3) == 2 and len(set(x[3] for x in s3)) == 2   # :263
    s4 = [x for x in c if 460 <= x[0] <= 560 and 260 <= x[1] <= 360 and x[3] == 0]
    # :270-303: either one box survives, or the survivors overlap by less than the threshold
    assert abs(max(x[2] for x in s4) - 0.95 * 0.9) < 1e-6
    rows = [d for d in det if 460 <= float(d[0] + d[2]) / 2 <= 560 and 260 <= float(d[1] + d[3]) / 2 <= 360]
    for i in range(len(rows)):
        for j in range(i + 1, len(rows)):
            a, b = rows[i][:4].tolist(), rows[j][:4].tolist()
            iou = O.get_iou([a[0], a[1], a[2] - a[0], a[3] - a[1]], [b[0], b[1], b[2] - b[0], b[3] - b[1]])
            assert iou < 0.5


def test_nms_identical_boxes():
    blob = np.load(os.path.join(GOLDEN, 'nms_reference.npz'))
    det = O.nms(torch.from_numpy(blob['identical/pred']), 0.3, 0.5, 300)[0]
    assert det.shape[0] == 1 and abs(float(det[0, 4]) - 0.81) < 0.01   # :322-331


def _canon(a):
    """sort rows by (conf desc, then all columns) so unspecified tie order does not matter"""
    a = np.asarray(a, dtype=np.float32)
    if a.shape[0] == 0:
        return a
    keys = (a[:, 5], a[:, 3], a[:, 2], a[:, 1], a[:, 0], -a[:, 4])
    return a[np.lexsort(keys)]


@pytest.mark.parametrize('case', ['synthetic', 'identical', 'rand_a', 'rand_b', 'rand_c', 'empty'])
def test_nms_matches_reference_fixture(case):
    blob = np.load(os.path.join(GOLDEN, 'nms_reference.npz'))
    pred = torch.from_numpy(blob[case + '/pred'])
    ct, it, md = blob[case + '/params']
    res = O.nms(pred, conf_thres=float(ct), iou_thres=float(it), max_det=int(md))
    for i, r in enumerate(res):
        ref = blob['{}/out{}'.format(case, i)]
        assert r.shape == ref.shape
        np.testing.assert_array_equal(_canon(r.numpy()), _canon(ref))


# ---- topology pinned by the figures the reference cites --------------------------------
def test_topology_reproduces_published_flops_and_params():
    # reference docs/release-notes/mdv1000-release.md:279: YOLOv5x6, 209.8 GFLOPs; 140.7 M params
    gmac, n_convs, params = Y.count_work(yolo_yaml.YOLOV5X6_COCO, 640, 640)
    assert abs(2 * gmac - 209.8) / 209.8 < 0.005
    assert abs(params / 1e6 - 140.7) / 140.7 < 0.005
    gmac, n_convs, _ = Y.count_work(yolo_yaml.YOLOV5X6_MD, 1280, 1280)
    assert n_convs == 163
    assert abs(gmac - 415.82) < 0.01


def test_letterbox_geometry_common_shapes():
    # 4:3 -> 960x1280, 16:9 -> 768x1280 (720 + 48 pad)
    g = O.letterbox_geometry((1536, 2048))
    assert g['out_hw'] == (960, 1280) and g['new_unpad'] == (1280, 960)
    g = O.letterbox_geometry((1080, 1920))
    assert g['out_hw'] == (768, 1280) and g['top'] == 24 and g['bottom'] == 24
    g = O.letterbox_geometry((1280, 1280))
    assert g['out_hw'] == (1280, 1280) and g['ratio'] == (1.0, 1.0) and g['pad'] == (0.0, 0.0)


def test_resize_linear_identity_and_constant():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(O.resize_linear_u8(img, 53, 37), img)
    const = np.full((40, 60, 3), 77, dtype=np.uint8)
    assert np.all(O.resize_linear_u8(const, 91, 33) == 77)
    # exact 2x upscale of a 2-pixel ramp: interior samples at 1/4, 3/4
    ramp = np.array([[[0, 0, 0], [200, 200, 200]]], dtype=np.uint8)
    up = O.resize_linear_u8(ramp, 4, 1)
    assert up[0, :, 0].tolist() == [0, 50, 150, 200]


def test_detection_list_comparison_rule_matches_the_reference():
    """md_tests.py:418-531 compare_detection_lists: outputs of the REAL function (tests/golden/gen_compare_golden.py)
    against the two restatements -- the oracle's (used by the parity tests) and tools/parity_real.py's (the
    real-weights harness)"""
    import importlib.util
    kat = json.load(open(os.path.join(GOLDEN, 'compare_kat.json')))
    assert kat['iou_threshold'] == 0.85 and kat['max_conf_error'] == 0.005 and kat['max_coord_error'] == 0.001
    spec = importlib.util.spec_from_file_location('parity_real', os.path.join(os.path.dirname(GOLDEN), '..', 'tools', 'parity_real.py'))
    PR = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(PR)
    assert (PR.IOU_MATCH, PR.MAX_CONF_ERROR, PR.MAX_COORD_ERROR) == (0.85, 0.005, 0.001)
    n_nonzero = 0
    for c in kat['cases']:
        want = (c['max_conf_error'], c['max_coord_error'])
        assert O.compare_detection_lists(c['a'], c['b']) == want
        assert PR.compare_detection_lists(c['a'], c['b']) == want
        n_nonzero += want[0] > 0
    assert n_nonzero >= 20
    # compare_results (md_tests.py:533-640) on whole files: failures must agree, worst image reported
    a = {'images': [{'file': 'x/1.jpg', 'detections': kat['cases'][1]['a']}, {'file': 'x\\\\2.jpg', 'failure': 'f', 'detections': None}]}
    b = {'images': [{'file': 'x/1.jpg', 'detections': kat['cases'][1]['b']}, {'file': 'x/2.jpg', 'failure': 'f'}]}
    b['images'][1]['file'] = 'x/2.jpg'
    a['images'][1]['file'] = 'x\\2.jpg'
    c, cf, x, xf = PR.compare_results(a, b)
    assert (c, x) == (kat['cases'][1]['max_conf_error'], kat['cases'][1]['max_coord_error']) and cf == 'x/1.jpg'
