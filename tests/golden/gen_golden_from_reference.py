"""
Generates tests/golden/*.npz|json by importing the *real* reference code from
/root/reference (read-only) in the build container.  The reference cannot travel to the
GPU box, so its outputs are committed as small fixtures together with this script.

What is exercised from the reference itself:
  * megadetector/utils/ct_utils.py: truncate_float(_array), round_float(_array),
    convert_yolo_to_xywh, get_iou  (pure Python; jsonpickle is stubbed, it is unused here)
  * megadetector/detection/pytorch_detector.py:nms()  -- the in-tree part (thresholding,
    obj*cls, argmax class, per-class loop, concat, sort, max_det).  Its inner call
    torchvision.ops.nms is a third-party kernel absent from this container; it is bound
    to oracle.pre_post._greedy_nms (the restatement of torchvision's published greedy
    rule), so these fixtures pin the reference's *own* statements around it, not
    torchvision.  cv2 / humanfriendly / jsonpickle are stubbed as empty modules: nothing
    on this path touches them.

Run:  python tests/golden/gen_golden_from_reference.py
"""

import json
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, '/root/reference')

from oracle import pre_post as O  # noqa: E402
sys.path.insert(0, os.path.join(REPO, 'tests'))
from parity_util import random_predictions  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    for name in ('cv2', 'jsonpickle', 'humanfriendly'):
        if name not in sys.modules:
            _stub(name)
    tv = _stub('torchvision')
    tv.ops = _stub('torchvision.ops', nms=lambda b, s, t: O._greedy_nms(b.cpu(), s.cpu(), t))
    # tqdm / requests exist in this image; PIL exists.
    import megadetector.utils.ct_utils as ct
    import megadetector.detection.pytorch_detector as ptd
    return ct, ptd


def synthetic_predictions():
    """Inputs restated from reference megadetector/tests/test_nms_synthetic.py:83-131."""
    boxes = [
        [100, 100, 80, 80, 0.9, 0.8, 0.1, 0.1], [105, 105, 80, 80, 0.9, 0.5, 0.1, 0.1],
        [200, 100, 60, 60, 0.9, 0.9, 0.05, 0.05], [202, 102, 60, 60, 0.9, 0.7, 0.1, 0.1],
        [300, 100, 60, 60, 0.9, 0.7, 0.1, 0.1], [380, 100, 60, 60, 0.9, 0.6, 0.1, 0.1],
        [100, 300, 70, 70, 0.9, 0.7, 0.1, 0.1], [100, 300, 70, 70, 0.9, 0.1, 0.7, 0.1],
        [500, 300, 80, 80, 0.95, 0.9, 0.05, 0.05], [510, 310, 80, 80, 0.9, 0.7, 0.1, 0.1],
        [520, 320, 80, 80, 0.85, 0.6, 0.15, 0.15], [200, 500, 50, 50, 0.1, 0.05, 0.02, 0.03],
    ]
    p = torch.zeros(1, 20, 8)
    for i, b in enumerate(boxes):
        p[0, i, :] = torch.tensor(b)
    return p


def identical_predictions():
    """reference test_nms_synthetic.py:309-316"""
    p = torch.zeros(1, 5, 8)
    p[0, 0, :] = torch.tensor([100, 100, 50, 50, 0.9, 0.9, 0.05, 0.05])
    p[0, 1, :] = torch.tensor([100, 100, 50, 50, 0.9, 0.7, 0.1, 0.1])
    return p


def main():
    ct, ptd = import_reference()
    out_dir = os.path.join(REPO, 'tests', 'golden')
    os.makedirs(out_dir, exist_ok=True)

    # ---- ct_utils known answers on seeded inputs ----
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.random(200), rng.random(50) * 3.0, [0.12345, 1.999, 0.0003214884,
                        1.0003214884, 0.12378, 0.0, 1.0, 0.9999999, 0.0049999, 0.005]])
    kat = {'x': xs.tolist()}
    for prec in (2, 3, 4, 6):
        kat['truncate_{}'.format(prec)] = [ct.truncate_float(float(x), precision=prec) for x in xs]
        kat['round_{}'.format(prec)] = [ct.round_float(float(x), precision=prec) for x in xs]
    yolo = rng.random((64, 4))
    kat['yolo_boxes'] = yolo.tolist()
    kat['yolo_to_xywh'] = [ct.convert_yolo_to_xywh(list(map(float, b))) for b in yolo]
    a = np.concatenate([rng.random((64, 2)) * 0.5, 0.05 + rng.random((64, 2)) * 0.5], 1)
    b = np.concatenate([rng.random((64, 2)) * 0.5, 0.05 + rng.random((64, 2)) * 0.5], 1)
    kat['iou_a'] = a.tolist()
    kat['iou_b'] = b.tolist()
    kat['iou'] = [ct.get_iou(list(map(float, p)), list(map(float, q))) for p, q in zip(a, b)]
    with open(os.path.join(out_dir, 'ct_utils_kat.json'), 'w') as f:
        json.dump(kat, f)

    # ---- reference nms() on synthetic + random predictions ----
    cases = {
        'synthetic': (synthetic_predictions(), 0.3, 0.5, 300),
        'identical': (identical_predictions(), 0.3, 0.5, 300),
        'rand_a': (random_predictions(1, 2, 600), 0.05, 0.45, 300),
        'rand_b': (random_predictions(2, 3, 2000), 1e-5, 0.45, 300),
        'rand_c': (random_predictions(3, 1, 1500), 0.2, 0.6, 50),
        'empty': (torch.zeros(2, 64, 8), 0.1, 0.45, 300),
    }
    blob = {}
    for name, (pred, ct_, it_, md_) in cases.items():
        res = ptd.nms(pred.clone(), conf_thres=ct_, iou_thres=it_, max_det=md_)
        blob[name + '/pred'] = pred.numpy()
        blob[name + '/params'] = np.array([ct_, it_, md_], dtype=np.float64)
        for i, r in enumerate(res):
            blob['{}/out{}'.format(name, i)] = r.numpy().astype(np.float32)
        print(name, [tuple(r.shape) for r in res])
    np.savez_compressed(os.path.join(out_dir, 'nms_reference.npz'), **blob)
    print('wrote', out_dir)


if __name__ == '__main__':
    main()
