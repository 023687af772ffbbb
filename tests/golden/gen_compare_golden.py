"""
Generates tests/golden/compare_kat.json by calling the *real* reference comparison rule
(/root/reference/megadetector/utils/md_tests.py:418-531 compare_detection_lists, with MDTestOptions' defaults
:96-124) on seeded pairs of detection lists.  The reference cannot travel to the GPU box, so its outputs are
committed as a small fixture together with this script.  jsonpickle is stubbed (unused on this path).

Run (build container only):  python tests/golden/gen_compare_golden.py
"""

import json
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, '/root/reference')
for name in ('jsonpickle', 'cv2', 'humanfriendly'):
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)

from megadetector.utils import md_tests  # noqa: E402


def random_pair(rng, n, jitter, drop, conf_noise):
    a = []
    for _ in range(n):
        x, y = rng.uniform(0, 0.7, 2)
        w, h = rng.uniform(0.05, 0.3, 2)
        a.append({'category': str(int(rng.integers(1, 4))), 'conf': round(float(rng.uniform(0.01, 0.99)), 3),
                  'bbox': [round(float(v), 4) for v in (x, y, w, h)]})
    b = []
    for d in a:
        if rng.uniform() < drop:
            continue
        bb = [round(float(v + rng.normal(0, jitter)), 4) for v in d['bbox']]
        bb[2], bb[3] = max(bb[2], 0.01), max(bb[3], 0.01)
        cat = d['category'] if rng.uniform() > 0.05 else str(int(rng.integers(1, 4)))
        b.append({'category': cat, 'conf': round(float(np.clip(d['conf'] + rng.normal(0, conf_noise), 0.001, 0.999)), 3),
                  'bbox': bb})
    if rng.uniform() < 0.3 and a:                       # a near-duplicate: exercises the many-to-one matching
        d = dict(a[0])
        d['conf'] = round(d['conf'] * 0.5, 3)
        b.append(d)
    return a, b


def main():
    rng = np.random.default_rng(20240926)
    options = md_tests.MDTestOptions()
    cases = []
    for i in range(40):
        n = int(rng.integers(0, 12))
        a, b = random_pair(rng, n, jitter=[0.0, 0.0005, 0.003, 0.02][i % 4], drop=[0.0, 0.1, 0.3][i % 3],
                           conf_noise=[0.0, 0.002, 0.02][i % 3])
        r = md_tests.compare_detection_lists(a, b, options, bidirectional_comparison=True)
        cases.append({'a': a, 'b': b, 'max_conf_error': r['max_conf_error'], 'max_coord_error': r['max_coord_error']})
    out = {'source': 'megadetector/utils/md_tests.py:418-531 (real function), options: iou {} conf {} coord {}'.format(
        options.iou_threshold_for_file_comparison, options.max_conf_error, options.max_coord_error),
        'iou_threshold': options.iou_threshold_for_file_comparison,
        'max_conf_error': options.max_conf_error, 'max_coord_error': options.max_coord_error, 'cases': cases}
    path = os.path.join(REPO, 'tests', 'golden', 'compare_kat.json')
    with open(path, 'w') as f:
        json.dump(out, f, indent=0)
    print('wrote', path, len(cases), 'cases;', sum(1 for c in cases if c['max_conf_error'] > 0), 'with conf error')


if __name__ == '__main__':
    main()
