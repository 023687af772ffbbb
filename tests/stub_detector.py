"""
Deterministic stand-in for the detector behind the host path (the 3-method duck type the reference's loop calls,
reference tf_detector.py:136 shows the minimal one), and the seeded inputs shared by
tests/golden/gen_host_golden_from_reference.py (which drives the REAL reference loop / writer with them) and
tests/test_host_path_reference.py (which drives megadetector_amd with them and compares byte for byte).
"""

import os

import numpy as np


class StubDetector:
    """Deterministic fake of the 3-method detector interface (reference tf_detector.py:136 shows the
    minimal duck type); detections are a pure function of the pixels, so any orchestration must
    give identical output."""

    default_image_size = 1280
    letterbox_stride = 64

    def __init__(self, fail_on=None):
        self.fail_on = fail_on or set()
        self.batches = []

    def _one(self, img, name):
        if isinstance(img, dict):
            img = img['img_original']
        a = np.asarray(img)
        if a.ndim != 3:
            return {'file': name, 'detections': None, 'failure': 'image access failure'}
        s = int(a.astype(np.int64).sum())
        dets = []
        for k in range(s % 4 + 1):
            conf = ((s >> (3 * k)) % 1000) / 1000.0
            dets.append({'category': str(1 + (s + k) % 3), 'conf': conf,
                         'bbox': [0.1 * k, 0.05, 0.2, 0.3]})
        return {'file': name, 'detections': dets, 'max_detection_conf': max(d['conf'] for d in dets)}

    def generate_detections_one_batch(self, imgs, names, detection_threshold=1e-5, image_size=None,
                                      augment=False, verbose=False):
        self.batches.append(len(imgs))
        if any(n in self.fail_on for n in names):
            raise RuntimeError('simulated device failure')
        return [self._one(i, n) for i, n in zip(imgs, names)]

    def generate_detections_one_image(self, img, name='unknown', detection_threshold=1e-5, image_size=None,
                                      augment=False, verbose=False):
        r = self._one(img, name)
        if r.get('detections') is not None:
            r['detections'] = [d for d in r['detections'] if d['conf'] >= detection_threshold]
        return r


# (file name, height, width, seed): lossless PNGs, so the stub's detections are a function of the spec
IMAGE_SPECS = [['img_{:02d}.png'.format(i), 24 + i, 40 - i, 100 + i] for i in range(13)]


def write_test_images(folder, specs=None):
    from PIL import Image
    names = []
    for name, h, w, seed in (specs or IMAGE_SPECS):
        rng = np.random.default_rng(seed)
        p = os.path.join(folder, name)
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(p)
        names.append(p)
    return names


def sample_results():
    """Per-image result dicts as the loop produces them (absolute and Windows-style paths, unsorted files and
    detections, a failure, an image without detections, float noise in conf / bbox)."""
    return [
        {'file': '/data/cam/b/img_2.jpg', 'max_detection_conf': 0.912,
         'detections': [{'category': '1', 'conf': 0.131, 'bbox': [0.1, 0.2, 0.3, 0.4]},
                        {'category': '2', 'conf': 0.912, 'bbox': [0.5001, 0.25, 0.125, 0.3333]},
                        {'category': '3', 'conf': 0.5, 'bbox': [0.0, 0.0, 1.0, 1.0]}]},
        {'file': '/data/cam/a/img_1.jpg', 'max_detection_conf': 0.0, 'detections': []},
        {'file': '/data/cam/a\\sub\\img_0.jpg', 'max_detection_conf': 0.007,
         'detections': [{'category': '1', 'conf': 0.007, 'bbox': [0.9999, 0.0001, 0.0001, 0.9998]}]},
        {'file': '/data/cam/c/broken.jpg', 'failure': 'image access failure'},
        {'file': '/data/cam/c/oom.jpg', 'failure': 'inference failure', 'detections': None},
        {'file': '/data/cam/a/img_9.jpg', 'max_detection_conf': 0.3, 'width': 1920, 'height': 1080,
         'datetime': '2023:01:02 03:04:05',
         'detections': [{'category': '2', 'conf': 0.3, 'bbox': [0.25, 0.25, 0.5, 0.5]},
                        {'category': '2', 'conf': 0.3, 'bbox': [0.26, 0.25, 0.5, 0.5]}]},
    ]
