#!/usr/bin/env python3
"""
Real-weights parity harness (levels L2 / L3) -- ONE command for whoever has the checkpoint.

It cannot run in the build environment (no network: no md_v5a.0.0.pt, no md-test-package.zip); it exists so that the
first person with the files gets the reference's own verdict:

    MDV5A=/path/to/md_v5a.0.0.pt python tools/parity_real.py /path/to/md-test-images \
        [--expected /path/to/mdv5a-image-gpu-pt2.x.json] [--dtypes fp16 bf16] [--batch_size 8]

What it does, per storage type (fp16 = the detector default, bf16 = the benchmarked throughput mode):
  1. runs megadetector_amd.run_detector_batch over the folder (recursive, relative filenames, the reference's
     batch-mode defaults: detection threshold 1e-5 in the detector, output threshold 0.005) and writes
     <out_dir>/parity_<dtype>.json in MegaDetector batch format 1.6;
  2. L3 -- with --expected: compares against the reference's expected-results file with the reference's own rule,
     restated below from megadetector/utils/md_tests.py:418-531 (compare_detection_lists) and :533-640
     (compare_results): per image, every detection is matched to the same-category detection of the other file with
     the highest IoU >= 0.85; |d conf| and max |d coord| over the matches; an unmatched detection counts its own
     confidence as confidence error; both directions.  Bars: max_conf_error 0.005 and max_coord_error 0.001
     (md_tests.py:96-100), the CI's relaxed bar 0.01 (md_tests.py:1779) is reported as well;
  3. L2 -- without --expected, when both storage types ran: compares fp16 against bf16 the same way (a consistency
     figure, not a reference verdict).

Exit code 0 when every comparison that was made meets the reference's bar, 1 otherwise, 2 on usage errors.
The md5 of the checkpoint is printed next to the value the reference pins (run_detector.py:185).
"""

import argparse
import hashlib
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

MAX_CONF_ERROR = 0.005        # md_tests.py:100
MAX_COORD_ERROR = 0.001       # md_tests.py:96
CI_CONF_ERROR = 0.01          # md_tests.py:1779
IOU_MATCH = 0.85              # md_tests.py:124
KNOWN_MD5 = {'md_v5a.0.0.pt': None, 'md_v5a.0.1.pt': None, 'md_v5b.0.0.pt': None, 'md_v5b.0.1.pt': None}


def get_iou(bb1, bb2):
    """ct_utils.py:291-339 (boxes are [x_min, y_min, w, h])"""
    a = [bb1[0], bb1[1], bb1[0] + bb1[2], bb1[1] + bb1[3]]
    b = [bb2[0], bb2[1], bb2[0] + bb2[2], bb2[1] + bb2[3]]
    assert a[0] < a[2] and a[1] < a[3] and b[0] < b[2] and b[1] < b[3], 'Malformed bounding box'
    x_left, y_top = max(a[0], b[0]), max(a[1], b[1])
    x_right, y_bottom = min(a[2], b[2]), min(a[3], b[3])
    if x_right < x_left or y_bottom < y_top:
        return 0.0
    inter = (x_right - x_left) * (y_bottom - y_top)
    return inter / float((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)


def compare_detection_lists(dets_a, dets_b, bidirectional=True):
    """md_tests.py:418-531, statement for statement (matches may be many-to-one, as there)"""
    max_conf, max_coord = 0, 0
    for da in dets_a:
        match, best = None, -1
        for db in dets_b:
            if db['category'] != da['category']:
                continue
            iou = get_iou(da['bbox'], db['bbox'])
            if iou >= IOU_MATCH and iou > best:
                match, best = db, iou
        if match is None:
            if da['conf'] > max_conf:
                max_conf = da['conf']
            continue
        conf_err = abs(da['conf'] - match['conf'])
        coord_err = max(abs(da['bbox'][i] - match['bbox'][i]) for i in range(4))
        max_conf = max(max_conf, conf_err)
        max_coord = max(max_coord, coord_err)
    if bidirectional:
        r_conf, r_coord = compare_detection_lists(dets_b, dets_a, bidirectional=False)
        max_conf, max_coord = max(max_conf, r_conf), max(max_coord, r_coord)
    return max_conf, max_coord


def compare_results(actual, expected):
    """md_tests.py:533-640 on two loaded result dicts; returns (max_conf_err, file, max_coord_err, file)"""
    fa = {im['file'].replace('\\', '/'): im for im in actual['images']}
    fe = {im['file'].replace('\\', '/'): im for im in expected['images']}
    assert len(fa) == len(fe), 'expected {} files in results, found {}'.format(len(fe), len(fa))
    worst_conf, worst_conf_file, worst_coord, worst_coord_file = -1, None, -1, None
    for fn, a in fa.items():
        e = fe[fn]
        if 'failure' in a:
            assert 'failure' in e and a.get('detections') is None and e.get('detections') is None, fn
            continue
        assert 'failure' not in e, fn
        c, x = compare_detection_lists(a['detections'], e['detections'])
        if c > worst_conf:
            worst_conf, worst_conf_file = c, fn
        if x > worst_coord:
            worst_coord, worst_coord_file = x, fn
    return worst_conf, worst_conf_file, worst_coord, worst_coord_file


def md5(path):
    h = hashlib.md5()
    with open(path, 'rb') as f:
        for chunk in iter(lambda: f.read(1 << 20), b''):
            h.update(chunk)
    return h.hexdigest()


def verdict(label, res):
    c, cf, x, xf = res
    ok = c <= MAX_CONF_ERROR and x <= MAX_COORD_ERROR
    print('{}: max conf error {:.4f} ({}), max coord error {:.4f} ({})'.format(label, c, cf, x, xf))
    print('{}: reference bar (conf <= {}, coord <= {}): {}   CI bar (conf <= {}): {}'.format(
        label, MAX_CONF_ERROR, MAX_COORD_ERROR, 'PASS' if ok else 'FAIL', CI_CONF_ERROR,
        'PASS' if (c <= CI_CONF_ERROR and x <= MAX_COORD_ERROR) else 'FAIL'))
    return ok


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n\n')[0])
    ap.add_argument('image_folder')
    ap.add_argument('--model', default=os.environ.get('MDV5A'), help='checkpoint (default: $MDV5A, as run_detector.py:1083-1087)')
    ap.add_argument('--expected', default=None, help='expected-results .json of the reference (md_tests.py:155-218)')
    ap.add_argument('--dtypes', nargs='+', default=['fp16', 'bf16'], choices=['fp16', 'bf16', 'fp8'])
    ap.add_argument('--batch_size', type=int, default=8)
    ap.add_argument('--out_dir', default=None)
    ap.add_argument('--threshold', type=float, default=0.005)
    ap.add_argument('--compatibility_mode', default='classic-test', help='md_tests.py:124 uses classic-test')
    args = ap.parse_args(argv)
    if not args.model or not os.path.isfile(args.model):
        print('no checkpoint: set MDV5A=/path/to/md_v5a.0.0.pt or pass --model', file=sys.stderr)
        return 2
    if not os.path.isdir(args.image_folder):
        print('not a folder: {}'.format(args.image_folder), file=sys.stderr)
        return 2
    from megadetector_amd import run_detector_batch as RDB
    from megadetector_amd import run_detector
    print('checkpoint {}: md5 {} (the reference pins the md5 of its release assets at run_detector.py:177-216)'.format(
        args.model, md5(args.model)))
    out_dir = args.out_dir or os.path.join(os.getcwd(), 'parity_out')
    os.makedirs(out_dir, exist_ok=True)
    files = RDB._resolve_image_list(args.image_folder)
    print('{} images'.format(len(files)))
    outputs = {}
    for dt in args.dtypes:
        opts = {'dtype': dt, 'compatibility_mode': args.compatibility_mode}
        res = RDB.load_and_run_detector_batch(args.model, files, confidence_threshold=args.threshold, quiet=True,
                                              batch_size=args.batch_size, detector_options=opts)
        out = os.path.join(out_dir, 'parity_{}.json'.format(dt))
        outputs[dt] = RDB.write_results_to_file(res, out, relative_path_base=os.path.abspath(args.image_folder),
                                                detector_file=args.model)
        outputs[dt] = json.load(open(out))
    ok = True
    if args.expected:
        expected = json.load(open(args.expected))
        for dt in args.dtypes:
            ok &= verdict('L3 {} vs {}'.format(dt, os.path.basename(args.expected)), compare_results(outputs[dt], expected))
    elif len(args.dtypes) >= 2:
        a, b = args.dtypes[0], args.dtypes[1]
        ok &= verdict('L2 {} vs {}'.format(a, b), compare_results(outputs[a], outputs[b]))
    else:
        print('nothing to compare against: pass --expected or two --dtypes')
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
