"""
End-to-end rate of the batch driver on real files: JPEG decode (PIL) + EXIF
handling + device letterbox + forward + NMS + formatting through load_and_run_detector_batch, for several
loader configurations.  The JPEGs are synthesised here (camera-trap-like 4:3 frames, photo-like content so
that the decode cost is realistic) -- there are no image files offline.

Usage (GPU box): python tools/e2e_feed_bench.py [--n 512] [--shape 1536x2048] [--batch 32]
                                               [--workers 8,32,64] [--out gpurun_out/e2e_feed.json]
"""

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def make_jpegs(folder, n_unique, n_total, h, w, seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    files = []
    for i in range(n_unique):
        img = np.zeros((h, w, 3), np.float32)
        for c in range(3):                                  # smooth background + blobs + sensor noise
            img[..., c] = 90 + 60 * np.sin(xx / rng.uniform(80, 400) + rng.uniform(0, 6)) * np.cos(yy / rng.uniform(80, 400))
        for _ in range(12):
            cy, cx, s = rng.uniform(0, h), rng.uniform(0, w), rng.uniform(20, 200)
            img += np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))[..., None] * rng.uniform(-90, 90, 3)
        img += rng.normal(0, 6, img.shape)
        p = os.path.join(folder, 'u{:04d}.jpg'.format(i))
        Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(p, quality=90)
        files.append(p)
    out = []
    for i in range(n_total):
        p = os.path.join(folder, 'img{:05d}.jpg'.format(i))
        os.symlink(files[i % n_unique], p)
        out.append(p)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=2048)
    ap.add_argument('--unique', type=int, default=32)
    ap.add_argument('--shape', default='1536x2048')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--workers', default='8,32,64')
    ap.add_argument('--model', default='synthetic:YOLOV5X6_MD:0')
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    h, w = (int(v) for v in args.shape.lower().split('x'))
    from megadetector_amd import run_detector, run_detector_batch as RDB
    tmp = tempfile.mkdtemp(prefix='mdhip_e2e_')
    t0 = time.time()
    files = make_jpegs(tmp, args.unique, args.n, h, w)
    print('{} JPEGs ({} unique, {}x{}) written in {:.1f} s; {:.2f} MB each'.format(
        len(files), args.unique, h, w, time.time() - t0, os.path.getsize(files[0]) / 1e6))
    # decode cost of one image on one core
    t0 = time.time()
    for f in files[:8]:
        RDB.load_image(f)
    dec_ms = (time.time() - t0) / 8 * 1e3
    print('PIL decode + EXIF: {:.1f} ms per image on one core'.format(dec_ms))
    det = run_detector.load_detector(args.model, detector_options={'batch_size': args.batch})
    rows = []

    def run(label, n=None, **kw):
        use = files[:n] if n else files
        RDB.load_and_run_detector_batch(args.model, files[:args.batch * 2], detector=det, batch_size=args.batch, quiet=True, **kw)
        # steady-state rate: from the first batch of results to the last (loader start-up excluded), and overall
        stamps = []

        class Timed(list):
            def extend(self, new):
                stamps.append((time.time(), len(new)))
                super().extend(new)
        t = time.time()
        res = RDB.load_and_run_detector_batch(args.model, use, detector=det, batch_size=args.batch, quiet=True,
                                              results=Timed(), **kw)
        el = time.time() - t
        ok = sum(1 for r in res if 'failure' not in r)
        steady = float('nan')
        if len(stamps) > 2:
            steady = sum(c for _, c in stamps[1:]) / max(1e-9, stamps[-1][0] - stamps[0][0])
        rows.append({'config': label, 'images': len(res), 'ok': ok, 'seconds': el, 'images_per_s': len(res) / el,
                     'steady_images_per_s': steady})
        print('{:52s} {:6.1f} images/s overall, {:6.1f} steady  ({} images, {} ok, {:.1f} s)'.format(
            label, len(res) / el, steady, len(res), ok, el))

    run('in-line decode (no queue)', n=256)
    for nw in [int(v) for v in args.workers.split(',')]:
        run('image queue, {} loader threads'.format(nw), use_image_queue=True, loader_workers=nw)
        run('image queue, {} loader threads, preprocess on queue'.format(nw), use_image_queue=True, loader_workers=nw,
            preprocess_on_image_queue=True)
        run('shared-memory ring, {} loader processes'.format(nw), use_image_queue=True, use_threads_for_queue=False,
            loader_workers=nw)
    if args.out:
        with open(args.out, 'w') as f:
            json.dump({'n': args.n, 'shape': [h, w], 'batch': args.batch, 'decode_ms_one_core': dec_ms, 'rows': rows}, f, indent=1)


if __name__ == '__main__':
    main()
