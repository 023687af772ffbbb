"""
Graph replay of mdhip_forward (include/mdhip.h: mdhip_set_graph): launch plumbing for small batches -- the same kernels
with the same arguments, so the predictions must be the same BITS as the eager forward's, on every replay, for both
prediction buffers, after calls that change what a forward launches, and through the detector (whose default is 'auto').
"""

import numpy as np
import pytest

import parity_util as PU

pytestmark = pytest.mark.gpu


def _identity_geoms(imgs):
    return [(im.shape[0], im.shape[1], im.shape[0], im.shape[1], 0, 0) for im in imgs]


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
def test_graph_replay_is_bit_identical_to_the_eager_forward(dtype):
    from megadetector_amd import weights_io, yolo_yaml
    from megadetector_amd.hip_backend import HipContext
    W = weights_io.synthetic_weights(yolo_yaml.YOLOV5X6_MD, seed=0)
    n, hh, ww = 2, 384, 640
    ctx = HipContext(W, device=0, dtype=dtype, max_batch=4, max_h=hh, max_w=ww)
    try:
        batches = [PU.random_images(n, hh, ww, seed=50 + i) for i in range(5)]

        def run(imgs, k=n):
            ctx.preprocess(imgs[:k], _identity_geoms(imgs[:k]), hh, ww)
            ctx.forward(k, hh, ww)
            return ctx.read_predictions(k).copy()
        ctx.set_graph('off')
        want = [run(b) for b in batches]
        want1 = run(batches[0], 1)
        ctx.set_graph('on')
        for rep in range(2):                      # eager, capture (both prediction buffers), then replays
            got = [run(b) for b in batches]
            for a, b in zip(got, want):
                np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(run(batches[0], 1), want1)          # another batch size: its own graph
        np.testing.assert_array_equal(run(batches[0], 1), want1)
        np.testing.assert_array_equal(run(batches[0], 1), want1)
        np.testing.assert_array_equal(run(batches[1]), want[1])
        # a call that changes what a forward launches drops the graphs: fused bottlenecks off = other launches, same bits
        ctx.set_fuse(False)
        for b, w in zip(batches[:3], want[:3]):
            np.testing.assert_array_equal(run(b), w)
        ctx.set_fuse(True)
        # 'auto' with a bound below this batch size runs eagerly
        ctx.set_graph('auto', 1)
        np.testing.assert_array_equal(run(batches[2]), want[2])
        np.testing.assert_array_equal(run(batches[2]), want[2])
        # test-time augmentation is not replayed and still right after replays
        ctx.set_graph('on')
        ctx.preprocess(batches[3], _identity_geoms(batches[3]), hh, ww)
        ctx.forward_tta(n, hh, ww)
        tta_on = ctx.read_predictions(n).copy()
        ctx.set_graph('off')
        ctx.preprocess(batches[3], _identity_geoms(batches[3]), hh, ww)
        ctx.forward_tta(n, hh, ww)
        np.testing.assert_array_equal(ctx.read_predictions(n), tta_on)
    finally:
        ctx.close()


def test_detector_results_do_not_depend_on_graph_replay():
    from megadetector_amd import run_detector
    imgs = PU.random_images(3, 480, 640, seed=9)
    ids = ['a.jpg', 'b.jpg', 'c.jpg']
    out = {}
    for mode in ('off', 'auto', 'on'):
        det = run_detector.load_detector('synthetic', detector_options={'batch_size': 3, 'max_image_size': 640, 'hip_graph': mode})
        det.default_image_size = 640
        res = [det.generate_detections_one_batch(imgs, ids, detection_threshold=0.001) for _ in range(3)]
        assert res[0] == res[1] == res[2]
        out[mode] = res[0]
        det._ctx.close()
    assert out['off'] == out['auto'] == out['on']
    assert sum(len(r['detections']) for r in out['off']) > 0
