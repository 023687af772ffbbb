"""
Second-source check of the oracle's test-time augmentation (oracle/yolov5.py Forward.forward_augment, what
`model(batch, augment=True)` computes at pytorch_detector.py:1313).

The arithmetic lives in ultralytics-yolov5 (models/yolo.py _forward_augment / _descale_pred / _clip_augmented,
utils/torch_utils.py scale_img), which is not available offline, so the oracle's version is "parity unpinned".  Here
the same published procedure is written a second time, in the shape the package has it -- three methods around an
nn.Module's own forward (tests/fake_yolov5.py's independent DetectionModel) -- and the oracle's functional version on
the weights read back from that module's checkpoint must give the same tensor.  That rules out a transcription error
in one of the two (scale order, which pass is flipped, the pad value, which anchors of the first / last pass are
dropped); it does not pin either against upstream.
"""

import math
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fake_yolov5 as FY  # noqa: E402
import parity_util as PU  # noqa: E402

from megadetector_amd import weights_io, yolo_yaml  # noqa: E402


class PackageStyleTTA:
    """the augmentation as the package structures it: scale_img, _descale_pred, _clip_augmented around model(x)"""

    def __init__(self, model):
        self.model = model
        self.gs = int(model.stride.max())
        self.nl = len(model.stride)

    def scale_img(self, img, ratio):
        if ratio == 1.0:
            return img
        h, w = img.shape[2:]
        s = (int(h * ratio), int(w * ratio))
        img = F.interpolate(img, size=s, mode='bilinear', align_corners=False)
        h, w = (math.ceil(v * ratio / self.gs) * self.gs for v in (h, w))
        return F.pad(img, [0, w - s[1], 0, h - s[0]], value=0.447)

    @staticmethod
    def descale(p, flips, scale, img_size):
        p = p.clone()
        p[..., :4] /= scale
        if flips == 2:
            p[..., 1] = img_size[0] - p[..., 1]
        elif flips == 3:
            p[..., 0] = img_size[1] - p[..., 0]
        return p

    def clip_augmented(self, y):
        g = sum(4 ** x for x in range(self.nl))
        e = 1
        i = (y[0].shape[1] // g) * sum(4 ** x for x in range(e))
        y[0] = y[0][:, :-i]
        i = (y[-1].shape[1] // g) * sum(4 ** (self.nl - 1 - x) for x in range(e))
        y[-1] = y[-1][:, i:]
        return y

    def __call__(self, x):
        img_size = x.shape[-2:]
        y = []
        for si, fi in zip([1, 0.83, 0.67], [None, 3, None]):
            xi = self.scale_img(x.flip(fi) if fi else x, si)
            y.append(self.descale(self.model(xi), fi, si, img_size))
        return torch.cat(self.clip_augmented(y), 1)


@pytest.mark.parametrize('yaml_name,hw', [('YOLOV5N6_TEST', (256, 384)), ('YOLOV5N_P5_TEST', (160, 224))])
def test_forward_augment_equals_the_package_style_restatement(tmp_path, yaml_name, hw):
    yaml = getattr(yolo_yaml, yaml_name)
    model = FY.build_model(yaml, seed=4, gain=1.3)
    path = str(tmp_path / 'tta.pt')
    FY.save_checkpoint(model, path)
    ref_model = model.half().float()
    x = torch.rand(2, 3, hw[0], hw[1], generator=torch.Generator().manual_seed(21))
    with torch.no_grad():
        want = PackageStyleTTA(ref_model)(x)
    FY.uninstall()
    W = weights_io.load_checkpoint(path)
    got, _ = PU.oracle_forward(W, x, emulate_bf16=False, augment=True)
    assert got.shape == want.shape
    # anchors: all of the three passes minus the dropped tails
    nl = len(ref_model.stride)
    g = sum(4 ** l for l in range(nl))
    plain, _ = PU.oracle_forward(W, x, emulate_bf16=False)
    assert got.shape[1] > 2 * plain.shape[1] * 0.6 and got.shape[1] < 3 * plain.shape[1]
    assert plain.shape[1] % g == 0
    np.testing.assert_allclose(got[..., :4].numpy(), want[..., :4].numpy(), rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(got[..., 4:].numpy(), want[..., 4:].numpy(), rtol=0, atol=5e-4)
