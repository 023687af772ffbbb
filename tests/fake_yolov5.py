"""
A small, independent torch.nn implementation of the YOLOv5 v6 architecture (Conv / Bottleneck / C3 /
SPPF / Concat / Detect / DetectionModel), registered under the module names the real package uses
(`models.common`, `models.yolo`), so that `torch.save({'model': model.half(), ...})` produces a file
with the same pickle layout as md_v5a.0.0.pt: whole-module pickles that name classes of a package
that is NOT importable when the file is read (reference pytorch_detector.py:929-957 needs
`models.*` importable; megadetector_amd.weights_io.load_checkpoint must not).

Test infrastructure only.  Written from the published architecture description (SURVEY.md section 8(a)
P4), not from the yolov5 sources.
"""

import math
import sys
import types

import torch
import torch.nn as nn


def _install():
    """Creates (or returns) the fake `models`, `models.common`, `models.yolo` modules."""
    if 'models.yolo' in sys.modules and getattr(sys.modules['models.yolo'], '_mdhip_fake', False):
        return sys.modules['models.common'], sys.modules['models.yolo']
    pkg = types.ModuleType('models')
    pkg.__path__ = []
    common = types.ModuleType('models.common')
    yolo = types.ModuleType('models.yolo')
    yolo._mdhip_fake = True

    class Conv(nn.Module):
        def __init__(self, c1, c2, k=1, s=1, p=None):
            super().__init__()
            self.conv = nn.Conv2d(c1, c2, k, s, k // 2 if p is None else p, bias=False)
            self.bn = nn.BatchNorm2d(c2, eps=1e-3, momentum=0.03)
            self.act = nn.SiLU()

        def forward(self, x):
            return self.act(self.bn(self.conv(x)))

    class Bottleneck(nn.Module):
        def __init__(self, c1, c2, shortcut=True):
            super().__init__()
            self.cv1 = Conv(c1, c2, 1, 1)
            self.cv2 = Conv(c2, c2, 3, 1)
            self.add = shortcut and c1 == c2

        def forward(self, x):
            y = self.cv2(self.cv1(x))
            return x + y if self.add else y

    class C3(nn.Module):
        def __init__(self, c1, c2, n=1, shortcut=True):
            super().__init__()
            c_ = int(c2 * 0.5)
            self.cv1 = Conv(c1, c_, 1, 1)
            self.cv2 = Conv(c1, c_, 1, 1)
            self.cv3 = Conv(2 * c_, c2, 1, 1)
            self.m = nn.Sequential(*(Bottleneck(c_, c_, shortcut) for _ in range(n)))

        def forward(self, x):
            return self.cv3(torch.cat((self.m(self.cv1(x)), self.cv2(x)), 1))

    class SPPF(nn.Module):
        def __init__(self, c1, c2, k=5):
            super().__init__()
            c_ = c1 // 2
            self.cv1 = Conv(c1, c_, 1, 1)
            self.cv2 = Conv(c_ * 4, c2, 1, 1)
            self.m = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)

        def forward(self, x):
            x = self.cv1(x)
            y1 = self.m(x)
            y2 = self.m(y1)
            return self.cv2(torch.cat((x, y1, y2, self.m(y2)), 1))

    class Concat(nn.Module):
        def __init__(self, dimension=1):
            super().__init__()
            self.d = dimension

        def forward(self, xs):
            return torch.cat(xs, self.d)

    class Detect(nn.Module):
        def __init__(self, nc, anchors, ch):
            super().__init__()
            self.nc, self.no = nc, nc + 5
            self.nl, self.na = len(anchors), len(anchors[0]) // 2
            self.register_buffer('anchors', torch.tensor(anchors).float().view(self.nl, -1, 2))
            self.m = nn.ModuleList(nn.Conv2d(c, self.no * self.na, 1) for c in ch)
            self.stride = None
            self.inplace = True

        def forward(self, xs):
            z = []
            for i, x in enumerate(xs):
                x = self.m[i](x)
                bs, _, ny, nx = x.shape
                y = x.view(bs, self.na, self.no, ny, nx).permute(0, 1, 3, 4, 2).sigmoid()
                gy, gx = torch.meshgrid(torch.arange(ny), torch.arange(nx), indexing='ij')
                grid = torch.stack((gx, gy), 2).view(1, 1, ny, nx, 2).to(y.dtype)
                anchor_grid = (self.anchors[i] * self.stride[i]).view(1, self.na, 1, 1, 2)
                xy = (y[..., 0:2] * 2 - 0.5 + grid) * self.stride[i]
                wh = (y[..., 2:4] * 2) ** 2 * anchor_grid
                z.append(torch.cat((xy, wh, y[..., 4:]), -1).view(bs, -1, self.no))
            return torch.cat(z, 1)

    class DetectionModel(nn.Module):
        def __init__(self, yaml):
            super().__init__()
            self.yaml = dict(yaml)
            gd, gw = yaml['depth_multiple'], yaml['width_multiple']
            nc, anchors = yaml['nc'], yaml['anchors']
            no = (len(anchors[0]) // 2) * (nc + 5)
            ch, layers, strides_div = [3], [], []
            for i, (f, n, m, args) in enumerate(list(yaml['backbone']) + list(yaml['head'])):
                n = max(round(n * gd), 1) if n > 1 else n
                div = lambda c: c if c == no else int(math.ceil(c * gw / 8) * 8)
                src = f if isinstance(f, int) else f[0]
                d_in = 1 if (i == 0) else strides_div[src if src >= 0 else i + src]
                if m == 'Conv':
                    c2 = div(args[0])
                    mod = Conv(ch[f], c2, *args[1:])
                    d_out = d_in * (args[2] if len(args) > 2 else 1)
                elif m == 'C3':
                    c2 = div(args[0])
                    mod = C3(ch[f], c2, n, *(args[1:2]))
                    d_out = d_in
                elif m == 'SPPF':
                    c2 = div(args[0])
                    mod = SPPF(ch[f], c2, *args[1:])
                    d_out = d_in
                elif m == 'nn.Upsample':
                    c2 = ch[f]
                    mod = nn.Upsample(None, args[1], args[2])
                    d_out = d_in // 2
                elif m == 'Concat':
                    c2 = sum(ch[x] for x in f)
                    mod = Concat(args[0])
                    d_out = d_in
                elif m == 'Detect':
                    mod = Detect(nc, anchors, [ch[x] for x in f])
                    mod.stride = torch.tensor([float(strides_div[x]) for x in f])
                    mod.anchors /= mod.stride.view(-1, 1, 1)
                    c2, d_out = None, d_in
                else:
                    raise ValueError(m)
                mod.i, mod.f, mod.type = i, f, 'models.common.' + m       # attributes the real parser attaches
                mod.np = sum(p.numel() for p in mod.parameters())
                layers.append(mod)
                if i == 0:
                    ch = []
                ch.append(c2)
                strides_div.append(d_out)
            self.model = nn.Sequential(*layers)
            self.save = sorted(set(x % len(layers) for m in layers for x in ([m.f] if isinstance(m.f, int) else m.f)
                                   if x != -1))
            self.stride = layers[-1].stride
            self.names = {0: 'animal', 1: 'person', 2: 'vehicle'} if nc == 3 else {i: str(i) for i in range(nc)}
            self.inplace = True

        def forward(self, x):
            y = []
            for m in self.model:
                if m.f != -1:
                    x = y[m.f] if isinstance(m.f, int) else [x if j == -1 else y[j] for j in m.f]
                x = m(x)
                y.append(x if m.i in self.save else None)
            return x

    for cls in (Conv, Bottleneck, C3, SPPF, Concat):
        cls.__module__ = 'models.common'
        cls.__qualname__ = cls.__name__
        setattr(common, cls.__name__, cls)
    for cls in (Detect, DetectionModel):
        cls.__module__ = 'models.yolo'
        cls.__qualname__ = cls.__name__
        setattr(yolo, cls.__name__, cls)
    yolo.Model = DetectionModel
    pkg.common, pkg.yolo = common, yolo
    sys.modules['models'], sys.modules['models.common'], sys.modules['models.yolo'] = pkg, common, yolo
    return common, yolo


def uninstall():
    for name in ('models', 'models.common', 'models.yolo'):
        sys.modules.pop(name, None)


def build_model(yaml, seed=0, gain=1.75):
    """
    A DetectionModel with random conv weights AND non-trivial BatchNorm statistics, in eval mode.  Scales are
    chosen as in megadetector_amd.weights_io.synthetic_weights (zero-mean kernels, gain 1.75 per Conv, 0.6 on the
    residual branch) so that activations stay O(1..10) through the 33 layers -- a network whose activations
    explode turns rounding differences into sign flips of the logits and tests nothing.  On the deep x6 topology
    1.75 already amplifies a perturbation ~4x per head C3 (measured: 5e-6 at layer 11 -> 2e-2 at layer 32); pass a
    smaller gain (1.3) for a network that is contractive like a trained one.
    """
    common, yolo = _install()
    torch.manual_seed(seed)
    model = yolo.DetectionModel(yaml)
    g = torch.Generator().manual_seed(seed + 1)
    residual_cv2 = set()
    for m in model.modules():
        if isinstance(m, common.Bottleneck) and m.add:
            residual_cv2.add(id(m.cv2))
    for m in model.modules():
        if isinstance(m, common.Conv):
            w = torch.randn(m.conv.weight.shape, generator=g)
            w -= w.mean(dim=(1, 2, 3), keepdim=True)
            fan = w.shape[1] * w.shape[2] * w.shape[3]
            g_conv = 0.6 if id(m) in residual_cv2 else gain
            nf = m.bn.num_features
            m.bn.weight.data = 0.8 + 0.4 * torch.rand(nf, generator=g)
            m.bn.bias.data = 0.1 * torch.randn(nf, generator=g)
            m.bn.running_mean.data = 0.1 * torch.randn(nf, generator=g)
            m.bn.running_var.data = 0.7 + 0.6 * torch.rand(nf, generator=g)
            m.conv.weight.data = w * (g_conv / fan ** 0.5)
        elif isinstance(m, yolo.Detect):
            for conv in m.m:
                fan = conv.weight.shape[1]
                conv.weight.data = torch.randn(conv.weight.shape, generator=g) * (3.0 / fan ** 0.5)
                b = 0.5 * torch.randn(m.na, m.no, generator=g)
                b[:, 4] -= 1.0
                conv.bias.data = b.view(-1)
    return model.eval()


def sparsify_objectness(model, x, score_thr=0.2, per_plane=(2, 6), logit_std=0.5):
    """
    Turns the dense, nearly input-independent predictions of a random-weight model into a camera-trap-like SPARSE set
    on the batch x.  A contractive random network (gain 1.3) forgets its input: the objectness logits of one (Detect
    level, anchor) plane differ by 0.01 .. 0.05 over all positions of all images, so any bias switches whole planes on
    or off.  Here, per plane,
      * the objectness row of the Detect conv is scaled so that the plane's logits have a standard deviation of
        `logit_std` over the batch (the bias keeps the plane's mean where it was) -- the rounding noise of a reduced
        storage type is amplified by the same factor, so `logit_std` IS the conditioning of the fixture (0.5: the
        input-dependent signal is 50 .. 100x a 16-bit mantissa's noise in fp16, 6 .. 12x in bf16);
      * the bias is then shifted so that only the k most confident anchors of the plane (per_plane[0] <= k <=
        per_plane[1], over the whole batch) have  obj * max(cls) > score_thr, k chosen so that the threshold falls into
        the WIDEST gap between consecutive scores: no anchor sits on the threshold, where a rounding difference would
        create / remove a detection (the reference's own comparison avoids that by running at its output threshold
        0.005 = its confidence bar, md_tests.py:100).
    Returns the number of anchors above the threshold per image.
    """
    common, yolo = _install()
    det = [m for m in model.modules() if isinstance(m, yolo.Detect)][0]
    with torch.no_grad():
        feats = {}
        hooks = [conv.register_forward_pre_hook(lambda mod, inp, i=i: feats.__setitem__(i, inp[0].detach().double()))
                 for i, conv in enumerate(det.m)]
        model(x)
        for h in hooks:
            h.remove()
        for i, conv in enumerate(det.m):
            f = feats[i]                                               # (B, C, ny, nx) Detect input of this level
            C = f.shape[1]
            f = f.permute(1, 0, 2, 3).reshape(C, -1)                   # (C, B * ny * nx)
            fbar = f.mean(1)
            W = conv.weight.data.view(det.na, det.no, C)
            b = conv.bias.data.view(det.na, det.no)
            for a in range(det.na):
                w0, b0 = W[a, 4].double(), float(b[a, 4])
                # the row is made orthogonal to the mean feature vector (its constant part moves into the bias), so that
                # amplifying it amplifies the input-dependent part only and the fp16 file holds O(1) numbers
                b1 = b0 + float(w0 @ fbar)
                w1 = w0 - (w0 @ fbar) / (fbar @ fbar) * fbar
                amp = logit_std / float((w1 @ f).std())
                w2 = (amp * w1).half().double()                        # what the fp16 checkpoint will hold
                obj_logit = w2 @ f + b1
                cls = torch.sigmoid(W[a, 5:].double() @ f + b[a, 5:].double()[:, None]).max(0)[0]
                # the objectness logit an anchor needs for obj * cls == score_thr
                need = torch.log((score_thr / cls) / (1 - score_thr / cls).clamp(min=1e-9))
                margin = torch.sort(obj_logit - need, descending=True)[0]            # > 0 <=> above the threshold
                lo, hi = per_plane
                best = None
                for k in range(lo, hi + 1):                                          # threshold between the k-th and the next
                    b2 = float(torch.tensor(b1 - 0.5 * float(margin[k - 1] + margin[k])).half())     # fp16-representable
                    m = margin + (b2 - b1)
                    if not (m[k - 1] > 0 > m[k]):
                        continue
                    clear = float(min(m[k - 1], -m[k]))
                    if best is None or clear > best[0]:
                        best = (clear, b2)
                assert best is not None
                W[a, 4] = w2.float()
                b[a, 4] = best[1]
    with torch.no_grad():
        p = model(x)
    score = (p[..., 4:5] * p[..., 5:]).max(-1)[0]
    return [int(v) for v in (score > score_thr).sum(1)]


def save_checkpoint(model, path):
    """Same container as yolov5's strip_optimizer leaves behind: fp16 module pickled whole."""
    import copy
    ck = {'epoch': -1, 'best_fitness': None, 'model': copy.deepcopy(model).half(), 'ema': None, 'updates': None,
          'optimizer': None, 'wandb_id': None, 'date': '2022-06-01T00:00:00'}
    for p in ck['model'].parameters():
        p.requires_grad = False
    torch.save(ck, path)
