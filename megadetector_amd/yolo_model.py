"""
Host-side model description for the HIP runtime: resolves a YOLOv5 yaml dict (the dict
pickled as `model.yaml` inside md_v5a.0.0.pt, or megadetector_amd.yolo_yaml.YOLOV5X6_MD) into
the flat layer / conv tables of include/mdhip.h, and holds BN-folded fp32 weights.

Mirrors what the reference obtains from `checkpoint['model'].float().fuse().eval()`
(reference megadetector/detection/pytorch_detector.py:957) -- a module list plus fused conv
weights -- without needing the yolov5 package.
"""

import math

import numpy as np

# module kinds of include/mdhip.h
MDHIP_CONV, MDHIP_C3, MDHIP_SPPF, MDHIP_UPSAMPLE, MDHIP_CONCAT, MDHIP_DETECT = range(6)


def _make_divisible(x, divisor):
    return int(math.ceil(x / divisor) * divisor)


class LayerSpec:
    __slots__ = ('index', 'type', 'frm', 'c_in', 'c_out', 'k', 's', 'p', 'n', 'shortcut',
                 'hidden', 'conv_names')

    def __init__(self, **kw):
        for name in self.__slots__:
            setattr(self, name, kw.get(name))


def resolve_yaml(yaml, ch=3):
    """
    yaml dict -> list[LayerSpec]; every 'from' is an absolute layer index (-1 = network input);
    conv_names lists the state_dict prefixes ('model.2.cv1.conv', ...) of the layer's convs in
    the order include/mdhip.h prescribes.
    """
    anchors, nc = yaml['anchors'], yaml['nc']
    gd, gw = yaml['depth_multiple'], yaml['width_multiple']
    na = (len(anchors[0]) // 2) if isinstance(anchors, (list, tuple)) else int(anchors)
    no = na * (nc + 5)
    out_ch = []
    specs = []
    rows = list(yaml['backbone']) + list(yaml['head'])
    for i, (f, n, m, args) in enumerate(rows):
        args = list(args)
        n_rep = max(round(n * gd), 1) if n > 1 else n
        frm = [f] if isinstance(f, int) else list(f)
        frm = [(i - 1 if x == -1 else (x if x >= 0 else i + x)) for x in frm]   # absolute; layer -1 = input
        c_in = ch if frm[0] < 0 else out_ch[frm[0]]
        pre = 'model.{}'.format(i)
        if m == 'Conv':
            c2 = args[0] if args[0] == no else _make_divisible(args[0] * gw, 8)
            k = args[1] if len(args) > 1 else 1
            s = args[2] if len(args) > 2 else 1
            p = args[3] if len(args) > 3 else k // 2
            spec = LayerSpec(index=i, type=MDHIP_CONV, frm=frm, c_in=c_in, c_out=c2, k=k, s=s, p=p,
                             n=1, shortcut=0, conv_names=[pre + '.conv'])
        elif m == 'C3':
            c2 = _make_divisible(args[0] * gw, 8)
            shortcut = args[1] if len(args) > 1 else True
            names = [pre + '.cv1.conv', pre + '.cv2.conv', pre + '.cv3.conv']
            for j in range(n_rep):
                names += ['{}.m.{}.cv1.conv'.format(pre, j), '{}.m.{}.cv2.conv'.format(pre, j)]
            spec = LayerSpec(index=i, type=MDHIP_C3, frm=frm, c_in=c_in, c_out=c2, k=1, s=1, p=0,
                             n=n_rep, shortcut=int(bool(shortcut)), hidden=int(c2 * 0.5),
                             conv_names=names)
        elif m == 'SPPF':
            c2 = _make_divisible(args[0] * gw, 8)
            k = args[1] if len(args) > 1 else 5
            spec = LayerSpec(index=i, type=MDHIP_SPPF, frm=frm, c_in=c_in, c_out=c2, k=k, s=1,
                             p=k // 2, n=1, shortcut=0, hidden=c_in // 2,
                             conv_names=[pre + '.cv1.conv', pre + '.cv2.conv'])
        elif m in ('nn.Upsample', 'Upsample'):
            if args[1] != 2 or (len(args) > 2 and args[2] != 'nearest'):
                raise ValueError('only nearest x2 upsampling is supported')
            spec = LayerSpec(index=i, type=MDHIP_UPSAMPLE, frm=frm, c_in=c_in, c_out=c_in, k=0, s=1,
                             p=0, n=1, shortcut=0, conv_names=[])
        elif m == 'Concat':
            c2 = sum(out_ch[x] for x in frm)
            spec = LayerSpec(index=i, type=MDHIP_CONCAT, frm=frm, c_in=c_in, c_out=c2, k=0, s=1, p=0,
                             n=1, shortcut=0, conv_names=[])
        elif m == 'Detect':
            spec = LayerSpec(index=i, type=MDHIP_DETECT, frm=frm, c_in=c_in, c_out=no, k=1, s=1, p=0,
                             n=1, shortcut=0,
                             conv_names=['{}.m.{}'.format(pre, l) for l in range(len(frm))])
        else:
            raise ValueError('unsupported YOLOv5 module "{}" at layer {}'.format(m, i))
        specs.append(spec)
        out_ch.append(spec.c_out if m != 'Detect' else None)
    return specs


def model_strides(specs):
    """Stride of every Detect input level, derived from the graph (== model.stride)."""
    div = []
    for s in specs:
        d = 1 if s.frm[0] < 0 else div[s.frm[0]]
        if s.type == MDHIP_CONV:
            d *= s.s
        elif s.type == MDHIP_UPSAMPLE:
            d //= 2
        div.append(d)
    det = specs[-1]
    if det.type != MDHIP_DETECT:
        return []
    return [float(div[f]) for f in det.frm]


class YoloWeights:
    """
    BN-folded fp32 weights of a YOLOv5 model plus its description.

    weights: dict  state_dict-style name -> np.float32 array, e.g.
             'model.0.conv.weight' (OIHW), 'model.0.conv.bias', 'model.33.m.0.weight',
             'model.33.anchors' ((nl,na,2), in units of the level's stride, as in the checkpoint)
    """

    def __init__(self, yaml, weights, names=None, source='unknown'):
        self.yaml = yaml
        self.specs = resolve_yaml(yaml)
        self.weights = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in weights.items()}
        self.nc = int(yaml['nc'])
        self.strides = model_strides(self.specs)
        self.nl = len(self.strides)
        det = self.specs[-1]
        if det.type == MDHIP_DETECT:
            a = self.weights['model.{}.anchors'.format(det.index)].reshape(self.nl, -1, 2)
            self.na = a.shape[1]
            self.anchors_px = np.ascontiguousarray(
                a * np.asarray(self.strides, dtype=np.float32).reshape(-1, 1, 1), dtype=np.float32)
        else:
            self.na = 0
            self.anchors_px = np.zeros((0, 0, 2), dtype=np.float32)
        self.names = names or {0: 'animal', 1: 'person', 2: 'vehicle'}
        self.source = source
        self._check()

    def _check(self):
        for s in self.specs:
            for name in s.conv_names:
                w = self.weights.get(name + '.weight')
                b = self.weights.get(name + '.bias')
                if w is None or b is None:
                    raise KeyError('missing fused weights for {}'.format(name))
                if w.ndim != 4 or b.shape != (w.shape[0],):
                    raise ValueError('bad weight shape for {}: {} / {}'.format(name, w.shape, b.shape))

    @property
    def max_stride(self):
        return int(max(self.strides)) if self.strides else 2

    def torch_state(self):
        """weights as torch tensors (for the oracle in tests/bench -- not used by the product)."""
        import torch
        return {k: torch.from_numpy(v.copy()) for k, v in self.weights.items()}
