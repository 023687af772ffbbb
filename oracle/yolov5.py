"""
ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the YOLOv5 forward pass the reference executes at
megadetector/detection/pytorch_detector.py:1313 (`self.model(batch)[0]`) on the module
built at :957 (`checkpoint['model'].float().fuse().eval()`).

The arithmetic lives in the third-party package ultralytics-yolov5==0.1.1
(reference pyproject.toml:70), which is absent from /root/reference and not installable
here.  This file restates that package's published algorithm (models/yolo.py:
parse_model, DetectionModel._forward_once, Detect.forward; models/common.py: Conv,
Bottleneck, C3, SPPF, Concat; utils/torch_utils.py: fuse_conv_and_bn) with
torch.nn.functional primitives -- the same CPU kernels the reference's CPU path runs.

PARITY UNPINNED for the conv stack: the reference tree holds no weights, no expected
logits and no expected detections for this path (they live in md-test-package.zip on
lila.science, reference md_tests.py:82).  What *is* pinned in-tree and checked in
tests/test_oracle_golden.py::test_topology_reproduces_published_flops_and_params: the topology reproduces the upstream FLOP/parameter counts
the reference cites (docs/release-notes/mdv1000-release.md:279: YOLOv5x6, 209.8 GFLOPs,
140.7 M params @640, nc=80).

Weights are a flat dict  name -> torch.Tensor  using the state_dict names of a *fused*
YOLOv5 model ("model.0.conv.weight", "model.2.m.0.cv1.conv.bias", "model.33.m.0.weight",
"model.33.anchors" ...).
"""

import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# parse_model (yolov5 models/yolo.py) restated
# --------------------------------------------------------------------------------------

def make_divisible(x, divisor):
    # yolov5 utils/general.py:make_divisible
    return math.ceil(x / divisor) * divisor


def parse_model(yaml, ch=3):
    """
    Resolve a yaml dict into a list of layer dicts
      {'i', 'f' (from, as in yaml), 'type', 'c1', 'c2', ...module args...}
    following yolov5 models/yolo.py:parse_model.
    """
    anchors, nc = yaml['anchors'], yaml['nc']
    gd, gw = yaml['depth_multiple'], yaml['width_multiple']
    na = len(anchors[0]) // 2
    no = na * (nc + 5)
    chs = [ch]
    layers = []
    for i, (f, n, m, args) in enumerate(yaml['backbone'] + yaml['head']):
        args = list(args)
        n_ = max(round(n * gd), 1) if n > 1 else n
        layer = {'i': i, 'f': f, 'type': m}
        if m in ('Conv', 'C3', 'SPPF'):
            c1 = chs[f]
            c2 = args[0]
            if c2 != no:
                c2 = make_divisible(c2 * gw, 8)
            layer.update(c1=c1, c2=c2)
            if m == 'Conv':
                k = args[1] if len(args) > 1 else 1
                s = args[2] if len(args) > 2 else 1
                p = args[3] if len(args) > 3 else k // 2   # autopad
                layer.update(k=k, s=s, p=p)
            elif m == 'C3':
                shortcut = args[1] if len(args) > 1 else True
                layer.update(n=n_, shortcut=shortcut, c_=int(c2 * 0.5))
            else:
                layer.update(k=args[1] if len(args) > 1 else 5, c_=c1 // 2)
        elif m == 'nn.Upsample':
            c2 = chs[f]
            layer.update(c2=c2, scale=args[1])
        elif m == 'Concat':
            c2 = sum(chs[x] for x in f)
            layer.update(c2=c2)
        elif m == 'Detect':
            layer.update(nc=nc, na=na, no=nc + 5, nl=len(anchors),
                         ch=[chs[x] for x in f], c2=None)
            c2 = None
        else:
            raise ValueError('unsupported module {}'.format(m))
        layers.append(layer)
        if i == 0:
            chs = []
        chs.append(c2)
    return layers


# --------------------------------------------------------------------------------------
# Conv+BN folding (yolov5 utils/torch_utils.py:fuse_conv_and_bn) restated
# --------------------------------------------------------------------------------------

def fuse_conv_bn(conv_w, bn_w, bn_b, bn_mean, bn_var, eps, conv_b=None):
    """Returns (w, b) of the single conv equivalent to conv(bias=conv_b) followed by BN(eval)."""
    c2 = conv_w.shape[0]
    w_bn = torch.diag(bn_w.div(torch.sqrt(eps + bn_var)))
    w = torch.mm(w_bn, conv_w.reshape(c2, -1)).view(conv_w.shape)
    b_conv = torch.zeros(c2, dtype=conv_w.dtype) if conv_b is None else conv_b
    b = torch.mm(w_bn, b_conv.reshape(-1, 1)).reshape(-1) + \
        (bn_b - bn_w.mul(bn_mean).div(torch.sqrt(bn_var + eps)))
    return w, b


# --------------------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------------------

def _bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _fp16_round(x):
    return x.to(torch.float16).to(torch.float32)


class Forward:
    """
    Functional YOLOv5 forward on CPU.

    emulate_bf16=False : the fp32 computation the reference performs (half_precision is
                         hard-wired False, reference pytorch_detector.py:848).
    emulate_bf16=True  : the *storage* rounding of the MI355X path is applied (weights and
                         every conv output rounded to bf16, fp32 accumulation, Detect
                         logits kept fp32) so the HIP kernels can be checked tightly;
                         residual adds and SiLU are evaluated in fp32 before rounding,
                         exactly as the fused HIP epilogue does.
    emulate_bf16='fp16': the same with fp16 storage (the HIP path's MDHIP_DTYPE_FP16 mode).
    emulate_bf16='fp8' : the MDHIP_DTYPE_FP8 mode (no upstream counterpart: the reference runs fp32 only,
                         pytorch_detector.py:848): bf16 storage as above, and in every C3 bottleneck whose hidden width
                         is a multiple of 16 the hidden tensor is quantised to OCP e4m3 with the per-tensor scale
                         fp8_scales[(layer, j)] (value = e4m3 x scale; the 1x1 conv's SiLU output times 1/scale,
                         clamped to +-448, rounded to nearest even, no bf16 rounding in between) and the 3x3 conv
                         runs on e4m3 weights quantised per output channel (scale = max|w| / 448 of the fp32 folded
                         weights), fp32 accumulation, accumulator x (tensor scale x channel scale) + bias.
    """

    def __init__(self, yaml, weights, emulate_bf16=False, keep=None, fp8_scales=None):
        self.yaml = yaml
        self.layers = parse_model(yaml)
        self.emulate = bool(emulate_bf16)
        self._round = _fp16_round if emulate_bf16 == 'fp16' else _bf16_round
        self.fp8 = emulate_bf16 == 'fp8'
        self.fp8_scales = dict(fp8_scales or {})
        self.w = {}
        self.w32 = {}
        for k, v in weights.items():
            v = v.detach().to(torch.float32)
            if self.fp8 and k.endswith('.weight'):
                self.w32[k] = v
            if emulate_bf16 and k.endswith('.weight'):
                v = self._round(v)
            self.w[k] = v
        self.keep = keep          # optional dict: layer index -> output tensor (NCHW fp32)
        det = self.layers[-1]
        assert det['type'] == 'Detect'
        self.stride = None

    # -- building blocks --------------------------------------------------------------
    def _conv(self, x, name, k, s, p, act=True, residual=None):
        w = self.w[name + '.weight']
        b = self.w[name + '.bias']
        y = F.conv2d(x, w, b, stride=s, padding=p)
        if act:
            y = F.silu(y)
        if residual is not None:
            y = residual + y
        if self.emulate:
            y = self._round(y)
        return y

    def _bottleneck_fp8(self, y1, pre, j, scale, shortcut):
        """one bottleneck with its hidden tensor and 3x3 weights in e4m3 (see the class docstring)"""
        n1, n2 = '{}.m.{}.cv1.conv'.format(pre, j), '{}.m.{}.cv2.conv'.format(pre, j)
        t = F.silu(F.conv2d(y1, self.w[n1 + '.weight'], self.w[n1 + '.bias']))
        s_t = torch.tensor(scale, dtype=torch.float32)
        q = (t * (torch.tensor(1.0, dtype=torch.float32) / s_t)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)
        w = self.w32[n2 + '.weight']
        amax = w.abs().amax(dim=(1, 2, 3))
        s_w = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
        wq = (w / s_w.view(-1, 1, 1, 1)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)
        acc = F.conv2d(q, wq, None, stride=1, padding=1)
        y = F.silu(acc * (s_t * s_w).view(1, -1, 1, 1) + self.w[n2 + '.bias'].view(1, -1, 1, 1))
        if shortcut:
            y = y1 + y
        return self._round(y)

    def _c3(self, x, L):
        pre = 'model.{}'.format(L['i'])
        y1 = self._conv(x, pre + '.cv1.conv', 1, 1, 0)
        y2 = self._conv(x, pre + '.cv2.conv', 1, 1, 0)
        for j in range(L['n']):
            if self.fp8 and (L['i'], j) in self.fp8_scales:
                y1 = self._bottleneck_fp8(y1, pre, j, self.fp8_scales[(L['i'], j)], L['shortcut'])
                continue
            t = self._conv(y1, '{}.m.{}.cv1.conv'.format(pre, j), 1, 1, 0)
            y1 = self._conv(t, '{}.m.{}.cv2.conv'.format(pre, j), 3, 1, 1,
                            residual=y1 if L['shortcut'] else None)
        return self._conv(torch.cat((y1, y2), 1), pre + '.cv3.conv', 1, 1, 0)

    def _sppf(self, x, L):
        pre = 'model.{}'.format(L['i'])
        k = L['k']
        x = self._conv(x, pre + '.cv1.conv', 1, 1, 0)
        y1 = F.max_pool2d(x, k, 1, k // 2)
        y2 = F.max_pool2d(y1, k, 1, k // 2)
        y3 = F.max_pool2d(y2, k, 1, k // 2)
        return self._conv(torch.cat((x, y1, y2, y3), 1), pre + '.cv2.conv', 1, 1, 0)

    def _detect(self, xs, L, in_hw):
        pre = 'model.{}'.format(L['i'])
        na, no, nl = L['na'], L['no'], L['nl']
        anchors = self.w[pre + '.anchors'].view(nl, na, 2)     # in units of stride
        z = []
        raw = []
        for l in range(nl):
            x = xs[l]
            w = self.w['{}.m.{}.weight'.format(pre, l)]
            b = self.w['{}.m.{}.bias'.format(pre, l)]
            x = F.conv2d(x, w, b)                              # fp32 logits, no activation
            bs, _, ny, nx = x.shape
            stride = float(in_hw[0]) / ny
            x = x.view(bs, na, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
            raw.append(x)
            yv, xv = torch.meshgrid(torch.arange(ny, dtype=torch.float32),
                                    torch.arange(nx, dtype=torch.float32), indexing='ij')
            grid = torch.stack((xv, yv), 2).expand(1, na, ny, nx, 2) - 0.5
            anchor_grid = (anchors[l] * stride).view(1, na, 1, 1, 2).expand(1, na, ny, nx, 2)
            s = x.sigmoid()
            xy, wh, conf = s.split((2, 2, no - 4), 4)
            xy = (xy * 2 + grid) * stride
            wh = (wh * 2) ** 2 * anchor_grid
            y = torch.cat((xy, wh, conf), 4)
            z.append(y.view(bs, na * ny * nx, no))
        return torch.cat(z, 1), raw

    # -- whole network ----------------------------------------------------------------
    def __call__(self, x):
        """x: (B,3,H,W) fp32 in [0,1].  Returns (B, n_anchors, 5+nc) fp32."""
        if self.emulate:
            x = self._round(x)
        in_hw = x.shape[2:]
        outs = []
        for L in self.layers:
            f = L['f']
            if isinstance(f, int):
                xin = x if f == -1 else outs[f]
            else:
                xin = [x if j == -1 else outs[j] for j in f]
            t = L['type']
            if t == 'Conv':
                x = self._conv(xin, 'model.{}.conv'.format(L['i']), L['k'], L['s'], L['p'])
            elif t == 'C3':
                x = self._c3(xin, L)
            elif t == 'SPPF':
                x = self._sppf(xin, L)
            elif t == 'nn.Upsample':
                x = F.interpolate(xin, scale_factor=L['scale'], mode='nearest')
            elif t == 'Concat':
                x = torch.cat(xin, 1)
            elif t == 'Detect':
                x, self.raw = self._detect(xin, L, in_hw)
            outs.append(x)
            if self.keep is not None and t != 'Detect':
                self.keep[L['i']] = x
        return x

    # -- test-time augmentation -------------------------------------------------------
    def forward_augment(self, x):
        """
        What `model(batch, augment=True)` computes (reference pytorch_detector.py:1313; the arithmetic lives
        in ultralytics-yolov5 0.1.1 models/yolo.py `_forward_augment`, `_descale_pred`, `_clip_augmented`
        and utils/torch_utils.py `scale_img`, restated here from their published behaviour): passes at
        scales 1 / 0.83 / 0.67, the second on the left-right flipped batch; `scale_img` = bilinear
        F.interpolate(align_corners=False) to (int(h*s), int(w*s)), padded right/bottom with 0.447 to a
        multiple of the largest stride; boxes divided by the scale, x un-flipped against the original
        width; the last (A//g) anchors of the first pass and the first (A//g)*4^(nl-1) anchors of the last
        pass dropped, g = sum(4^l); concatenation along the anchor axis.  Parity unpinned (no yolov5 here);
        second source: tests/test_oracle_tta_second_source.py (package-style restatement around an nn.Module).
        """
        import math
        det = self.layers[-1]
        nl = det['nl']
        h, w = x.shape[2:]
        gs = None
        ys = []
        if self.emulate:
            x = self._round(x)
        for si, flip in ((1.0, False), (0.83, True), (0.67, False)):
            xi = x.flip(3) if flip else x
            if si != 1.0:
                sh, sw = int(h * si), int(w * si)
                xi = F.interpolate(xi, size=(sh, sw), mode='bilinear', align_corners=False)
                if gs is None:
                    gs = self._max_stride()
                oh, ow = (math.ceil(v * si / gs) * gs for v in (h, w))
                xi = F.pad(xi, [0, ow - sw, 0, oh - sh], value=0.447)
            yi = self(xi).clone()
            yi[..., :4] /= si
            if flip:
                yi[..., 0] = w - yi[..., 0]
            ys.append(yi)
        g = sum(4 ** l for l in range(nl))
        i = (ys[0].shape[1] // g) * 1
        ys[0] = ys[0][:, :-i]
        i = (ys[-1].shape[1] // g) * 4 ** (nl - 1)
        ys[-1] = ys[-1][:, i:]
        return torch.cat(ys, 1)

    def _max_stride(self):
        # strides of the Detect inputs, from the graph (every stride-2 conv doubles, every upsample halves)
        div = []
        for L in self.layers:
            f = L['f'] if isinstance(L['f'], int) else L['f'][0]
            d = 1 if (f == -1 and not div) else div[f if f >= 0 else len(div) + f]
            if L['type'] == 'Conv':
                d *= L['s']
            elif L['type'] == 'nn.Upsample':
                d //= 2
            div.append(d)
        return max(div[j] for j in self.layers[-1]['f'])


# --------------------------------------------------------------------------------------
# work accounting (used to pin the topology and by bench.py for the roofline)
# --------------------------------------------------------------------------------------

def conv_shapes(yaml, h, w):
    """
    List every conv in execution order as dicts
      {'name','c1','c2','k','s','h_out','w_out','macs','params'}   (per image)
    """
    layers = parse_model(yaml)
    res = []
    hw = []          # output (h,w) per layer

    def add(name, c1, c2, k, s, hi, wi, p=None, bias_params=True, bn=True):
        p = k // 2 if p is None else p
        ho = (hi + 2 * p - k) // s + 1
        wo = (wi + 2 * p - k) // s + 1
        params = c1 * c2 * k * k + (2 * c2 if bn else c2)   # conv + BN(w,b) or conv bias
        res.append(dict(name=name, c1=c1, c2=c2, k=k, s=s, h_out=ho, w_out=wo,
                        macs=c1 * c2 * k * k * ho * wo, params=params))
        return ho, wo

    for L in layers:
        f = L['f']
        if isinstance(f, int):
            hi, wi = (h, w) if (L['i'] == 0) else hw[f if f >= 0 else L['i'] + f]
        else:
            hi, wi = hw[f[0] if f[0] >= 0 else L['i'] + f[0]]
        t = L['type']
        pre = 'model.{}'.format(L['i'])
        if t == 'Conv':
            ho, wo = add(pre + '.conv', L['c1'], L['c2'], L['k'], L['s'], hi, wi, p=L['p'])
        elif t == 'C3':
            c_ = L['c_']
            add(pre + '.cv1.conv', L['c1'], c_, 1, 1, hi, wi)
            add(pre + '.cv2.conv', L['c1'], c_, 1, 1, hi, wi)
            for j in range(L['n']):
                add('{}.m.{}.cv1.conv'.format(pre, j), c_, c_, 1, 1, hi, wi)
                add('{}.m.{}.cv2.conv'.format(pre, j), c_, c_, 3, 1, hi, wi)
            ho, wo = add(pre + '.cv3.conv', 2 * c_, L['c2'], 1, 1, hi, wi)
        elif t == 'SPPF':
            add(pre + '.cv1.conv', L['c1'], L['c_'], 1, 1, hi, wi)
            ho, wo = add(pre + '.cv2.conv', 4 * L['c_'], L['c2'], 1, 1, hi, wi)
        elif t == 'nn.Upsample':
            ho, wo = hi * L['scale'], wi * L['scale']
        elif t == 'Concat':
            ho, wo = hi, wi
        elif t == 'Detect':
            for l, fl in enumerate(f):
                hl, wl = hw[fl]
                add('{}.m.{}'.format(pre, l), L['ch'][l], L['na'] * L['no'], 1, 1, hl, wl, bn=False)
            ho, wo = None, None
        hw.append((ho, wo))
    return res


def count_work(yaml, h, w):
    """Returns (GMAC per image, number of convs, parameter count)."""
    shapes = conv_shapes(yaml, h, w)
    return (sum(s['macs'] for s in shapes) / 1e9, len(shapes), sum(s['params'] for s in shapes))
