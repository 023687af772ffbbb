#!/usr/bin/env python3
"""
What would e4m3 STORAGE of the tensors that only 1x1 / stride-2 convs read cost in accuracy?  (VERDICT r3 item 7:
"fp8, second half"; BASELINE.json configs[4].)  A study on the CPU with the oracle -- test infrastructure, nothing
of the product path runs here -- so that the row can be closed with a number instead of an argument.

Network and images are the ones of tests/test_gpu_precision_x6.py (x6 checkpoint FILE, gain 1.3 = contractive like a
trained network; 2560-pixel originals letterboxed to 640; reference = the file's own fp32 forward,
pytorch_detector.py:957,1313; bar = md_tests.py:96-100: 0.005 on the confidence).  Scales are calibrated on OTHER
images, with the product's rule (largest magnitude seen x 2 / 448, mdhip_capi.cpp kFp8RangeMargin).

Modes:
  bf16          every tensor in bf16 (BASELINE configs[1])
  fp8           today's contract: the hidden tensor of every bottleneck + its 3x3 weights in e4m3
  fp8-all       + every tensor that is read only by 1x1 / stride-2 convs stored as e4m3 (per-tensor scale, written from
                the fp32 epilogue): stem and stride-2 conv outputs, C3 cv2 / cv3 outputs, the last bottleneck's output,
                the whole bottleneck chain of the head's C3s (no residual there), SPPF, the head's 1x1 convs.  The
                consumers read e4m3 activations against their 16-bit weights (the most favourable variant: weights
                not quantised).  Tensors that feed a residual add stay in bf16.
  fp8-all-w     the same with the consumers' weights in e4m3 too (per output channel), which is what an e4m3 MFMA needs
  [r5] the same four under MX BLOCK SCALES (OCP microscaling: every 32 consecutive channels of a pixel share one E8M0
  scale 2^(floor(log2 amax) - 8), computed by the producer from its fp32 values -- the operand `v_mfma_scale_f32_16x16x128
  _f8f6f4` takes per 32 k and that conv_f8.cpp feeds with unit scales today; no calibration, no static ranges):
  mx            today's contract with the hidden tensors block-scaled (3x3 weights per output channel as today)
  mx-w          + the 3x3 weights block-scaled along K too (32 input channels of one tap share a scale)
  mx-all        + the 71 tensors that only 1x1 / stride-2 convs read, block-scaled e4m3; consumers' weights 16-bit
  mx-all-w      + the consumers' weights block-scaled e4m3 (what an all-e4m3 MFMA path needs)

Usage: python tests/fp8_all_storage_study.py [--size 640] > profiles/rN_fp8_all_storage_study.txt
"""

import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

from oracle import yolov5 as Y  # noqa: E402

MARGIN = 2.0


def e4m3(t, scale):
    s = torch.tensor(scale, dtype=torch.float32)
    return (t * (torch.tensor(1.0, dtype=torch.float32) / s)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32) * s


def mx_e4m3(t, dim=1, block=32):
    """OCP MX e4m3: blocks of `block` consecutive elements along `dim` share the power-of-two scale
    2^(floor(log2(max|x|)) - 8) (8 = emax of e4m3: the block's largest element lands in [256, 512) -> saturates at 448 like
    the spec's conversion); elements are rounded to e4m3 (RNE) after the division.  Returns the dequantised tensor."""
    t = t.movedim(dim, -1)
    shp = t.shape
    c = shp[-1]
    pad = (-c) % block
    if pad:
        t = F.pad(t, (0, pad))
    b = t.reshape(*t.shape[:-1], -1, block)
    amax = b.abs().amax(-1, keepdim=True)
    e = torch.floor(torch.log2(amax.clamp(min=2.0 ** -120))) - 8.0
    scale = torch.pow(torch.tensor(2.0, dtype=torch.float32), e.clamp(-127.0, 127.0))
    q = (b / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32) * scale
    q = q.reshape(*t.shape)[..., :c].reshape(shp)
    return q.movedim(-1, dim)


class StudyForward(Y.Forward):
    """oracle forward with e4m3 storage of a named set of conv outputs (the names are decided by `wanted`)"""

    def __init__(self, yaml, weights, mode, scales=None, hidden_scales=None):
        super().__init__(yaml, weights, emulate_bf16='fp8' if mode != 'bf16' else True, fp8_scales=hidden_scales)
        self.mx = mode.startswith('mx')
        self.mx_w3 = mode in ('mx-w', 'mx-all-w')          # bottleneck 3x3 weights block-scaled along K
        if self.mx:
            mode = {'mx': 'fp8', 'mx-w': 'fp8', 'mx-all': 'fp8-all', 'mx-all-w': 'fp8-all-w'}[mode]
        self.mode = mode
        self.scales = scales                 # None: record ranges (calibration pass)
        self.amax = {}
        self.hidden_amax = {}
        self.qnames = set()
        self.residual_c3 = {L['i'] for L in self.layers if L['type'] == 'C3' and L['shortcut']}
        self.n_of = {L['i']: L['n'] for L in self.layers if L['type'] == 'C3'}
        self.q_weights = mode == 'fp8-all-w'

    def wanted(self, name):
        """is the OUTPUT of conv `name` read only by 1x1 / stride-2 convs (and not by a residual add)?"""
        parts = name.split('.')              # model.<i>.conv | model.<i>.cvK.conv | model.<i>.m.<j>.cvK.conv
        i = int(parts[1])
        if len(parts) == 3:
            return True                      # plain Conv layers: stem, stride-2 convs, the head's 1x1 convs
        if parts[2] in ('cv2', 'cv3'):
            return True                      # C3 cv2 (read by cv3), cv3 (read by the next stride-2 conv / concat -> 1x1s); SPPF cv2
        if parts[2] == 'cv1':
            if i not in self.n_of:
                return True                  # SPPF cv1 (max pools + the 1x1 cv2)
            return i not in self.residual_c3  # C3 cv1 = first y1: feeds bottleneck 0's residual in the backbone
        if parts[2] == 'm':
            j = int(parts[3])
            if parts[4] == 'cv1':
                return False                 # hidden tensor: today's contract handles it
            return i not in self.residual_c3 or j == self.n_of[i] - 1
        return False

    def _conv(self, x, name, k, s, p, act=True, residual=None):
        if self.mode in ('bf16', 'fp8'):
            return super()._conv(x, name, k, s, p, act, residual)
        w = self.w[name + '.weight']
        if self.q_weights and (k == 1 or s == 2) and name != 'model.0.conv':      # (the stem reads the image)
            w32 = self.w32[name + '.weight']
            amax = w32.abs().amax(dim=(1, 2, 3))
            s_w = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax)).view(-1, 1, 1, 1)
            w = (w32 / s_w).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32) * s_w
            if self.mx:
                w = mx_e4m3(w32, dim=1)
        y = F.conv2d(x, w, self.w[name + '.bias'], stride=s, padding=p)
        if act:
            y = F.silu(y)
        if residual is not None:
            y = residual + y
        if self.wanted(name):
            self.qnames.add(name)
            if self.scales is None:
                self.amax[name] = max(self.amax.get(name, 0.0), float(y.abs().max()))
                y = self._round(y)
            elif self.mx:
                y = mx_e4m3(y, dim=1)
            else:
                y = e4m3(y, self.scales[name])
            return y
        return self._round(y)

    def _bottleneck_fp8(self, y1, pre, j, scale, shortcut):
        if scale is None:                    # calibration pass: 16-bit bottleneck, range of the hidden tensor recorded
            t = Y.Forward._conv(self, y1, '{}.m.{}.cv1.conv'.format(pre, j), 1, 1, 0)
            key = (int(pre.split('.')[1]), j)
            self.hidden_amax[key] = max(self.hidden_amax.get(key, 0.0), float(t.abs().max()))
            return self._conv(t, '{}.m.{}.cv2.conv'.format(pre, j), 3, 1, 1, residual=y1 if shortcut else None)
        if self.mx:
            # hidden tensor block-scaled from the 1x1's fp32 epilogue values; 3x3 weights per output channel (today's
            # packing) or block-scaled along K; fp32 accumulation; the scales ride with the operands (v_mfma_scale)
            n1, n2 = '{}.m.{}.cv1.conv'.format(pre, j), '{}.m.{}.cv2.conv'.format(pre, j)
            t = F.silu(F.conv2d(y1, self.w[n1 + '.weight'], self.w[n1 + '.bias']))
            q = mx_e4m3(t, dim=1)
            w = self.w32[n2 + '.weight']
            if self.mx_w3:
                wq = mx_e4m3(w, dim=1)
            else:
                amax = w.abs().amax(dim=(1, 2, 3))
                s_w = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax)).view(-1, 1, 1, 1)
                wq = (w / s_w).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32) * s_w
            y = F.silu(F.conv2d(q, wq, self.w[n2 + '.bias'], stride=1, padding=1))
            if shortcut:
                y = y1 + y
            y = self._round(y)
        else:
            y = super()._bottleneck_fp8(y1, pre, j, scale, shortcut)
        name = '{}.m.{}.cv2.conv'.format(pre, j)
        if self.mode.startswith('fp8-all') and self.scales is not None and self.wanted(name):
            # (super() rounded to bf16 first: one extra rounding on these few tensors, far below the e4m3 step)
            self.qnames.add(name)
            y = mx_e4m3(y, dim=1) if self.mx else e4m3(y, self.scales[name])
        return y

    def _c3(self, x, L):
        if self.scales is None and self.mode != 'bf16':
            # calibration: every bottleneck in 16 bits, ranges recorded
            self.fp8_scales = {(L['i'], j): None for j in range(L['n'])}
        return super()._c3(x, L)


def main():
    size = 640
    if '--size' in sys.argv:
        size = int(sys.argv[sys.argv.index('--size') + 1])
    import fake_yolov5 as FY
    import parity_util as PU
    from megadetector_amd import weights_io, yolo_yaml
    torch.set_num_threads(os.cpu_count() or 8)
    model = FY.build_model(yolo_yaml.YOLOV5X6_MD, seed=7, gain=1.3)
    path = '/tmp/fp8_study_x6.pt'
    FY.save_checkpoint(model, path)
    ref_model = model.half().float()
    imgs = PU.random_images(2, 2560, 2560, seed=71)
    calib = PU.random_images(2, 2560, 2560, seed=1234)
    x, _ = PU.oracle_input(imgs, size, 64)
    xc, _ = PU.oracle_input(calib, size, 64)
    with torch.no_grad():
        ref = ref_model(x).numpy()
    del model, ref_model
    FY.uninstall()
    W = weights_io.load_checkpoint(path)
    state = W.torch_state()
    score = (ref[..., 4:5] * ref[..., 5:]).max(-1)
    print('# x6 checkpoint (gain 1.3), {} px, {} anchors above 0.1, {} above 0.005; reference = fp32 forward of the file'.format(
        size, int((score > 0.1).sum()), int((score > 0.005).sum())))

    # calibration pass (16-bit everywhere, ranges of every candidate tensor + the hidden tensors) on OTHER images
    cal = StudyForward(W.yaml, state, 'fp8-all', scales=None)
    with torch.no_grad():
        cal(xc)
    scales = {k: max(v, 1e-20) * MARGIN / 448.0 for k, v in cal.amax.items()}
    hidden = {k: max(v, 1e-20) * MARGIN / 448.0 for k, v in cal.hidden_amax.items()}
    print('# calibrated on 2 other images: {} hidden tensors (today\'s contract), {} further tensors for "all"'.format(len(hidden), len(scales)))

    print('{:10s} {:>12s} {:>16s} {:>16s} {:>12s}'.format('mode', 'max|dconf|', 'max|dconf| >0.1', 'max|dscore|>.005', 'box rel max'))
    for mode in ('bf16', 'fp8', 'fp8-all', 'fp8-all-w', 'mx', 'mx-w', 'mx-all', 'mx-all-w'):
        fw = StudyForward(W.yaml, state, mode, scales=scales, hidden_scales=hidden if mode != 'bf16' else None)
        fw.q_weights = fw.mode == 'fp8-all-w'
        with torch.no_grad():
            got = fw(x).numpy()
        d = np.abs(got[..., 4:] - ref[..., 4:])
        hi = score > 0.1
        mid = score > 0.005
        sc = (got[..., 4:5] * got[..., 5:]).max(-1)
        e = PU.rel_err(got[..., :4], ref[..., :4])
        print('{:10s} {:12.5f} {:16.5f} {:16.5f} {:12.2e}   ({} tensors in e4m3 besides the hidden ones)'.format(
            mode, d.max(), d[hi].max(), float(np.abs(sc - score)[mid].max()), e[0], len(fw.qnames)))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
