"""
Thin Python owner of one mdhip context (one GPU): turns YoloWeights into the C model
description, and exposes the hot-path stages with numpy in / numpy out.
All arithmetic happens in libmdhip.so (hand-written HIP); nothing here computes.
"""

import ctypes as C
import json
import os

import numpy as np

from . import _lib
from ._lib import HipError
from .yolo_model import MDHIP_DETECT


class HipContext:

    def __init__(self, weights, device=0, dtype='bf16', max_batch=32, max_h=1280, max_w=1280):
        self.lib = _lib.load()
        self.weights = weights
        self.device = int(device)
        self.max_batch = int(max_batch)
        self._keep = []
        specs = weights.specs
        convs = []
        layers = (_lib.mdhip_layer * len(specs))()
        for i, s in enumerate(specs):
            L = layers[i]
            L.type = s.type
            L.n_from = len(s.frm)
            for j, f in enumerate(s.frm):
                L.from_[j] = f
            L.c_out = s.c_out if s.c_out is not None else 0
            L.k, L.s, L.p = s.k or 0, s.s or 1, s.p or 0
            L.n = s.n or 1
            L.shortcut = s.shortcut or 0
            L.first_conv = len(convs)
            for name in s.conv_names:
                w = weights.weights[name + '.weight']
                b = weights.weights[name + '.bias']
                self._keep += [w, b]
                cv = _lib.mdhip_conv()
                cv.weight = w.ctypes.data_as(C.POINTER(C.c_float))
                cv.bias = b.ctypes.data_as(C.POINTER(C.c_float))
                cv.c_out, cv.c_in, cv.kh, cv.kw = w.shape
                convs.append(cv)
        conv_arr = (_lib.mdhip_conv * max(1, len(convs)))(*convs)
        m = _lib.mdhip_model()
        m.n_layers = len(specs)
        m.layers = layers
        m.n_convs = len(convs)
        m.convs = conv_arr
        m.nc = weights.nc
        m.na = weights.na
        m.nl = weights.nl
        anchors = np.ascontiguousarray(weights.anchors_px.reshape(-1), dtype=np.float32)
        strides = np.ascontiguousarray(np.asarray(weights.strides, dtype=np.float32))
        self._keep += [anchors, strides, layers, conv_arr]
        m.anchors_px = anchors.ctypes.data_as(C.POINTER(C.c_float)) if anchors.size else None
        m.strides = strides.ctypes.data_as(C.POINTER(C.c_float)) if strides.size else None
        dt = {'bf16': _lib.MDHIP_DTYPE_BF16, 'fp8': _lib.MDHIP_DTYPE_FP8, 'fp16': _lib.MDHIP_DTYPE_FP16}[dtype]
        self.dtype = dtype
        handle = C.c_void_p()
        rc = self.lib.mdhip_create(C.byref(m), self.device, dt, int(max_batch), int(max_h), int(max_w),
                                   C.byref(handle))
        if rc != 0:
            raise HipError('mdhip_create failed ({}): {}'.format(
                rc, self.lib.mdhip_last_error(None).decode()))
        self.h = handle
        self.no = weights.nc + 5
        self.has_detect = specs[-1].type == MDHIP_DETECT
        self.max_stride = self.lib.mdhip_max_stride(self.h)
        self.load_tuned()

    TUNED_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tuned_cfgs.json')

    def load_tuned(self, path=None):
        """
        Hands the measured tile choices (tools/autotune.py) to the library; returns the number of
        entries.  Shapes without an entry use the built-in heuristic, so a missing or stale file only
        costs speed.
        """
        if path is None:
            # a table measured for this storage type, if there is one (tuned_cfgs_fp16.json), else the bf16 table
            path = self.TUNED_PATH
            alt = self.TUNED_PATH.replace('.json', '_{}.json'.format(getattr(self, 'dtype', 'bf16')))
            if getattr(self, 'dtype', 'bf16') != 'bf16' and os.path.exists(alt):
                path = alt
        if not os.path.exists(path):
            return 0
        try:
            entries = json.load(open(path)).get('entries', [])
        except Exception:
            return 0
        ncfg = self.lib.mdhip_num_conv_cfgs()
        # an entry names its configuration; the id is looked up in THIS build (ids shift when a kernel family is
        # added or removed), entries without a name keep their id, entries naming an unknown configuration are dropped
        by_name = {self.lib.mdhip_conv_cfg_name(i).decode(): i for i in range(ncfg)}
        resolved = []
        for e in entries:
            if e.get('name'):
                if e['name'] not in by_name:
                    continue
                e = dict(e, cfg=by_name[e['name']])
            if 0 <= int(e['cfg']) < ncfg:
                resolved.append(e)
        entries = resolved
        arr = (_lib.mdhip_tuned * max(1, len(entries)))()
        for i, e in enumerate(entries):
            arr[i].m, arr[i].n, arr[i].k = int(e['m']), int(e['n']), int(e['k'])
            arr[i].ntaps, arr[i].stride = int(e['ntaps']), int(e['stride'])
            arr[i].has_res, arr[i].cfg = int(e['has_res']), int(e['cfg'])
            arr[i].batch = int(e.get('batch', 32))
        self._check(self.lib.mdhip_set_tuned(self.h, arr, len(entries)), 'mdhip_set_tuned')
        return len(entries)

    # -- plumbing ---------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != 0:
            raise HipError('{} failed ({}): {}'.format(what, rc, self.lib.mdhip_last_error(self.h).decode()))

    def close(self):
        if getattr(self, 'h', None) is not None and self.h.value:
            self.lib.mdhip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- hot path ---------------------------------------------------------------------
    def preprocess(self, images, geoms, out_h, out_w, stream=0):
        """
        images: list of HWC uint8 RGB numpy arrays (host) or integer device pointers.
        geoms:  list of (src_h, src_w, resized_h, resized_w, top, left[, interp]); interp 0 = cv2.INTER_LINEAR
                (default), 1 = cv2.INTER_AREA.
        """
        n = len(images)
        ptrs = (C.c_void_p * n)()
        hold = []
        for i, im in enumerate(images):
            if isinstance(im, np.ndarray):
                if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
                    raise ValueError('image {} must be HWC uint8 RGB'.format(i))
                im = np.ascontiguousarray(im)
                hold.append(im)
                ptrs[i] = im.ctypes.data
            else:
                ptrs[i] = int(im)
        g = (_lib.mdhip_letterbox * n)()
        for i, q in enumerate(geoms):
            g[i].src_h, g[i].src_w, g[i].resized_h, g[i].resized_w, g[i].top, g[i].left = [int(v) for v in q[:6]]
            g[i].interp = int(q[6]) if len(q) > 6 else 0
        self._check(self.lib.mdhip_preprocess(self.h, C.cast(ptrs, C.POINTER(C.c_void_p)), g, n, int(out_h), int(out_w),
                                              C.c_void_p(stream)), 'mdhip_preprocess')

    def forward(self, n, h, w, stream=0):
        self._check(self.lib.mdhip_forward(self.h, int(n), int(h), int(w), C.c_void_p(stream)), 'mdhip_forward')

    def forward_tta(self, n, h, w, stream=0):
        """augmented inference (yolov5 _forward_augment) on the batch left by preprocess()"""
        self._check(self.lib.mdhip_forward_tta(self.h, int(n), int(h), int(w), C.c_void_p(stream)), 'mdhip_forward_tta')

    def last_num_anchors(self):
        return self.lib.mdhip_last_num_anchors(self.h)

    # -- fp8 mode (dtype='fp8') -----------------------------------------------------------
    def calibrate(self, n, h, w, stream=0):
        """records the ranges of the e4m3 tensors on the batch left by preprocess() and derives their scales"""
        self._check(self.lib.mdhip_calibrate(self.h, int(n), int(h), int(w), C.c_void_p(stream)), 'mdhip_calibrate')

    def fp8_scales(self):
        """[(scale, model layer, op index)] of the e4m3 tensors, in execution order"""
        n = self.lib.mdhip_fp8_num_tensors(self.h)
        if n <= 0:
            return []
        sc = (C.c_float * n)()
        ly = (C.c_int32 * n)()
        op = (C.c_int32 * n)()
        self.lib.mdhip_fp8_get_scales(self.h, sc, ly, op, n)
        return [(float(sc[i]), int(ly[i]), int(op[i])) for i in range(n)]

    def set_fp8_scales(self, scales):
        arr = (C.c_float * len(scales))(*[float(v) for v in scales])
        self._check(self.lib.mdhip_fp8_set_scales(self.h, arr, len(scales)), 'mdhip_fp8_set_scales')

    def nms(self, n, conf_thres, iou_thres, max_det=300, stream=0):
        out = np.empty((n, max_det, 6), dtype=np.float32)
        counts = np.empty((n,), dtype=np.int32)
        self._check(self.lib.mdhip_nms(self.h, int(n), float(conf_thres), float(iou_thres), int(max_det),
                                       _lib.np_ptr(out), _lib.np_ptr(counts), C.c_void_p(stream)), 'mdhip_nms')
        return out, counts

    def nms_enqueue(self, n, conf_thres, iou_thres, max_det=300, slot=0, stream=0):
        """asynchronous NMS + D2H into pinned slot `slot`; pair with nms_wait(slot)"""
        self._check(self.lib.mdhip_nms_enqueue(self.h, int(n), float(conf_thres), float(iou_thres), int(max_det),
                                               int(slot), C.c_void_p(stream)), 'mdhip_nms_enqueue')
        self._slot_shape = getattr(self, '_slot_shape', {})
        self._slot_shape[slot] = (int(n), int(max_det))

    def nms_wait(self, slot=0):
        """blocks until slot is complete; returns numpy *views* of the pinned slot (valid until re-enqueued)"""
        out = C.POINTER(C.c_float)()
        cnt = C.POINTER(C.c_int32)()
        self._check(self.lib.mdhip_nms_wait(self.h, int(slot), C.byref(out), C.byref(cnt)), 'mdhip_nms_wait')
        n, max_det = self._slot_shape[slot]
        det = np.ctypeslib.as_array(out, shape=(n, max_det, 6))
        counts = np.ctypeslib.as_array(cnt, shape=(n,))
        return det, counts

    def nms_on(self, pred, conf_thres, iou_thres, max_det=300, stream=0):
        pred = np.ascontiguousarray(pred, dtype=np.float32)
        n, a, no = pred.shape
        if no != self.no:
            raise ValueError('prediction width {} != {}'.format(no, self.no))
        out = np.empty((n, max_det, 6), dtype=np.float32)
        counts = np.empty((n,), dtype=np.int32)
        self._check(self.lib.mdhip_nms_on(self.h, _lib.np_ptr(pred), n, a, float(conf_thres), float(iou_thres),
                                          int(max_det), _lib.np_ptr(out), _lib.np_ptr(counts),
                                          C.c_void_p(stream)), 'mdhip_nms_on')
        return out, counts

    # -- introspection ------------------------------------------------------------------
    def num_anchors(self, h, w):
        return self.lib.mdhip_num_anchors(self.h, int(h), int(w))

    def read_predictions(self, n, h=None, w=None, stream=0):
        out = np.empty((n, self.last_num_anchors(), self.no), dtype=np.float32)
        self._check(self.lib.mdhip_read_predictions(self.h, n, _lib.np_ptr(out), C.c_void_p(stream)),
                    'mdhip_read_predictions')
        return out

    def read_input(self, n, h, w, stream=0):
        out = np.empty((n, 3, h, w), dtype=np.float32)
        self._check(self.lib.mdhip_read_input(self.h, n, h, w, _lib.np_ptr(out), C.c_void_p(stream)), 'mdhip_read_input')
        return out

    def read_layer(self, layer, n, stream=0):
        c, h, w = C.c_int(), C.c_int(), C.c_int()
        self._check(self.lib.mdhip_read_layer(self.h, layer, n, None, C.byref(c), C.byref(h), C.byref(w),
                                              C.c_void_p(stream)), 'mdhip_read_layer')
        out = np.empty((n, c.value, h.value, w.value), dtype=np.float32)
        self._check(self.lib.mdhip_read_layer(self.h, layer, n, _lib.np_ptr(out), C.byref(c), C.byref(h),
                                              C.byref(w), C.c_void_p(stream)), 'mdhip_read_layer')
        return out

    def num_ops(self):
        return self.lib.mdhip_num_ops(self.h)

    def op_infos(self):
        res = []
        for i in range(self.num_ops()):
            info = _lib.mdhip_op_info()
            self._check(self.lib.mdhip_get_op_info(self.h, i, C.byref(info)), 'mdhip_get_op_info')
            res.append(dict(op=i, name=info.name.decode(), kind=info.kind, layer=info.layer, m=info.m,
                            n=info.n, k=info.k, flops=info.flops, bytes=info.bytes, cfg=info.cfg,
                            ntaps=info.ntaps, stride=info.stride, has_res=info.has_res))
        return res

    def forward_timed(self, n, h, w, stream=0):
        ms = np.zeros((self.num_ops(),), dtype=np.float32)
        self._check(self.lib.mdhip_forward_timed(self.h, n, h, w, _lib.np_ptr(ms), C.c_void_p(stream)),
                    'mdhip_forward_timed')
        return ms

    def time_forwards(self, enable=True):
        """bracket every forward() with a HIP event pair on its stream (see forward_times)"""
        self._check(self.lib.mdhip_time_forwards(self.h, 1 if enable else 0), 'mdhip_time_forwards')

    def forward_times(self, max_n=64):
        """durations (ms) of the most recent forwards measured since time_forwards(True), oldest first"""
        ms = np.zeros((max_n,), dtype=np.float32)
        n = self.lib.mdhip_forward_times(self.h, _lib.np_ptr(ms), int(max_n))
        if n < 0:
            self._check(n, 'mdhip_forward_times')
        return ms[:n].copy()

    def set_op_cfg(self, op, cfg):
        self._check(self.lib.mdhip_set_op_cfg(self.h, op, cfg), 'mdhip_set_op_cfg')

    def set_fuse(self, on):
        """fused bottleneck launches on (default) / off (the 1x1 and the 3x3 as two launches: same bits)"""
        self._check(self.lib.mdhip_set_fuse(self.h, 1 if on else 0), 'mdhip_set_fuse')

    def set_option(self, name, value):
        """named integer switch of the context (include/mdhip.h: mdhip_set_option)"""
        self._check(self.lib.mdhip_set_option(self.h, name.encode(), int(value)), 'mdhip_set_option({})'.format(name))

    def set_graph(self, mode, max_n=0):
        """graph replay of forward(): 0 / False = off, 1 / True = every forward, 2 / 'auto' = forwards of at most max_n
        images (default 8); same kernels and arguments, bit-identical results (include/mdhip.h: mdhip_set_graph)"""
        mode = {'off': 0, 'on': 1, 'auto': 2, False: 0, True: 1}.get(mode, mode)
        self._check(self.lib.mdhip_set_graph(self.h, int(mode), int(max_n)), 'mdhip_set_graph')

    def op_supports_cfg(self, op, cfg):
        return self.lib.mdhip_op_supports_cfg(self.h, int(op), int(cfg)) == 1

    def conv_cfg_name(self, cfg):
        return self.lib.mdhip_conv_cfg_name(int(cfg)).decode()

    def cfg_is_bitwise(self, cfg):
        return self.lib.mdhip_cfg_is_bitwise(int(cfg)) == 1

    def num_conv_cfgs(self):
        return self.lib.mdhip_num_conv_cfgs()

    def time_op(self, op, n, h, w, iters=10, stream=0):
        ms = C.c_float()
        self._check(self.lib.mdhip_time_op(self.h, op, n, h, w, iters, C.byref(ms), C.c_void_p(stream)), 'mdhip_time_op')
        return ms.value
