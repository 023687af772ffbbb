"""
Host-side box rescale + MegaDetector output formatting, vectorised over the detections of
one image.

Replaces the per-detection Python loop of the reference
(megadetector/detection/pytorch_detector.py:1361-1422, 'classic' branch) while keeping its
arithmetic bit for bit: fp32 for everything the reference does on torch tensors
(scale_coords, clip, round, xyxy2xywh, division by the int64 `gn`), float64 for what it does on
Python floats afterwards (convert_yolo_to_xywh at ct_utils.py:255-270, truncate_float at
ct_utils.py:82-103).
"""

import math

import numpy as np

from .constants import CONF_DIGITS, COORD_DIGITS

_F = np.float32


def letterbox_geometry(shape_hw, new_shape=1280, stride=64, auto=True, scaleup=True):
    """
    yolov5 letterbox() ratio / padding arithmetic (restated in-tree at reference
    pytorch_detector.py:434-454).  Returns dict(ratio, pad, new_unpad (w,h), top, left, out_hw).
    """
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    h, w = int(shape_hw[0]), int(shape_hw[1])
    r = min(new_shape[0] / h, new_shape[1] / w)
    if not scaleup:
        r = min(r, 1.0)
    new_unpad = (int(round(w * r)), int(round(h * r)))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = int(np.mod(dw, stride)), int(np.mod(dh, stride))
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return dict(ratio=(r, r), pad=(dw, dh), new_unpad=new_unpad, top=top, left=left,
                out_hw=(new_unpad[1] + top + bottom, new_unpad[0] + left + right))


def modern_geometry(shape_hw, image_size=1280, stride=64, use_ceil=False):
    """
    compatibility_mode != 'classic' (reference pytorch_detector.py:1036-1109): the image is first resized so
    that its long side is image_size (cv2.INTER_LINEAR when growing, cv2.INTER_AREA when shrinking; int() or,
    with 'use_ceil_for_resize', ceil()), then padded -- letterbox(auto=False, scaleup=False) -- into
    ceil(normalised_shape * image_size / stride + 0.5) * stride.  Returns dict(resized_hw, interp (0 linear /
    1 area), target_shape, letterbox = letterbox_geometry of the resized image).
    """
    h, w = int(shape_hw[0]), int(shape_hw[1])
    ratio = image_size / max(h, w)
    rh, rw, interp = h, w, 0
    if ratio != 1:
        interp = 0 if ratio > 1 else 1
        rw = math.ceil(w * ratio) if use_ceil else int(w * ratio)
        rh = math.ceil(h * ratio) if use_ceil else int(h * ratio)
    md = max(rh, rw, 3)                    # max(img_original.shape): the channel count takes part
    norm = np.array([rh / md, rw / md])
    target = np.ceil((norm * image_size) / stride + 0.5).astype(int) * stride
    g = letterbox_geometry((rh, rw), new_shape=(int(target[0]), int(target[1])), stride=stride, auto=False,
                           scaleup=False)
    return dict(resized_hw=(rh, rw), interp=interp, target_shape=(int(target[0]), int(target[1])), letterbox=g)


def format_detections(det, batch_hw, img_original_shape, scaling_shape, detection_threshold,
                      use_model_native_classes=False, modern=False, letterbox_pad=None):
    """
    det: (k,6) float32 [x1,y1,x2,y2,conf,cls] in letterboxed pixels, confidence-descending
    (what mdhip_nms returns).  Returns (detections, max_conf) exactly as the reference builds
    them: ascending confidence order (it iterates `reversed(det)`), truncated values.
    modern=True (reference :1369-1381,:1396-1397): `img_original_shape` is the shape of the RESIZED image,
    scale_coords gets ratio_pad = ((resized / original per axis), letterbox_pad) -- gain = the first ratio --
    and coordinates / confidences are rounded instead of truncated.
    """
    det = np.asarray(det, dtype=_F)
    k = det.shape[0]
    if k == 0:
        return [], 0.0
    h1, w1 = int(batch_hw[0]), int(batch_hw[1])
    h0, w0 = int(img_original_shape[0]), int(img_original_shape[1])
    if modern:
        gain = h0 / int(scaling_shape[0])                   # ratio_pad[0][0]
        pad = (letterbox_pad[0], letterbox_pad[1])
        h0, w0 = int(scaling_shape[0]), int(scaling_shape[1])   # clip_coords against scaling_shape
    else:
        # scale_coords (ratio_pad=None) -- gain/pad are Python floats, tensor math is fp32
        gain = min(h1 / h0, w1 / w0)
        pad = ((w1 - w0 * gain) / 2, (h1 - h0 * gain) / 2)
    xyxy = det[:, :4].copy()
    xyxy[:, [0, 2]] -= _F(pad[0])
    xyxy[:, [1, 3]] -= _F(pad[1])
    xyxy /= _F(gain)
    xyxy[:, [0, 2]] = np.clip(xyxy[:, [0, 2]], _F(0), _F(w0))
    xyxy[:, [1, 3]] = np.clip(xyxy[:, [1, 3]], _F(0), _F(h0))
    xyxy = np.rint(xyxy)                                    # torch .round(): half to even

    conf32 = det[:, 4]
    keep = ~(conf32 < _F(detection_threshold))              # `if conf < threshold: continue`
    order = np.arange(k - 1, -1, -1)                        # reversed(det)
    order = order[keep[order]]
    if order.size == 0:
        return [], 0.0

    # xyxy2xywh in fp32, then / gn (int64 tensor -> fp32 division)
    gn = np.array([scaling_shape[1], scaling_shape[0], scaling_shape[1], scaling_shape[0]], dtype=_F)
    x1, y1, x2, y2 = xyxy[:, 0], xyxy[:, 1], xyxy[:, 2], xyxy[:, 3]
    xywh = np.stack([(x1 + x2) / _F(2), (y1 + y2) / _F(2), x2 - x1, y2 - y1], axis=1).astype(_F)
    xywh = (xywh / gn).astype(np.float64)                   # .tolist() -> Python floats
    # convert_yolo_to_xywh in float64
    api = np.stack([xywh[:, 0] - xywh[:, 2] / 2.0, xywh[:, 1] - xywh[:, 3] / 2.0,
                    xywh[:, 2], xywh[:, 3]], axis=1)
    if modern:                                              # ct_utils.round_float = Python round()
        api = np.array([[round(float(v), COORD_DIGITS) for v in row] for row in api], dtype=np.float64).reshape(-1, 4)
        conf = np.array([round(float(v), CONF_DIGITS) for v in conf32.astype(np.float64)], dtype=np.float64)
    else:
        api = np.floor(api * (10 ** COORD_DIGITS)) / (10 ** COORD_DIGITS)
        conf = np.floor(conf32.astype(np.float64) * (10 ** CONF_DIGITS)) / (10 ** CONF_DIGITS)

    cls = det[:, 5].astype(np.int64)
    if not use_model_native_classes:
        cls = cls + 1
        bad = cls[order][(cls[order] < 1) | (cls[order] > 3)]
        if bad.size:
            raise KeyError('{} is not a valid class.'.format(int(bad[0])))
    # bulk conversion to Python objects (one .tolist() per array) instead of per-element float()/str()
    cats = [str(c) for c in cls[order].tolist()]
    confs = conf[order].tolist()
    boxes = api[order].tolist()
    detections = [{'category': c, 'conf': f, 'bbox': b} for c, f, b in zip(cats, confs, boxes)]
    max_conf = float(max(0.0, max(confs)))
    return detections, max_conf

