"""
ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of everything on the MDv5 batch-inference hot path that is *not* the conv
stack: letterbox preprocessing, tensor prep, NMS, box rescaling and output formatting.
Each function cites the reference lines it follows; third-party pieces (cv2.resize,
yolov5 letterbox / scale_coords / xyxy2xywh, torchvision.ops.nms) are restated from their
published algorithms and anchored on the reference's call sites.

Pinned (tests/test_oracle_*.py) against the golden vectors the reference tree holds:
  * megadetector/tests/test_nms_synthetic.py (inputs :83-117,:309-310; expectations
    :188-208,:247,:263,:295-303,:322-331)                          -> nms()
  * megadetector/utils/ct_utils.py:1332-1346,:1467-1493            -> truncate/round/yolo->xywh/IoU
PARITY UNPINNED: cv2.resize bit-exactness (OpenCV is not installed here and the reference
holds no resize fixtures); the fixed-point bilinear below follows OpenCV's imgproc/resize.cpp
(HResizeLinear / VResizeLinear<uchar,int,short>, INTER_RESIZE_COEF_BITS=11).  Second source (not a pin):
tests/test_oracle_resize_second_source.py holds both resize restatements against torch's float bilinear
interpolation / block means (within one grey level everywhere).
"""

import math

import numpy as np
import torch

# reference: megadetector/detection/run_detector.py:55-60
FAILURE_INFER = 'inference failure'
FAILURE_IMAGE_OPEN = 'image access failure'
CONF_DIGITS = 3
COORD_DIGITS = 4


# --------------------------------------------------------------------------------------
# ct_utils restatements (reference: megadetector/utils/ct_utils.py)
# --------------------------------------------------------------------------------------

def truncate_float(x, precision=3):
    # ct_utils.py:82-103
    return math.floor(x * (10 ** precision)) / (10 ** precision)


def truncate_float_array(xs, precision=3):
    # ct_utils.py:35-48
    return [truncate_float(x, precision=precision) for x in xs]


def round_float(x, precision=3):
    # ct_utils.py:67-79
    return round(x, precision)


def round_float_array(xs, precision=3):
    # ct_utils.py:51-64
    return [round_float(x, precision) for x in xs]


def convert_yolo_to_xywh(yolo_box):
    # ct_utils.py:255-270
    x_center, y_center, width_of_box, height_of_box = yolo_box
    x_min = x_center - width_of_box / 2.0
    y_min = y_center - height_of_box / 2.0
    return [x_min, y_min, width_of_box, height_of_box]


def get_iou(bb1, bb2):
    # ct_utils.py:291-340 ([x_min,y_min,w,h] boxes)
    a = [bb1[0], bb1[1], bb1[0] + bb1[2], bb1[1] + bb1[3]]
    b = [bb2[0], bb2[1], bb2[0] + bb2[2], bb2[1] + bb2[3]]
    assert a[0] < a[2] and a[1] < a[3] and b[0] < b[2] and b[1] < b[3]
    x_left, y_top = max(a[0], b[0]), max(a[1], b[1])
    x_right, y_bottom = min(a[2], b[2]), min(a[3], b[3])
    if x_right < x_left or y_bottom < y_top:
        return 0.0
    inter = (x_right - x_left) * (y_bottom - y_top)
    area_a = (a[2] - a[0]) * (a[3] - a[1])
    area_b = (b[2] - b[0]) * (b[3] - b[1])
    return inter / float(area_a + area_b - inter)


# --------------------------------------------------------------------------------------
# cv2.resize(INTER_LINEAR) for 8-bit images, restated  [3P: OpenCV imgproc/resize.cpp]
# --------------------------------------------------------------------------------------

def _linear_coeffs(dst_len, src_len):
    """Per destination index: source index pair base and the two 11-bit weights."""
    scale = 1.0 / (float(dst_len) / float(src_len))          # double, as in cv::resize
    d = np.arange(dst_len, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)          # fx = (float)((dx+0.5)*scale_x - 0.5)
    s = np.floor(f).astype(np.int64)                          # cvFloor
    f = f - s.astype(np.float32)
    lo = s < 0
    f[lo] = 0.0
    s[lo] = 0
    hi = s >= src_len - 1
    f[hi] = 0.0
    s[hi] = src_len - 1
    # saturate_cast<short>(x * INTER_RESIZE_COEF_SCALE) == cvRound (round-half-even)
    w1 = np.rint(f.astype(np.float32) * np.float32(2048.0)).astype(np.int64)
    w0 = np.rint((np.float32(1.0) - f.astype(np.float32)) * np.float32(2048.0)).astype(np.int64)
    s1 = np.minimum(s + 1, src_len - 1)
    return s, s1, w0, w1


def resize_linear_u8(img, dst_w, dst_h):
    """cv2.resize(img, (dst_w, dst_h), interpolation=cv2.INTER_LINEAR) for HxWxC uint8."""
    assert img.dtype == np.uint8 and img.ndim == 3
    src_h, src_w = img.shape[:2]
    if (src_h, src_w) == (dst_h, dst_w):
        return img.copy()
    x0, x1, a0, a1 = _linear_coeffs(dst_w, src_w)
    y0, y1, b0, b1 = _linear_coeffs(dst_h, src_h)
    src = img.astype(np.int64)
    # horizontal pass (HResizeLinear<uchar,int,short>): int rows scaled by 2^11
    rows = src[:, x0, :] * a0[None, :, None] + src[:, x1, :] * a1[None, :, None]
    r0 = rows[y0]
    r1 = rows[y1]
    # vertical pass (VResizeLinear<uchar,int,short>, FixedPtCast<int,uchar,22>)
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def area_tab(ssize, dsize):
    """
    cv::computeResizeAreaTab (OpenCV imgproc/resize.cpp) for one axis: per destination index the list of
    (source index, float32 weight) in the order the kernel accumulates them.  scale is the double
    1 / (dsize / ssize) of cv::resize.
    """
    scale = 1.0 / (float(dsize) / float(ssize))
    tab = []
    for d in range(dsize):
        fsx1 = d * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = int(math.ceil(fsx1)), int(math.floor(fsx2))
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        ent = []
        if sx1 - fsx1 > 1e-3:
            ent.append((sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            ent.append((sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            ent.append((sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
        tab.append(ent)
    return tab, scale


def resize_area_u8(img, dst_w, dst_h):
    """
    cv2.resize(img, (dst_w, dst_h), interpolation=cv2.INTER_AREA) for HxWx3 uint8 when shrinking in both
    directions (the only case the reference uses it for, pytorch_detector.py:1048-1062).  Integer scale
    factors take OpenCV's integer path (resizeAreaFast_: block sum, 2x2 as (s+2)>>2, else
    saturate_cast<uchar>(sum * (1.f/area))); everything else resizeArea_<uchar,float>: per source row
    buf[dx] = sum_k S[sx_k]*alpha_k (fp32, in table order), then sum[dx] = beta0*buf0, += beta_j*buf_j,
    output saturate_cast<uchar>(sum) = round-half-even.  PARITY UNPINNED (no OpenCV offline).
    """
    assert img.dtype == np.uint8 and img.ndim == 3
    src_h, src_w = img.shape[:2]
    assert dst_w <= src_w and dst_h <= src_h
    xtab, sx = area_tab(src_w, dst_w)
    ytab, sy = area_tab(src_h, dst_h)
    isx, isy = int(np.rint(sx)), int(np.rint(sy))
    if abs(sx - isx) < np.finfo(np.float64).eps and abs(sy - isy) < np.finfo(np.float64).eps:
        blocks = img[:dst_h * isy, :dst_w * isx].astype(np.int64).reshape(dst_h, isy, dst_w, isx, 3).sum(axis=(1, 3))
        if isx == 2 and isy == 2:
            return ((blocks + 2) >> 2).astype(np.uint8)
        scale = np.float32(1.0) / np.float32(isx * isy)
        return np.clip(np.rint(blocks.astype(np.float32) * scale), 0, 255).astype(np.uint8)
    src = img.astype(np.float32)
    buf = np.zeros((src_h, dst_w, 3), dtype=np.float32)
    for dx, ent in enumerate(xtab):
        acc = np.zeros((src_h, 3), dtype=np.float32)
        for sxk, a in ent:
            acc = acc + src[:, sxk, :] * a
        buf[:, dx, :] = acc
    out = np.zeros((dst_h, dst_w, 3), dtype=np.float32)
    for dy, ent in enumerate(ytab):
        first = True
        for syk, b in ent:
            if first:
                acc = b * buf[syk]
                first = False
            else:
                acc = acc + b * buf[syk]
        out[dy] = acc
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


# --------------------------------------------------------------------------------------
# letterbox  [3P: yolov5 utils/augmentations.py; ratio/pad arithmetic restated in-tree at
# reference pytorch_detector.py:434-454]
# --------------------------------------------------------------------------------------

def letterbox_geometry(shape_hw, new_shape=1280, stride=64, auto=True, scaleup=True):
    """
    Returns dict(ratio, pad=(dw,dh), new_unpad=(w,h), top, bottom, left, right, out_hw).
    """
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    h, w = shape_hw
    r = min(new_shape[0] / h, new_shape[1] / w)
    if not scaleup:
        r = min(r, 1.0)
    new_unpad = int(round(w * r)), int(round(h * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = int(np.mod(dw, stride)), int(np.mod(dh, stride))
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return dict(ratio=(r, r), pad=(dw, dh), new_unpad=new_unpad, top=top, bottom=bottom,
                left=left, right=right,
                out_hw=(new_unpad[1] + top + bottom, new_unpad[0] + left + right))


def letterbox(img, new_shape=1280, stride=64, auto=True, scaleup=True, color=(114, 114, 114)):
    """yolov5 letterbox(): resize (INTER_LINEAR) + constant border.  Returns (img, ratio, pad)."""
    g = letterbox_geometry(img.shape[:2], new_shape, stride, auto, scaleup)
    if (img.shape[1], img.shape[0]) != g['new_unpad']:
        img = resize_linear_u8(img, g['new_unpad'][0], g['new_unpad'][1])
    out = np.empty((g['out_hw'][0], g['out_hw'][1], 3), dtype=np.uint8)
    out[:] = np.array(color, dtype=np.uint8)
    out[g['top']:g['top'] + img.shape[0], g['left']:g['left'] + img.shape[1]] = img
    return out, g['ratio'], g['pad']


def preprocess_image_classic(img_original, image_size=1280, stride=64):
    """
    reference pytorch_detector.py:964-1119, compatibility_mode 'classic' (the default, :733):
    no pre-resize; letterbox(auto=True, scaleup=True).
    """
    img_original = np.asarray(img_original)
    img, ratio, pad = letterbox(img_original, new_shape=image_size, stride=stride,
                                auto=True, scaleup=True)
    return dict(img_processed=img, img_original=img_original, target_shape=image_size,
                scaling_shape=img_original.shape, letterbox_ratio=ratio, letterbox_pad=pad)


def modern_geometry(shape_hw, image_size=1280, stride=64, use_ceil=False):
    """
    reference pytorch_detector.py:1036-1109, compatibility_mode 'modern': resized shape (long side -> image_size;
    int() or ceil()), interpolation ('linear' when growing, 'area' when shrinking, None when the ratio is 1),
    target shape ceil(normalised * image_size / stride + 0.5) * stride, and the letterbox of the RESIZED image
    into it (auto=False, scaleup=False: padding only).
    """
    h, w = int(shape_hw[0]), int(shape_hw[1])
    ratio = image_size / max(h, w)
    rh, rw, interp = h, w, None
    if ratio != 1:
        interp = 'linear' if ratio > 1 else 'area'
        rw = math.ceil(w * ratio) if use_ceil else int(w * ratio)
        rh = math.ceil(h * ratio) if use_ceil else int(h * ratio)
    md = max(rh, rw, 3)                    # max(img_original.shape) includes the channel count
    norm = np.array([rh / md, rw / md])
    target = (np.ceil((norm * image_size) / stride + 0.5).astype(int) * stride)
    g = letterbox_geometry((rh, rw), new_shape=(int(target[0]), int(target[1])), stride=stride, auto=False, scaleup=False)
    return dict(resized_hw=(rh, rw), interp=interp, target_shape=(int(target[0]), int(target[1])), letterbox=g)


def preprocess_image_modern(img_original, image_size=1280, stride=64, use_ceil=False):
    """reference pytorch_detector.py:964-1119 with a compatibility_mode other than 'classic'."""
    img_original = np.asarray(img_original)
    scaling_shape = img_original.shape
    m = modern_geometry(img_original.shape[:2], image_size, stride, use_ceil)
    rh, rw = m['resized_hw']
    if m['interp'] == 'linear':
        img_resized = resize_linear_u8(img_original, rw, rh)
    elif m['interp'] == 'area':
        img_resized = resize_area_u8(img_original, rw, rh)
    else:
        img_resized = img_original
    img, ratio, pad = letterbox(img_resized, new_shape=m['target_shape'], stride=stride, auto=False, scaleup=False)
    return dict(img_processed=img, img_original=img_resized, target_shape=m['target_shape'],
                scaling_shape=scaling_shape, letterbox_ratio=ratio, letterbox_pad=pad)


def to_batch_tensor(imgs_processed):
    """reference pytorch_detector.py:1283-1306: HWC u8 -> NCHW fp32 / 255."""
    ts = [torch.from_numpy(np.ascontiguousarray(im.transpose((2, 0, 1)))) for im in imgs_processed]
    t = torch.stack(ts).float()
    t /= 255.0
    return t


# --------------------------------------------------------------------------------------
# NMS (reference pytorch_detector.py:502-610; inner greedy step = torchvision.ops.nms [3P])
# --------------------------------------------------------------------------------------

def _greedy_nms(boxes, scores, iou_thres):
    """
    torchvision.ops.nms restated (torchvision/csrc/ops/cpu/nms_kernel.cpp): process boxes in
    order of decreasing score; a box is suppressed when IoU with an already-kept box is
    strictly greater than iou_thres; area = (x2-x1)*(y2-y1).  Ties in score are taken in
    order of increasing input index (stable sort) -- torchvision leaves tie order
    unspecified; this is the order the HIP path implements.
    """
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros((0,), dtype=torch.int64)
    order = torch.sort(scores, descending=True, stable=True).indices
    b = boxes[order].numpy().astype(np.float32)
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr = np.float32(iou_thres)
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        if i + 1 >= n:
            break
        xx1 = np.maximum(x1[i], x1[i + 1:])
        yy1 = np.maximum(y1[i], y1[i + 1:])
        xx2 = np.minimum(x2[i], x2[i + 1:])
        yy2 = np.minimum(y2[i], y2[i + 1:])
        w = np.maximum(np.float32(0), xx2 - xx1)
        h = np.maximum(np.float32(0), yy2 - yy1)
        inter = w * h
        with np.errstate(divide='ignore', invalid='ignore'):
            ovr = inter / (areas[i] + areas[i + 1:] - inter)
        suppressed[i + 1:] |= (ovr > thr)
    return order[torch.tensor(keep, dtype=torch.int64)]


def nms(prediction, conf_thres=0.25, iou_thres=0.45, max_det=300):
    """
    reference pytorch_detector.py:502-610, statement for statement, with torchvision.ops.nms
    replaced by _greedy_nms.  The reference's final cross-class `argsort(descending=True)`
    (:597) leaves the order of equal confidences unspecified; here (and in the HIP path) ties
    are resolved by increasing anchor index.
    prediction: (B, n_anchors, 5+nc) fp32.  Returns list of (n_i, 6) fp32 tensors
    [x1,y1,x2,y2,conf,cls].
    """
    output = []
    for img_idx in range(prediction.shape[0]):
        x = prediction[img_idx]
        anchor_idx = torch.arange(x.shape[0])
        valid = x[:, 4] > conf_thres
        x = x[valid]
        anchor_idx = anchor_idx[valid]
        if x.shape[0] == 0:
            output.append(torch.zeros((0, 6)))
            continue
        box = x[:, :4].clone()
        box[:, 0] = x[:, 0] - x[:, 2] / 2.0
        box[:, 1] = x[:, 1] - x[:, 3] / 2.0
        box[:, 2] = x[:, 0] + x[:, 2] / 2.0
        box[:, 3] = x[:, 1] + x[:, 3] / 2.0
        class_conf = x[:, 5:] * x[:, 4:5]
        best_class_conf, best_class_idx = class_conf.max(1, keepdim=True)
        conf_mask = best_class_conf.view(-1) > conf_thres
        if conf_mask.sum() == 0:
            output.append(torch.zeros((0, 6)))
            continue
        box = box[conf_mask]
        best_class_conf = best_class_conf[conf_mask]
        best_class_idx = best_class_idx[conf_mask]
        anchor_idx = anchor_idx[conf_mask]
        final = []
        final_idx = []
        for class_id in best_class_idx.unique():
            class_mask = (best_class_idx == class_id).view(-1)
            class_boxes = box[class_mask]
            class_scores = best_class_conf[class_mask].view(-1)
            keep = _greedy_nms(class_boxes, class_scores, iou_thres)
            if len(keep) > 0:
                kept_classes = torch.full((len(keep), 1), float(class_id.item()))
                final.append(torch.cat([class_boxes[keep], class_scores[keep].unsqueeze(1),
                                        kept_classes], 1))
                final_idx.append(anchor_idx[class_mask][keep])
        if final:
            det = torch.cat(final, 0)
            idx = torch.cat(final_idx, 0)
            by_idx = torch.sort(idx, stable=True).indices
            det = det[by_idx]
            det = det[torch.sort(det[:, 4], descending=True, stable=True).indices]
            output.append(det[:max_det])
        else:
            output.append(torch.zeros((0, 6)))
    return output


# --------------------------------------------------------------------------------------
# box rescale + formatting (reference pytorch_detector.py:1352-1422)
# --------------------------------------------------------------------------------------

def scale_coords(img1_shape, coords, img0_shape):
    """[3P] yolov5 utils/general.py:scale_coords (ratio_pad=None) + clip_coords."""
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    coords[:, [0, 2]] -= pad[0]
    coords[:, [1, 3]] -= pad[1]
    coords[:, :4] /= gain
    coords[:, 0].clamp_(0, img0_shape[1])
    coords[:, 1].clamp_(0, img0_shape[0])
    coords[:, 2].clamp_(0, img0_shape[1])
    coords[:, 3].clamp_(0, img0_shape[0])
    return coords


def xyxy2xywh(x):
    """[3P] yolov5 utils/general.py:xyxy2xywh."""
    y = x.clone()
    y[:, 0] = (x[:, 0] + x[:, 2]) / 2
    y[:, 1] = (x[:, 1] + x[:, 3]) / 2
    y[:, 2] = x[:, 2] - x[:, 0]
    y[:, 3] = x[:, 3] - x[:, 1]
    return y


def scale_coords_ratio_pad(coords, img0_shape, ratio_pad):
    """[3P] yolov5 utils/general.py:scale_coords with ratio_pad given: gain = ratio_pad[0][0], pad = ratio_pad[1]."""
    gain = ratio_pad[0][0]
    pad = ratio_pad[1]
    coords[:, [0, 2]] -= pad[0]
    coords[:, [1, 3]] -= pad[1]
    coords[:, :4] /= gain
    coords[:, 0].clamp_(0, img0_shape[1])
    coords[:, 1].clamp_(0, img0_shape[0])
    coords[:, 2].clamp_(0, img0_shape[1])
    coords[:, 3].clamp_(0, img0_shape[0])
    return coords


def format_detections(det, batch_hw, img_original_shape, scaling_shape, detection_threshold,
                      use_model_native_classes=False, modern=False, letterbox_pad=None):
    """
    reference pytorch_detector.py:1361-1422.  det: (n,6) fp32 tensor from nms().  'classic' (default):
    scale_coords against the image shape, truncation; modern: ratio_pad = ((resized/original per axis),
    letterbox_pad), rounding instead of truncation.  Returns (detections list, max_conf).
    """
    detections = []
    max_conf = 0.0
    if len(det) > 0:
        det = det.clone()
        gn = torch.tensor(scaling_shape)[[1, 0, 1, 0]]
        if modern:
            ratio = (img_original_shape[0] / scaling_shape[0], img_original_shape[1] / scaling_shape[1])
            det[:, :4] = scale_coords_ratio_pad(det[:, :4], scaling_shape, (ratio, letterbox_pad)).round()
        else:
            det[:, :4] = scale_coords(batch_hw, det[:, :4], img_original_shape).round()
        for *xyxy, conf, cls in reversed(det):
            if conf < detection_threshold:
                continue
            xywh = (xyxy2xywh(torch.tensor(xyxy).view(1, 4)) / gn).view(-1).tolist()
            api_box = convert_yolo_to_xywh(xywh)
            if modern:
                api_box = round_float_array(api_box, precision=COORD_DIGITS)
                conf = round_float(conf.tolist(), precision=CONF_DIGITS)
            else:
                api_box = truncate_float_array(api_box, precision=COORD_DIGITS)
                conf = truncate_float(conf.tolist(), precision=CONF_DIGITS)
            if not use_model_native_classes:
                cls = int(cls.tolist()) + 1
                if cls not in (1, 2, 3):
                    raise KeyError('{} is not a valid class.'.format(cls))
            else:
                cls = int(cls.tolist())
            detections.append({'category': str(cls), 'conf': conf, 'bbox': api_box})
            max_conf = max(max_conf, conf)
    return detections, max_conf


# --------------------------------------------------------------------------------------
# the reference's definition of "same results" (reference md_tests.py:96-100,124,418-531)
# --------------------------------------------------------------------------------------

def compare_detection_lists(dets_a, dets_b, iou_match=0.85, bidirectional=True):
    """
    Returns (max_conf_err, max_coord_err) between two detection lists, following reference md_tests.py:418-531
    statement for statement: every detection of A is matched to the same-category detection of B with the highest
    IoU >= iou_match (matches may be many-to-one); |d conf| and max |d coord| over the matches; an unmatched detection
    contributes its own confidence as confidence error; then the same with the arguments reversed.
    Pinned against the real function by tests/golden/compare_kat.json (tests/test_oracle_golden.py).
    """
    max_conf_err, max_coord_err = 0, 0
    for da in dets_a:
        best, best_iou = None, -1
        for db in dets_b:
            if db['category'] != da['category']:
                continue
            iou = get_iou(da['bbox'], db['bbox'])
            if iou >= iou_match and iou > best_iou:
                best, best_iou = db, iou
        if best is None:
            if da['conf'] > max_conf_err:
                max_conf_err = da['conf']
            continue
        max_conf_err = max(max_conf_err, abs(da['conf'] - best['conf']))
        max_coord_err = max(max_coord_err, max(abs(p - q) for p, q in zip(da['bbox'], best['bbox'])))
    if bidirectional:
        rc, rx = compare_detection_lists(dets_b, dets_a, iou_match, bidirectional=False)
        max_conf_err, max_coord_err = max(max_conf_err, rc), max(max_coord_err, rx)
    return max_conf_err, max_coord_err
