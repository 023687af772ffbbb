"""
YOLOv5 model descriptions (pure data) for the MegaDetector v5 family.

The MDv5a/b checkpoints are YOLOv5x6 (P6, 1280 px) networks: the reference never
states the topology in-tree, it unpickles the nn.Module out of md_v5a.0.0.pt
(reference: megadetector/detection/pytorch_detector.py:929,957) and relies on the
third-party package ultralytics-yolov5==0.1.1 (reference: pyproject.toml:70) for
models/hub/yolov5x6.yaml.  What follows is that published model description
restated as Python data (SURVEY.md section 8(a), P4 layer table).

Each row is [from, number, module, args] exactly as in the YOLOv5 yaml format.
"""

# Default P6 anchors in pixels (yolov5 hub/yolov5*6.yaml); real checkpoints carry
# their own (possibly auto-anchored) values which always take precedence.
ANCHORS_P6 = [
    [19, 27, 44, 40, 38, 94],        # P3/8
    [96, 68, 86, 152, 180, 137],     # P4/16
    [140, 301, 303, 264, 238, 542],  # P5/32
    [436, 615, 739, 380, 925, 792],  # P6/64
]

ANCHORS_P5 = [
    [10, 13, 16, 30, 33, 23],        # P3/8
    [30, 61, 62, 45, 59, 119],       # P4/16
    [116, 90, 156, 198, 373, 326],   # P5/32
]

_BACKBONE_P6 = [
    [-1, 1, 'Conv', [64, 6, 2, 2]],    # 0-P1/2
    [-1, 1, 'Conv', [128, 3, 2]],      # 1-P2/4
    [-1, 3, 'C3', [128]],
    [-1, 1, 'Conv', [256, 3, 2]],      # 3-P3/8
    [-1, 6, 'C3', [256]],
    [-1, 1, 'Conv', [512, 3, 2]],      # 5-P4/16
    [-1, 9, 'C3', [512]],
    [-1, 1, 'Conv', [768, 3, 2]],      # 7-P5/32
    [-1, 3, 'C3', [768]],
    [-1, 1, 'Conv', [1024, 3, 2]],     # 9-P6/64
    [-1, 3, 'C3', [1024]],
    [-1, 1, 'SPPF', [1024, 5]],        # 11
]

_HEAD_P6 = [
    [-1, 1, 'Conv', [768, 1, 1]],
    [-1, 1, 'nn.Upsample', [None, 2, 'nearest']],
    [[-1, 8], 1, 'Concat', [1]],       # cat backbone P5
    [-1, 3, 'C3', [768, False]],       # 15
    [-1, 1, 'Conv', [512, 1, 1]],
    [-1, 1, 'nn.Upsample', [None, 2, 'nearest']],
    [[-1, 6], 1, 'Concat', [1]],       # cat backbone P4
    [-1, 3, 'C3', [512, False]],       # 19
    [-1, 1, 'Conv', [256, 1, 1]],
    [-1, 1, 'nn.Upsample', [None, 2, 'nearest']],
    [[-1, 4], 1, 'Concat', [1]],       # cat backbone P3
    [-1, 3, 'C3', [256, False]],       # 23 (P3/8-small)
    [-1, 1, 'Conv', [256, 3, 2]],
    [[-1, 20], 1, 'Concat', [1]],      # cat head P4
    [-1, 3, 'C3', [512, False]],       # 26 (P4/16-medium)
    [-1, 1, 'Conv', [512, 3, 2]],
    [[-1, 16], 1, 'Concat', [1]],      # cat head P5
    [-1, 3, 'C3', [768, False]],       # 29 (P5/32-large)
    [-1, 1, 'Conv', [768, 3, 2]],
    [[-1, 12], 1, 'Concat', [1]],      # cat head P6
    [-1, 3, 'C3', [1024, False]],      # 32 (P6/64-xlarge)
    [[23, 26, 29, 32], 1, 'Detect', ['nc', 'anchors']],
]

_BACKBONE_P5 = [
    [-1, 1, 'Conv', [64, 6, 2, 2]],    # 0-P1/2
    [-1, 1, 'Conv', [128, 3, 2]],      # 1-P2/4
    [-1, 3, 'C3', [128]],
    [-1, 1, 'Conv', [256, 3, 2]],      # 3-P3/8
    [-1, 6, 'C3', [256]],
    [-1, 1, 'Conv', [512, 3, 2]],      # 5-P4/16
    [-1, 9, 'C3', [512]],
    [-1, 1, 'Conv', [1024, 3, 2]],     # 7-P5/32
    [-1, 3, 'C3', [1024]],
    [-1, 1, 'SPPF', [1024, 5]],        # 9
]

_HEAD_P5 = [
    [-1, 1, 'Conv', [512, 1, 1]],
    [-1, 1, 'nn.Upsample', [None, 2, 'nearest']],
    [[-1, 6], 1, 'Concat', [1]],
    [-1, 3, 'C3', [512, False]],       # 13
    [-1, 1, 'Conv', [256, 1, 1]],
    [-1, 1, 'nn.Upsample', [None, 2, 'nearest']],
    [[-1, 4], 1, 'Concat', [1]],
    [-1, 3, 'C3', [256, False]],       # 17 (P3/8-small)
    [-1, 1, 'Conv', [256, 3, 2]],
    [[-1, 14], 1, 'Concat', [1]],
    [-1, 3, 'C3', [512, False]],       # 20 (P4/16-medium)
    [-1, 1, 'Conv', [512, 3, 2]],
    [[-1, 10], 1, 'Concat', [1]],
    [-1, 3, 'C3', [1024, False]],      # 23 (P5/32-large)
    [[17, 20, 23], 1, 'Detect', ['nc', 'anchors']],
]


def make_yaml(depth_multiple, width_multiple, nc=3, p6=True, anchors=None):
    """Build a YOLOv5 yaml dict (same keys as the dict pickled as model.yaml)."""
    return {
        'nc': nc,
        'depth_multiple': depth_multiple,
        'width_multiple': width_multiple,
        'anchors': [list(a) for a in (anchors or (ANCHORS_P6 if p6 else ANCHORS_P5))],
        'backbone': [list(r) for r in (_BACKBONE_P6 if p6 else _BACKBONE_P5)],
        'head': [list(r) for r in (_HEAD_P6 if p6 else _HEAD_P5)],
    }


#: MDv5a / MDv5b / MDv1000-redwood: YOLOv5x6, nc=3 (reference: run_detector.py:177-223)
YOLOV5X6_MD = make_yaml(1.33, 1.25, nc=3, p6=True)

#: upstream COCO model, used only to cross-check FLOP/param counts against the
#: figures the reference cites (docs/release-notes/mdv1000-release.md:279)
YOLOV5X6_COCO = make_yaml(1.33, 1.25, nc=80, p6=True)

#: MDv1000-spruce: YOLOv5s (P5, 3 heads) (reference: pytorch_detector.py:827,842)
YOLOV5S_MD = make_yaml(0.33, 0.50, nc=3, p6=False)

#: small P6 network for fast tests (same module mix as x6, ~1/60 of the FLOPs)
YOLOV5N6_TEST = make_yaml(0.33, 0.25, nc=3, p6=True)

#: wider small P6 network (hidden widths 64..256 at strides 8 and 16): exercises the kernels that need
#: at least 64 input channels (row-patch 3x3) in the tests
YOLOV5S6_TEST = make_yaml(0.33, 0.50, nc=3, p6=True)

#: small P5 (3 heads, stride 32) network with 5 classes: the non-P6 family members (MDv1000-spruce is a
#: YOLOv5s) and a class count other than 3, in the tests
YOLOV5N_P5_TEST = make_yaml(0.33, 0.25, nc=5, p6=False)
