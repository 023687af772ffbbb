"""
Weights for the HIP detector: either lifted out of a real MegaDetector / YOLOv5 checkpoint
without importing the yolov5 package, or seeded synthetic weights on the same topology.

Checkpoint path mirrors reference megadetector/detection/pytorch_detector.py:913-959
(_load_model: torch.load(weights_only=False) -> checkpoint['model'].float().fuse().eval());
the Conv+BatchNorm folding that `.fuse()` performs in the third-party yolov5 package
(utils/torch_utils.py:fuse_conv_and_bn) is done here in fp32 on the host.
"""

import io
import pickle
import sys
import types
import zipfile

import numpy as np

from . import yolo_yaml
from .yolo_model import YoloWeights, resolve_yaml, MDHIP_DETECT


# --------------------------------------------------------------------------------------
# seeded synthetic weights (no checkpoint available: there is no network in the build or
# bench environment; timing does not depend on the values, detections do)
# --------------------------------------------------------------------------------------

def synthetic_weights(yaml=None, seed=0, gain=1.75, res_gain=0.6, bias_std=0.1,
                      detect_gain=22.0, detect_obj_bias=-12.5):
    """
    Deterministic (numpy PCG64, platform independent) pseudo-trained weights.

    Conv weights are N(0, gain^2 / fan_in) with zero mean per output channel (what a folded
    BatchNorm achieves: no mean drift through the ~100 SiLU layers); the residual-branch 3x3 of
    every shortcut bottleneck is scaled by res_gain so the backbone does not blow up.  With
    gain below the critical value (~1.8) activations settle at std ~0.1-0.3.  The Detect biases
    make objectness rare, as in a trained detector: roughly one anchor in six clears the 1e-5
    batch-mode threshold (reference run_detector_batch.py:751 / pytorch_detector.py:1127) and
    a handful clear 0.005, so NMS sees a realistic candidate load.
    """
    from .yolo_model import model_strides
    yaml = yaml or yolo_yaml.YOLOV5X6_MD
    specs = resolve_yaml(yaml)
    rng = np.random.Generator(np.random.PCG64(seed))
    w = {}
    nc = yaml['nc']
    for s in specs:
        if s.type == MDHIP_DETECT:
            na = len(yaml['anchors'][0]) // 2
            strides = model_strides(specs)
            for l, name in enumerate(s.conv_names):
                c_in = _channels_of(specs, s.frm[l])
                wt = rng.standard_normal((na * (nc + 5), c_in, 1, 1), dtype=np.float32)
                wt -= wt.mean(axis=1, keepdims=True)
                wt *= np.float32(detect_gain / np.sqrt(c_in))
                # box logits stay small: (2*sigmoid)^2 * anchor then gives boxes of 0.3x..2x the
                # anchor instead of degenerate slivers
                wt.reshape(na, nc + 5, c_in)[:, :4, :] *= np.float32(0.2)
                b = np.zeros((na, nc + 5), dtype=np.float32)
                b[:, 4] = detect_obj_bias
                b[:, 5:] = rng.standard_normal((na, nc), dtype=np.float32) * np.float32(0.5)
                w[name + '.weight'] = wt
                w[name + '.bias'] = b.reshape(-1)
            anchors = np.asarray(yaml['anchors'], dtype=np.float32).reshape(len(strides), -1, 2)
            anchors = anchors / np.asarray(strides, dtype=np.float32).reshape(-1, 1, 1)
            w['model.{}.anchors'.format(s.index)] = anchors.astype(np.float32)
            continue
        shapes = _conv_shapes(s)
        for name, (c2, c1, k) in zip(s.conv_names, shapes):
            wt = rng.standard_normal((c2, c1, k, k), dtype=np.float32)
            wt -= wt.mean(axis=(1, 2, 3), keepdims=True)
            g = gain
            if s.shortcut and '.m.' in name and name.endswith('.cv2.conv'):
                g = res_gain
            wt *= np.float32(g / np.sqrt(c1 * k * k))
            w[name + '.weight'] = wt
            w[name + '.bias'] = rng.standard_normal(c2, dtype=np.float32) * np.float32(bias_std)
    return YoloWeights(yaml, w, source='synthetic(seed={})'.format(seed))


def _channels_of(specs, idx):
    return specs[idx].c_out


def _conv_shapes(s):
    from .yolo_model import MDHIP_CONV, MDHIP_C3, MDHIP_SPPF
    if s.type == MDHIP_CONV:
        return [(s.c_out, s.c_in, s.k)]
    if s.type == MDHIP_C3:
        h = s.hidden
        shapes = [(h, s.c_in, 1), (h, s.c_in, 1), (s.c_out, 2 * h, 1)]
        for _ in range(s.n):
            shapes += [(h, h, 1), (h, h, 3)]
        return shapes
    if s.type == MDHIP_SPPF:
        h = s.hidden
        return [(h, s.c_in, 1), (s.c_out, 4 * h, 1)]
    return []


# --------------------------------------------------------------------------------------
# real checkpoints
# --------------------------------------------------------------------------------------

class _StubModule:
    """Stand-in for any class of the yolov5 package (models.*, utils.*) found in the pickle."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):
            if isinstance(state[0], dict):
                self.__dict__.update(state[0])
            self.__dict__.update(state[1])


def _stub_class(module, name):
    return type(name, (_StubModule,), {'__module__': module})


def _plain_numpy_scalar(scalar):
    """numpy's pickled-scalar constructor, refusing object dtypes: scalar(dtype('O'), bytes) unpickles `bytes` with the
    stock, unrestricted pickle on older numpy"""
    def checked(dtype, *args):
        if getattr(dtype, 'hasobject', True):
            raise pickle.UnpicklingError('refusing to unpickle a numpy scalar of dtype {}'.format(dtype))
        return scalar(dtype, *args)
    return checked


class _CheckpointUnpickler(pickle.Unpickler):
    """
    Resolves torch classes normally and fabricates attribute-bag stand-ins for everything from
    the yolov5 package, so `models.yolo.DetectionModel`, `models.common.Conv` ... unpickle
    without that package (reference pytorch_detector.py:950-957 needs them importable).
    """
    # Exact (module, name) pairs, not namespaces: `torch.*` / `numpy.*` as a whole also hold callables a crafted file
    # could REDUCE (torch.hub.load, torch.utils.cpp_extension.load_inline, numpy.load ...).  This narrows what a file
    # can reach; it is NOT a sandbox -- load checkpoints you trust (the reference does torch.load(weights_only=False),
    # pytorch_detector.py:929-948).
    _SAFE_GLOBALS = {
        ('collections', 'OrderedDict'), ('collections', 'defaultdict'),
        ('torch._utils', '_rebuild_tensor_v2'), ('torch._utils', '_rebuild_tensor'),
        ('torch._utils', '_rebuild_parameter'), ('torch._utils', '_rebuild_parameter_with_state'),
        ('torch._utils', '_rebuild_qtensor'),
        ('torch._tensor', '_rebuild_from_type_v2'), ('torch', 'Tensor'), ('torch', 'Size'), ('torch', 'device'),
        ('torch', 'dtype'), ('torch.nn.parameter', 'Parameter'), ('torch.serialization', '_get_layout'),
        ('torch.storage', 'TypedStorage'), ('torch.storage', 'UntypedStorage'),
        # NOT ('torch.storage', '_load_from_bytes'): it is torch.load(BytesIO(b), weights_only=False), i.e. an
        # unrestricted nested unpickle; torch.save checkpoints reference their storages through persistent_load
        ('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'),
        ('numpy.core.multiarray', 'scalar'), ('numpy._core.multiarray', 'scalar'),
        ('numpy', 'ndarray'), ('numpy', 'dtype'), ('_codecs', 'encode'),
        ('pathlib', 'PosixPath'), ('pathlib', 'WindowsPath'), ('pathlib', 'PurePosixPath'), ('pathlib', 'PureWindowsPath'),
        ('pathlib', 'Path'),
    }
    _TORCH_STORAGE_OR_DTYPE = ('Storage',)          # torch.FloatStorage, torch.HalfStorage, ... (legacy typed storages)
    # plain data types only: `builtins` also holds eval / exec / getattr / __import__
    _SAFE_BUILTINS = ('set', 'frozenset', 'list', 'dict', 'tuple', 'slice', 'range', 'complex', 'int', 'float',
                      'bool', 'str', 'bytes', 'bytearray', 'object')

    def find_class(self, module, name):
        if module in ('builtins', '__builtin__'):          # '__builtin__': protocol-2 pickles (torch.save default)
            if name in self._SAFE_BUILTINS:
                import builtins
                return getattr(builtins, name)
            raise pickle.UnpicklingError('refusing to unpickle {}.{}'.format(module, name))
        if (module, name) in self._SAFE_GLOBALS:
            obj = super().find_class(module, name)
            if name == 'scalar' and module.endswith('multiarray'):
                return _plain_numpy_scalar(obj)
            return obj
        if module == 'torch' and name.endswith(self._TORCH_STORAGE_OR_DTYPE) and '.' not in name:
            return super().find_class(module, name)
        if module.startswith('torch.nn.modules.') and '.' not in name:
            # layer classes of torch.nn (Conv2d, BatchNorm2d, SiLU, Upsample, MaxPool2d, Sequential, ModuleList ...):
            # only actual nn.Module subclasses, never a function of those modules
            cls = super().find_class(module, name)
            import torch
            if isinstance(cls, type) and issubclass(cls, torch.nn.Module):
                return cls
            raise pickle.UnpicklingError('refusing to unpickle {}.{}'.format(module, name))
        if module.split('.')[0] in ('models', 'utils', 'yolov5', 'ultralytics', '__main__'):
            return _stub_class(module, name)
        raise pickle.UnpicklingError('refusing to unpickle {}.{}'.format(module, name))


class _PickleModule(types.ModuleType):
    """pickle_module argument for torch.load"""

    def __init__(self):
        super().__init__('mdhip_pickle')
        self.Unpickler = _CheckpointUnpickler
        self.load = lambda f, **kw: _CheckpointUnpickler(f, **kw).load()
        self.__name__ = 'pickle'


def _modules(obj):
    return getattr(obj, '_modules', {})


def _param(obj, name):
    for bag in ('_parameters', '_buffers'):
        d = getattr(obj, bag, None)
        if d is not None and name in d and d[name] is not None:
            return d[name]
    v = obj.__dict__.get(name)
    if v is None:
        raise KeyError(name)
    return v


def _np32(t):
    return t.detach().float().cpu().numpy().astype(np.float32)


def _fold(conv_mod):
    """yolov5 Conv module (conv + bn [+ act]) -> (w, b) fp32, as fuse_conv_and_bn does."""
    mods = _modules(conv_mod)
    conv = mods['conv']
    w = _np32(_param(conv, 'weight')).astype(np.float64)
    c2 = w.shape[0]
    try:
        cb = _np32(_param(conv, 'bias')).astype(np.float64)
    except KeyError:
        cb = np.zeros(c2)
    if 'bn' not in mods:        # already fused
        return w.astype(np.float32), cb.astype(np.float32)
    bn = mods['bn']
    gamma = _np32(_param(bn, 'weight')).astype(np.float64)
    beta = _np32(_param(bn, 'bias')).astype(np.float64)
    mean = _np32(_param(bn, 'running_mean')).astype(np.float64)
    var = _np32(_param(bn, 'running_var')).astype(np.float64)
    eps = float(bn.__dict__.get('eps', 1e-3))
    # fp32 arithmetic in the same order as fuse_conv_and_bn
    scale = (gamma.astype(np.float32) / np.sqrt(np.float32(eps) + var.astype(np.float32))).astype(np.float32)
    wf = (scale[:, None] * w.astype(np.float32).reshape(c2, -1)).reshape(w.shape).astype(np.float32)
    bf = (scale * cb.astype(np.float32) +
          (beta.astype(np.float32) - gamma.astype(np.float32) * mean.astype(np.float32) /
           np.sqrt(var.astype(np.float32) + np.float32(eps)))).astype(np.float32)
    return wf, bf


def load_checkpoint(path):
    """
    Reads md_v5a.0.0.pt-style checkpoints: {'model': DetectionModel(yaml, model=Sequential[...])}.
    Returns YoloWeights (BN folded, fp32).
    """
    import torch
    ckpt = torch.load(path, map_location='cpu', weights_only=False, pickle_module=_PickleModule())
    model = ckpt['model'] if isinstance(ckpt, dict) and 'model' in ckpt else ckpt
    if isinstance(ckpt, dict) and ckpt.get('ema') is not None and not hasattr(model, 'yaml'):
        model = ckpt['ema']
    yaml = dict(model.__dict__['yaml'])
    if 'anchors' in yaml and not isinstance(yaml['anchors'], (list, tuple)):
        raise ValueError('checkpoint yaml has no explicit anchor list')
    seq = _modules(model)['model']
    layers = _modules(seq)
    specs = resolve_yaml(yaml)
    w = {}
    for s in specs:
        mod = layers[str(s.index)]
        if s.type == MDHIP_DETECT:
            ml = _modules(_modules(mod)['m'])
            for l, name in enumerate(s.conv_names):
                w[name + '.weight'] = _np32(_param(ml[str(l)], 'weight'))
                w[name + '.bias'] = _np32(_param(ml[str(l)], 'bias'))
            w['model.{}.anchors'.format(s.index)] = _np32(_param(mod, 'anchors'))
            continue
        for name in s.conv_names:
            sub = mod
            for part in name.split('.')[2:-1]:        # e.g. 'cv1' / 'm','0','cv1'
                sub = _modules(sub)[part]
            wf, bf = _fold(sub)
            w[name + '.weight'] = wf
            w[name + '.bias'] = bf
    names = model.__dict__.get('names')
    if isinstance(names, (list, tuple)):
        names = {i: n for i, n in enumerate(names)}
    return YoloWeights(yaml, w, names=names, source=path)


def read_metadata_from_megadetector_model_file(model_file):
    """
    reference pytorch_detector.py:674-731: optional '<root>/megadetector_info.json' inside the
    .pt zip (absent for MDv5).  Returns dict or None.
    """
    import json
    try:
        with zipfile.ZipFile(model_file, 'r') as z:
            names = z.namelist()
            roots = set(n.split('/')[0] for n in names)
            if len(roots) != 1:
                return None
            target = next(iter(roots)) + '/megadetector_info.json'
            if target not in names:
                return None
            return json.loads(z.read(target).decode('utf-8'))
    except Exception:
        return None
