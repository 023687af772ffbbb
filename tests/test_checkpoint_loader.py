"""
The checkpoint reader (megadetector_amd.weights_io.load_checkpoint) against a checkpoint with the pickle
layout of md_v5a.0.0.pt -- whole fp16 `models.yolo.DetectionModel` module, BatchNorm not fused -- written
by an independent torch.nn implementation of the architecture (tests/fake_yolov5.py).  The reference
loads such a file with the yolov5 package importable and calls `.float().fuse().eval()`
(pytorch_detector.py:929-957); here the package is absent when the file is read.

What is pinned: (1) the stub unpickler resolves every class of the file; (2) BatchNorm folding and the
state-dict name mapping: the oracle's forward on the loaded weights equals the nn.Module's own eval-mode
forward; (3) anchors, strides, class names.
"""

import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fake_yolov5 as FY  # noqa: E402
import parity_util as PU  # noqa: E402

from megadetector_amd import weights_io, yolo_yaml  # noqa: E402


@pytest.fixture(scope='module')
def checkpoint(tmp_path_factory):
    model = FY.build_model(yolo_yaml.YOLOV5N6_TEST, seed=3)
    path = str(tmp_path_factory.mktemp('ckpt') / 'fake_md_v5.pt')
    FY.save_checkpoint(model, path)
    ref_model = FY.build_model(yolo_yaml.YOLOV5N6_TEST, seed=3).half().float()      # what .float() of the file holds
    x = torch.rand(2, 3, 128, 192, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        ref_pred = ref_model(x)
    FY.uninstall()          # from here on `models.*` is not importable, as in a MegaDetector-free process
    return path, x, ref_pred, ref_model


def test_checkpoint_reads_without_the_yolov5_package(checkpoint):
    path, x, ref_pred, ref_model = checkpoint
    assert 'models' not in sys.modules and 'models.yolo' not in sys.modules
    W = weights_io.load_checkpoint(path)
    assert 'models.yolo' not in sys.modules
    assert W.yaml['nc'] == 3 and W.names == {0: 'animal', 1: 'person', 2: 'vehicle'}
    assert W.max_stride == 64
    det = [k for k in W.weights if k.endswith('.anchors')]
    assert len(det) == 1
    want = np.asarray(yolo_yaml.YOLOV5N6_TEST['anchors'], np.float32).reshape(4, 3, 2) / \
        np.asarray([8, 16, 32, 64], np.float32).reshape(4, 1, 1)
    np.testing.assert_allclose(W.weights[det[0]], want, rtol=1e-3)      # stored in fp16 in the file
    # every conv of the topology has a folded weight and bias of the right shape
    n_convs = sum(1 for k in W.weights if k.endswith('.weight'))
    assert n_convs == sum(1 for m in ref_model.modules() if isinstance(m, torch.nn.Conv2d))


def test_folded_weights_reproduce_the_module_forward(checkpoint):
    path, x, ref_pred, _ = checkpoint
    W = weights_io.load_checkpoint(path)
    pred, _ = PU.oracle_forward(W, x, emulate_bf16=False)
    assert pred.shape == ref_pred.shape
    # same arithmetic up to the BatchNorm fold (fp32 re-association): tight tolerance
    np.testing.assert_allclose(pred[..., :4].numpy(), ref_pred[..., :4].numpy(), rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(pred[..., 4:].numpy(), ref_pred[..., 4:].numpy(), rtol=0, atol=5e-4)


def test_refuses_foreign_classes(tmp_path):
    import pickle

    class Evil:
        def __reduce__(self):
            return (os.system, ('true',))
    p = str(tmp_path / 'evil.pt')
    torch.save({'model': Evil()}, p)
    with pytest.raises((pickle.UnpicklingError, RuntimeError, Exception)) as ei:
        weights_io.load_checkpoint(p)
    assert 'refusing' in str(ei.value) or 'posix' in str(ei.value) or 'nt' in str(ei.value)


class _NestedPayload:
    """REDUCEs torch.storage._load_from_bytes(<torch.save blob>): an unrestricted torch.load inside the file"""

    def __init__(self, marker):
        self.marker = marker

    def __reduce__(self):
        import io

        class Inner:
            def __init__(self, marker):
                self.marker = marker

            def __reduce__(self):
                return (open, (self.marker, 'w'))
        buf = io.BytesIO()
        torch.save(Inner(self.marker), buf)
        return (torch.storage._load_from_bytes, (buf.getvalue(),))


def test_refuses_nested_unpickle_gadgets(tmp_path):
    """_load_from_bytes (a nested, unrestricted torch.load) and object-dtype numpy scalars (a nested
    pickle.loads on older numpy) must not be reachable from a checkpoint file"""
    import pickle
    marker = str(tmp_path / 'payload_ran')
    p = str(tmp_path / 'nested.pt')
    torch.save({'model': _NestedPayload(marker)}, p)
    with pytest.raises(Exception) as ei:
        weights_io.load_checkpoint(p)
    assert 'refusing' in str(ei.value) and '_load_from_bytes' in str(ei.value)
    assert not os.path.exists(marker)

    class ObjectScalar:
        def __reduce__(self):
            try:
                from numpy._core.multiarray import scalar
            except ImportError:
                from numpy.core.multiarray import scalar
            return (scalar, (np.dtype('O'), pickle.dumps({'x': 1})))
    p2 = str(tmp_path / 'objscalar.pt')
    torch.save({'model': ObjectScalar()}, p2)
    with pytest.raises(Exception) as ei:
        weights_io.load_checkpoint(p2)
    assert 'refusing' in str(ei.value)
    # plain numeric scalars (yolov5 checkpoints carry np.float64 fitness values) still load
    p3 = str(tmp_path / 'okscalar.pt')
    torch.save({'best_fitness': np.float64(0.5), 'model': None}, p3)
    with open(p3, 'rb') as f:
        d = torch.load(f, map_location='cpu', pickle_module=weights_io._PickleModule(), weights_only=False)
    assert float(d['best_fitness']) == 0.5


def test_oracle_forward_equals_independent_module_at_x6_width(tmp_path):
    """
    The oracle's functional forward (oracle/yolov5.py: the checker of every conv-stack parity test) against a second,
    independently written implementation -- tests/fake_yolov5.py's nn.Module -- on the topology BASELINE.json names
    (YOLOv5x6, widths 80..1280, C3 depths 4/8/12/4, K up to 11520), BatchNorm statistics not trivial, through a
    checkpoint file.  Both are restatements of the published architecture (the yolov5 package is not available
    offline), so this pins the oracle against a second reading, not against upstream.
    """
    yaml = yolo_yaml.YOLOV5X6_MD
    # gain 1.3: with the default 1.75 this depth amplifies fp32 re-association noise ~4x per head C3 (2e-2 at layer 32)
    model = FY.build_model(yaml, seed=5, gain=1.3)
    path = str(tmp_path / 'fake_x6.pt')
    FY.save_checkpoint(model, path)
    ref_model = model.half().float()
    x = torch.rand(1, 3, 128, 192, generator=torch.Generator().manual_seed(11))
    with torch.no_grad():
        ref_pred = ref_model(x)
    FY.uninstall()
    W = weights_io.load_checkpoint(path)
    assert W.max_stride == 64 and W.yaml['nc'] == 3
    assert sum(1 for k in W.weights if k.endswith('.weight')) == 163          # 159 Conv modules + 4 Detect convs (163 convs)
    keep = {}
    pred, _ = PU.oracle_forward(W, x, emulate_bf16=False, keep=keep)
    assert pred.shape == ref_pred.shape == (1, 3 * (16 * 24 + 8 * 12 + 4 * 6 + 2 * 3), 8)
    np.testing.assert_allclose(pred[..., :4].numpy(), ref_pred[..., :4].numpy(), rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(pred[..., 4:].numpy(), ref_pred[..., 4:].numpy(), rtol=0, atol=5e-4)
    # and layer by layer (hooks on the module): the two implementations agree on every intermediate tensor
    outs = {}
    hooks = [m.register_forward_hook(lambda mod, i, o, k=m.i: outs.__setitem__(k, o)) for m in ref_model.model[:-1]]
    with torch.no_grad():
        ref_model(x)
    for h in hooks:
        h.remove()
    for i, t in keep.items():
        emax, emean = PU.rel_err(t.numpy(), outs[i].numpy())
        assert emax < 2e-3 and emean < 2e-4, (i, emax, emean)


def test_refuses_callables_inside_allowed_namespaces(tmp_path):
    """the allow-list names exact (module, name) pairs: torch.* and numpy.* as namespaces also hold loaders a
    crafted file could REDUCE"""
    import pickle

    class ViaTorchHub:
        def __reduce__(self):
            return (torch.hub.load, ('x/y', 'z'))

    class ViaNumpyLoad:
        def __reduce__(self):
            return (np.load, ('/nonexistent.npy',))
    for i, obj in enumerate((ViaTorchHub(), ViaNumpyLoad())):
        p = str(tmp_path / 'evil{}.pt'.format(i))
        torch.save({'model': obj}, p)
        with pytest.raises(Exception) as ei:
            weights_io.load_checkpoint(p)
        assert 'refusing' in str(ei.value), str(ei.value)
