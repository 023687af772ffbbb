"""CPU-side checks of the drop-in boundary: the shared library builds, loads and exports every
symbol include/mdhip.h declares; without a GPU the product path fails loudly (no fallback)."""

import os
import re

import numpy as np
import pytest

from conftest import REPO


def _header_symbols():
    text = open(os.path.join(REPO, 'include', 'mdhip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mdhip_[a-z_0-9]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as G
    G.build()
    from megadetector_amd import _lib
    lib = _lib.load()
    declared = _header_symbols()
    assert declared, 'no symbols parsed from include/mdhip.h'
    for name in declared:
        assert hasattr(lib, name), 'libmdhip.so does not export {}'.format(name)
    assert sorted(_lib.SYMBOLS) == declared, 'ctypes table and header disagree'
    assert b'gfx950' in lib.mdhip_version()


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from megadetector_amd import weights_io, yolo_yaml
    from megadetector_amd._lib import HipError
    from megadetector_amd.detector import HIPDetector
    W = weights_io.synthetic_weights(yolo_yaml.YOLOV5N6_TEST, seed=1)
    with pytest.raises(HipError, match='no HIP device'):
        HIPDetector(W, {'batch_size': 2, 'max_image_size': 256})
    with pytest.raises(RuntimeError, match='no CPU path'):
        HIPDetector(W, {'force_cpu': 'true'})


def test_preprocess_only_instance_never_touches_hip_and_pickles():
    import pickle
    from megadetector_amd.detector import HIPDetector
    from oracle import pre_post as O
    d = HIPDetector('synthetic', {'preprocess_only': True})
    rng = np.random.default_rng(0)
    for shape in [(1536, 2048), (1080, 1920), (1280, 1280), (600, 901), (2448, 3264)]:
        img = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
        r = d.preprocess_image(img, 'x.jpg')
        g = O.letterbox_geometry(shape, 1280, 64)
        assert r['img_processed'].shape == g['out_hw'] + (3,)
        assert r['letterbox_ratio'] == g['ratio'] and r['letterbox_pad'] == g['pad']
        assert r['img_processed'].geometry == (shape[0], shape[1], g['new_unpad'][1], g['new_unpad'][0], g['top'], g['left'], 0)
        assert r['scaling_shape'] == img.shape and r['target_shape'] == 1280
        rr = pickle.loads(pickle.dumps(r))
        assert rr['img_processed'].shape == r['img_processed'].shape
    with pytest.raises(RuntimeError):
        d.generate_detections_one_batch([img], ['x.jpg'])


def test_postprocess_matches_reference_statements():
    """vectorised host formatting == the reference's per-detection loop (restated in the oracle)"""
    import torch
    from megadetector_amd.postprocess import format_detections
    from oracle import pre_post as O
    rng = np.random.default_rng(3)
    for (h0, w0), (h1, w1) in [((1536, 2048), (960, 1280)), ((1080, 1920), (768, 1280)),
                               ((1280, 1280), (1280, 1280)), ((333, 517), (832, 1280))]:
        k = 120
        x1 = rng.random(k) * w1 * 0.8
        y1 = rng.random(k) * h1 * 0.8
        det = np.stack([x1 - 20, y1 - 20, x1 + rng.random(k) * 400, y1 + rng.random(k) * 300,
                        np.sort(rng.random(k) ** 3)[::-1], rng.integers(0, 3, k)], 1).astype(np.float32)
        for thr in (1e-5, 0.005, 0.2):
            got, gmax = format_detections(det, (h1, w1), (h0, w0, 3), (h0, w0, 3), thr)
            ref, rmax = O.format_detections(torch.from_numpy(det), (h1, w1), (h0, w0, 3), (h0, w0, 3), thr)
            assert got == ref and gmax == rmax
    assert format_detections(np.zeros((0, 6), np.float32), (64, 64), (64, 64, 3), (64, 64, 3), 0.1) == ([], 0.0)
    bad = np.array([[0, 0, 10, 10, 0.9, 7]], dtype=np.float32)
    with pytest.raises(KeyError):
        format_detections(bad, (64, 64), (64, 64, 3), (64, 64, 3), 0.1)


def test_detector_input_validation_without_gpu():
    """reference pytorch_detector.py:1155-1182: an empty batch is an empty result; mixed / mismatched inputs are
    ValueErrors before anything touches the device"""
    from megadetector_amd.detector import HIPDetector
    det = HIPDetector('synthetic', {'preprocess_only': True})
    assert det.generate_detections_one_batch([], []) == []
    img = np.zeros((32, 48, 3), np.uint8)
    with pytest.raises(ValueError, match='must be a list'):
        det.generate_detections_one_batch(img, ['a'])
    with pytest.raises(ValueError, match='image_id must be a list'):
        det.generate_detections_one_batch([img], None)
    with pytest.raises(ValueError, match='Length mismatch'):
        det.generate_detections_one_batch([img, img], ['a'])
    info = det.preprocess_image(img, 'a')
    with pytest.raises(ValueError, match='Mixed input types'):
        det.generate_detections_one_batch([info, img], None)
    with pytest.raises(ValueError, match='Mixed input types'):
        det.generate_detections_one_batch([img, info], ['a', 'b'])
    with pytest.raises(RuntimeError, match='preprocess_only'):
        det.generate_detections_one_batch([img], ['a'])


def test_host_e4m3_quantiser_matches_torch():
    """fp8 mode: the library quantises the 3x3 weights on the host (mdhip_capi.cpp pack(), mdhip_internal.h
    f32_to_e4m3); the oracle uses torch.float8_e4m3fn after a clamp to +-448.  Every e4m3 value, every midpoint
    between neighbours (ties to even), the subnormal range, saturation, signs, and random values must agree."""
    import ctypes as C
    import torch
    from megadetector_amd import _lib
    lib = _lib.load()
    grid = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).to(torch.float32)
    grid = grid[torch.isfinite(grid)]
    pos = torch.sort(grid[grid >= 0]).values
    mids = (pos[:-1] + pos[1:]) / 2
    g = torch.Generator().manual_seed(5)
    rnd = torch.cat([torch.randn(20000, generator=g) * s for s in (1e-3, 0.02, 1.0, 30.0, 400.0)])
    x = torch.cat([grid, mids, -mids, torch.nextafter(mids, mids * 2), torch.nextafter(mids, mids * 0),
                   torch.tensor([0.0, -0.0, 448.0, 449.0, 464.0, 480.0, 1e9, -1e9, 2.0 ** -10, 2.0 ** -11, 1e-30, float('inf')]),
                   rnd]).to(torch.float32).contiguous()
    out = np.empty(x.numel(), dtype=np.uint8)
    xin = x.numpy()
    assert lib.mdhip_f32_to_e4m3(xin.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_uint8)), x.numel()) == 0
    want = x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    # -0.0 and +0.0 results may differ in the sign bit for inputs that round to zero; values must be equal
    got_f = torch.from_numpy(out).view(torch.float8_e4m3fn).to(torch.float32).numpy()
    want_f = torch.from_numpy(want).view(torch.float8_e4m3fn).to(torch.float32).numpy()
    np.testing.assert_array_equal(got_f, want_f)


@pytest.mark.parametrize('table', ['tuned_cfgs.json', 'tuned_cfgs_fp16.json', 'tuned_cfgs_fp8.json'])
def test_shipped_tile_tables_name_configurations_of_this_build(table):
    """every entry of the shipped tile tables names a configuration the library has (entries are resolved by name at
    load time and silently dropped when unknown: a typo would cost speed, not correctness -- so it is checked here);
    per layer geometry the entries of all batch sizes are of one kernel family (an image's result must not depend on
    the batch it travels in); and the 80-channel block has its strip configuration at batch 32"""
    import json
    from megadetector_amd import _lib
    lib = _lib.load()
    names = {lib.mdhip_conv_cfg_name(i).decode(): i for i in range(lib.mdhip_num_conv_cfgs())}
    entries = json.load(open(os.path.join(REPO, 'megadetector_amd', table)))['entries']
    assert len(entries) > 100
    unknown = sorted({e['name'] for e in entries if e.get('name') and e['name'] not in names})
    assert not unknown, unknown
    # the numeric id next to a name is the id of THAT name in this build (a C-ABI caller may hand the ids to
    # mdhip_set_tuned as they are; tools/normalize_tables.py rewrites them after the dispatch table changed)
    stale = sorted({(e['name'], e['cfg'], names[e['name']]) for e in entries if e.get('name') and e.get('cfg') != names[e['name']]})
    assert not stale, ('run tools/normalize_tables.py', stale[:5])
    fam = {}
    for e in entries:
        if not e.get('name'):
            continue
        per_img = round(e['m'] / max(1, e.get('batch', 32)))
        f8 = e['name'].startswith('f8:')                 # (the fp8 table holds e4m3 AND 16-bit entries of a geometry:
        key = (e['n'], e['k'], e['ntaps'], e['stride'], e['has_res'], per_img, f8)      # an op supports only one kind)
        fam.setdefault(key, set()).add(bool(lib.mdhip_cfg_is_bitwise(names[e['name']])))
    mixed = {k: v for k, v in fam.items() if len(v) > 1}
    assert not mixed, mixed
    strip = [e for e in entries if e['n'] == 80 and e['k'] == 720 and e['ntaps'] == 9 and e['has_res'] == 1 and e.get('batch') == 32]
    assert strip and all(e['name'].startswith('v5:strip') for e in strip), strip
