"""
tests/bf16_storage_study.py is an instrument (profiles/r6_bf16_storage_study.txt: where bf16 storage loses the reference's
bars, md_tests.py:96-100); its per-tensor storage emulation must BE the oracle's storage emulation -- the one the GPU
parity tests pin the kernels against -- whenever every tensor has one type.
"""

import torch

import bf16_storage_study as S
import parity_util as PU


def test_mixed_forward_with_one_type_is_the_oracles_emulation():
    from megadetector_amd import weights_io, yolo_yaml
    W = weights_io.synthetic_weights(yolo_yaml.YOLOV5S6_TEST, seed=3)
    x, _ = PU.oracle_input(PU.structured_images(2, 128, 192, seed=5), 192, 64)
    for name, emu in (('none', True), ('all', 'fp16')):
        want, _ = PU.oracle_forward(W, x, emulate_bf16=emu)
        with torch.no_grad():
            got = S.MixedForward(W.yaml, W.torch_state(), S.make_policy(name))(x)
        assert torch.equal(got, want), name


def test_policies_cover_every_tensor_role():
    for name in S.SETS:
        pol = S.make_policy(name)
        for i in range(-1, 33):
            for role in ('in', 'conv', 'cv1', 'cv2', 'hidden', 'm', 'cv3'):
                assert pol(i, role) in ('bf16', 'fp16')
    assert S.make_policy('head+c3out')(6, 'cv3') == 'fp16' and S.make_policy('head+c3out')(6, 'm') == 'bf16'
    assert S.make_policy('only_hidden16')(6, 'hidden') == 'fp16' and S.make_policy('only_hidden16')(23, 'cv3') == 'bf16'
