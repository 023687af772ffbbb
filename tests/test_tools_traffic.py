"""tools/hbm_traffic.py: the per-op alignment of rocprofv3 counter rows with the forward's ops (counter
bytes next to ALGORITHMIC bytes per kernel instantiation), on synthetic counter files -- the tool runs on the GPU box, its
arithmetic is checked here."""

import csv
import json
import os
import sys

from conftest import REPO

sys.path.insert(0, os.path.join(REPO, 'tools'))


def _write_pass(d, counter, steps, kernels, values):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, 'x_counter_collection.csv'), 'w', newline='') as f:
        w = csv.DictWriter(f, fieldnames=['Dispatch_Id', 'Kernel_Name', 'Counter_Name', 'Counter_Value'])
        w.writeheader()
        did = 1
        for s in range(steps):
            w.writerow({'Dispatch_Id': did, 'Kernel_Name': 'mdhip::letterbox_copy_s2d_kernel(args)', 'Counter_Name': counter, 'Counter_Value': 1})
            did += 1
            for k, v in zip(kernels, values):
                w.writerow({'Dispatch_Id': did, 'Kernel_Name': k + '(mdhip::ConvArgs)', 'Counter_Name': counter, 'Counter_Value': v})
                did += 1
            w.writerow({'Dispatch_Id': did, 'Kernel_Name': 'mdhip::nms_scan_kernel(x)', 'Counter_Name': counter, 'Counter_Value': 5})
            did += 1


def test_counter_rows_are_aligned_with_the_ops_and_priced_against_algorithmic_bytes(tmp_path):
    import hbm_traffic as T
    M = 1000
    ops = [
        {'op': 0, 'name': 'L1 conv 3x3s2', 'kind': 0, 'm': M, 'n': 160, 'k': 720, 'ntaps': 9, 'stride': 2, 'cfg': 5, 'bytes': M * 4 * 80 * 2.0 + M * 160 * 2.0},
        {'op': 1, 'name': 'L2 C3.m0.cv1 1x1', 'kind': 0, 'm': M, 'n': 80, 'k': 80, 'ntaps': 1, 'stride': 1, 'cfg': -1, 'bytes': 0.0},       # fused away
        {'op': 2, 'name': 'L2 C3.m0.cv2 3x3', 'kind': 0, 'm': M, 'n': 80, 'k': 720, 'ntaps': 9, 'stride': 1, 'cfg': 7, 'bytes': M * 80 * 2.0 * 2},
        {'op': 3, 'name': 'L13 upsample x2', 'kind': 2, 'm': 0, 'n': 0, 'k': 0, 'ntaps': 0, 'stride': 1, 'cfg': -1, 'bytes': 0.0},          # read in place
        {'op': 4, 'name': 'L33 Detect.decode0', 'kind': 3, 'm': 0, 'n': 0, 'k': 0, 'ntaps': 0, 'stride': 1, 'cfg': -1, 'bytes': 8000.0},
    ]
    json.dump(ops, open(tmp_path / 'ops.json', 'w'))
    kernels = ['void mdhip::st_bf16::conv_v7_kernel<0, true>', 'void mdhip::st_bf16::conv_c80f_kernel<160, 2>', 'mdhip::detect_decode_3x8_kernel']
    # counters are KiB; FETCH_SIZE reports half of the bytes of a wide stream
    _write_pass(str(tmp_path / 'f'), 'FETCH_SIZE', 3, kernels, [M * 4 * 80 * 2.0 * 1.5 / 2 / 1024, M * 80 * 2.0 / 2 / 1024, 4000.0 / 2 / 1024])
    _write_pass(str(tmp_path / 'w'), 'WRITE_SIZE', 3, kernels, [M * 160 * 2.0 / 1024, M * 80 * 2.0 / 1024, 4000.0 / 1024])
    tab, why = T.per_op_table(str(tmp_path / 'f'), str(tmp_path / 'w'), str(tmp_path / 'ops.json'))
    assert why is None and tab['steps_aligned'] == [3, 3]
    rows = {r['op']: r for r in tab['per_op']}
    assert sorted(rows) == [0, 2, 4]                                           # the fused 1x1 and the absorbed upsample launch nothing
    assert abs(rows[0]['read_ratio'] - 1.5) < 1e-9 and abs(rows[0]['write_ratio'] - 1.0) < 1e-9
    assert abs(rows[2]['read_ratio'] - 1.0) < 1e-9 and rows[2]['kernel'].endswith('conv_c80f_kernel<160, 2>')
    assert abs(rows[4]['read_algorithmic'] - 4000.0) < 1e-9
    txt = T.format_table(tab)
    assert 'conv_v7_kernel<0, true>' in txt and '1.50' in txt
    # a pass whose steps have another number of forward dispatches is refused with a reason, not mis-aligned
    _write_pass(str(tmp_path / 'f2'), 'FETCH_SIZE', 2, kernels[:2], [1.0, 1.0])
    tab, why = T.per_op_table(str(tmp_path / 'f2'), str(tmp_path / 'w'), str(tmp_path / 'ops.json'))
    assert tab is None and 'forward dispatches' in why
