"""
The fp8 mode never calibrates silently (ADVICE r2 / VERDICT r2 item 4): its static activation scales are explicit
(`fp8_scales`, `fp8_scales_file`) or calibration on the first batch is an explicit opt-in -- decided before anything
touches a GPU, so these checks run on CPU.  The GPU side (calibration saved to the file, reuse by another detector)
is in tests/test_gpu_fp8.py.
"""

import json

import pytest

from megadetector_amd import run_detector_batch as RDB
from megadetector_amd.detector import HIPDetector


def test_fp8_detector_refuses_to_start_without_scales():
    with pytest.raises(ValueError, match='fp8_scales'):
        HIPDetector('synthetic:YOLOV5N6_TEST', {'dtype': 'fp8'})
    with pytest.raises(ValueError, match='fp8_scales'):                     # a scales file that does not exist yet
        HIPDetector('synthetic:YOLOV5N6_TEST', {'dtype': 'fp8', 'fp8_scales_file': '/nonexistent/scales.json'})
    # a preprocess-only twin (producer processes) never needs them
    HIPDetector('synthetic:YOLOV5N6_TEST', {'dtype': 'fp8', 'preprocess_only': True})


def _never_called(*a, **k):
    raise AssertionError('the shards must not be spawned')


def _posting_worker(gpu, model_file, files, kwargs, out_q):
    out_q.put((gpu, [{'file': f, 'detections': []} for f in files], None))


def test_sharded_fp8_run_needs_saved_scales(tmp_path):
    files = ['a.jpg', 'b.jpg', 'c.jpg']
    with pytest.raises(ValueError, match='saved scales'):
        RDB.run_sharded('synthetic', files, 2, worker=_never_called, detector_options={'dtype': 'fp8'})
    with pytest.raises(ValueError, match='saved scales'):                   # every shard would calibrate on its own first batch
        RDB.run_sharded('synthetic', files, 2, worker=_never_called,
                        detector_options={'dtype': 'fp8', 'fp8_calibrate_on_first_batch': True})
    scales = tmp_path / 'scales.json'
    scales.write_text(json.dumps({'fp8_scales': [1.0, 2.0]}))
    res = RDB.run_sharded('synthetic', files, 2, worker=_posting_worker,
                          detector_options={'dtype': 'fp8', 'fp8_scales_file': str(scales)})
    assert sorted(r['file'] for r in res) == files
    res = RDB.run_sharded('synthetic', files, 2, worker=_posting_worker,
                          detector_options=['dtype=fp8', 'fp8_scales=0.5;0.25'])
    assert len(res) == 3


def test_sharded_fp8_video_run_needs_saved_scales(tmp_path):
    """the multi-GPU VIDEO path has the same guard as run_sharded (every shard would otherwise calibrate on
    its own first frames and race to replace the same fp8_scales_file)"""
    from megadetector_amd import process_video as PV
    videos = [('a.mp4', '/nonexistent/a.mp4'), ('b.mp4', '/nonexistent/b.mp4')]
    for opts in ({'dtype': 'fp8'}, {'dtype': 'fp8', 'fp8_calibrate_on_first_batch': True,
                                    'fp8_scales_file': str(tmp_path / 'not_yet.json')}):
        with pytest.raises(ValueError, match='saved scales'):
            PV.run_detector_on_videos_sharded('synthetic', videos, 2, detector_options=opts, worker=_never_called)
        with pytest.raises(ValueError, match='saved scales'):
            PV.process_videos('synthetic', '/nonexistent', str(tmp_path / 'out.json'), frame_sample=1, batch_size=2,
                              n_gpus=2, detector_options=opts, videos=videos, shard_worker=_never_called)
