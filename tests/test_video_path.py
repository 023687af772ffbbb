"""
Batched video path (megadetector_amd/process_video.py) on in-memory frame sources with
the stub detector of test_batch_loop: the batched run must equal the reference's one-frame-at-a-time loop
(reference video_utils.py:332-470) for every sampling mode, and the per-video JSON must have the
reference's shape (process_video.py:211-258).
"""

import json

import numpy as np
import pytest

from megadetector_amd import process_video as PV
from test_batch_loop import StubDetector, PipelinedStub


def _frames(n, h=36, w=48, seed=0):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(n)]


def _reference_loop(detector, frames, frame_rate, every_n_frames=None, frames_to_process=None, thr=0.005):
    """one frame at a time, as the reference does"""
    if isinstance(frames_to_process, int):
        frames_to_process = [frames_to_process]
    interval = PV._frame_interval(every_n_frames, frame_rate)
    names, res = [], []
    for i, f in enumerate(frames):
        if interval is not None and i % max(1, interval) != 0:
            continue
        if frames_to_process is not None:
            if i > max(frames_to_process):
                break
            if i not in frames_to_process:
                continue
        names.append(PV.frame_number_to_filename(i))
        res.append(detector.generate_detections_one_image(f, names[-1], detection_threshold=thr))
    return names, res


@pytest.mark.parametrize('kw', [dict(), dict(every_n_frames=3), dict(every_n_frames=0), dict(every_n_frames=-0.2),
                                dict(frames_to_process=[1, 4, 5, 11]), dict(frames_to_process=7)])
@pytest.mark.parametrize('batch_size,det_cls', [(1, StubDetector), (4, StubDetector), (4, PipelinedStub), (32, PipelinedStub)])
def test_batched_frames_equal_frame_by_frame(kw, batch_size, det_cls):
    frames = _frames(13)
    names, want = _reference_loop(StubDetector(), frames, 10.0, **kw)
    det = det_cls()
    got = PV.run_detector_on_frames(det, PV.ArrayFrameSource(frames, frame_rate=10.0), batch_size=batch_size,
                                    detection_threshold=0.005, **kw)
    assert got['frame_filenames'] == names and got['frame_rate'] == 10.0
    # the batched stub returns unfiltered detections; compare after the same threshold
    for g, w in zip(got['results'], want):
        g = dict(g)
        g['detections'] = [d for d in g['detections'] if d['conf'] >= 0.005]
        assert json.loads(json.dumps(g)) == json.loads(json.dumps(w))
    if batch_size > 1:
        assert max(det.batches) <= batch_size


def test_exclusive_sampling_arguments_and_empty_video():
    with pytest.raises(ValueError):
        PV.run_detector_on_frames(StubDetector(), PV.ArrayFrameSource(_frames(3)), every_n_frames=2, frames_to_process=[1])
    with pytest.raises(Exception, match='no frames'):
        PV.run_detector_on_frames(StubDetector(), PV.ArrayFrameSource([]))
    r = PV.run_detector_on_frames(StubDetector(), PV.ArrayFrameSource([]), allow_empty_videos=True)
    assert r['results'] == []


def test_video_json_shape_and_failed_video(tmp_path):
    vids = {'a/clip1.mp4': _frames(7, seed=1), 'b/clip2.mp4': _frames(5, seed=2), 'broken.mp4': None}

    def open_source(what):
        if what is None:
            raise RuntimeError('cannot open')
        return PV.ArrayFrameSource(what, frame_rate=25.0)
    md = PV.run_detector_on_videos(PipelinedStub(), list(vids.items()), open_source=open_source, every_n_frames=2,
                                   batch_size=3, detection_threshold=0.0)
    assert md['video_filenames'] == list(vids) and md['frame_rates'] == [25.0, 25.0, -1.0]
    images = PV.video_results_to_md_format(md)
    assert images[0]['frames_processed'] == [0, 2, 4, 6] and images[1]['frames_processed'] == [0, 2, 4]
    assert images[2]['detections'] is None and 'Failure processing video' in images[2]['failure']
    for im in images[:2]:
        assert im['frame_rate'] == 25.0
        assert all(d['frame_number'] in im['frames_processed'] for d in im['detections'])
    from megadetector_amd import run_detector_batch as RDB
    out = tmp_path / 'v.json'
    RDB.write_results_to_file(images, str(out), detector_file='md_v5a.0.0.pt')
    j = json.load(open(out))
    assert [im['file'] for im in j['images']] == sorted(vids)
    assert j['info']['format_version'] == '1.6'


def test_opencv_source_fails_loudly_without_cv2(tmp_path):
    p = tmp_path / 'x.mp4'
    p.write_bytes(b'0')
    try:
        import cv2  # noqa: F401
        pytest.skip('cv2 is installed here')
    except ImportError:
        pass
    with pytest.raises(RuntimeError, match='opencv'):
        PV.OpenCVFrameSource(str(p))
