"""
Batched video path (megadetector_amd/process_video.py) on in-memory frame sources with
the stub detector of test_batch_loop: the batched run must equal the reference's one-frame-at-a-time loop
(reference video_utils.py:332-470) for every sampling mode, and the per-video JSON must have the
reference's shape (process_video.py:211-258).
"""

import json
import os

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


class _StandInCv2:
    """the part of cv2's call protocol OpenCVFrameSource uses (video_utils.py:130-195,:377-440): VideoCapture(path)
    .read() -> (ok, BGR frame), .get(CAP_PROP_*), .release(); cvtColor(image, COLOR_BGR2RGB).  Frames come from an .npy
    file; `short_by` makes the stream end before the frame count the container reports (truncated files do that)."""
    CAP_PROP_FRAME_COUNT, CAP_PROP_FPS, COLOR_BGR2RGB = 7, 5, 4
    released = 0

    class VideoCapture:
        def __init__(self, path):
            d = np.load(path, allow_pickle=True).item() if os.path.getsize(path) > 8 else None
            self.frames = [] if d is None else list(d['bgr'])
            self.fps = 0.0 if d is None else d['fps']
            self.reported = 0 if d is None else len(self.frames) + d.get('short_by', 0)
            self.i = 0

        def read(self):
            if self.i >= len(self.frames):
                return False, None
            self.i += 1
            return True, self.frames[self.i - 1]

        def get(self, prop):
            return float(self.reported) if prop == _StandInCv2.CAP_PROP_FRAME_COUNT else self.fps

        def release(self):
            _StandInCv2.released += 1

    @staticmethod
    def cvtColor(image, code):
        assert code == _StandInCv2.COLOR_BGR2RGB
        return image[:, :, ::-1]


def test_opencv_source_protocol_with_a_stand_in_cv2(tmp_path, monkeypatch):
    """cv2 is not installed here, so the reference's decoder path (OpenCVFrameSource: first frame read at open, frame
    count and rate from the container, BGR -> RGB, early end of stream, unreadable file) runs against a stand-in module
    with cv2's call protocol; the frames it yields then go through the same batched loop as every other source"""
    import sys
    monkeypatch.setitem(sys.modules, 'cv2', _StandInCv2)
    rgb = _frames(7, seed=4)
    good = tmp_path / 'clip.mp4'
    with open(good, 'wb') as f:
        np.save(f, {'bgr': [fr[:, :, ::-1].copy() for fr in rgb], 'fps': 30.0}, allow_pickle=True)
    src = PV.OpenCVFrameSource(str(good))
    assert src.n_frames == 7 and src.frame_rate == 30.0
    got = list(src)
    assert len(got) == 7 and all(np.array_equal(a, b) for a, b in zip(got, rgb))         # RGB again, in order
    src.close()
    assert _StandInCv2.released == 1
    # the container reports more frames than can be decoded: the source stops at the last good frame
    short = tmp_path / 'short.mp4'
    with open(short, 'wb') as f:
        np.save(f, {'bgr': [fr[:, :, ::-1].copy() for fr in rgb[:4]], 'fps': 15.0, 'short_by': 3}, allow_pickle=True)
    s2 = PV.OpenCVFrameSource(str(short))
    assert s2.n_frames == 7 and len(list(s2)) == 4
    s2.close()
    # no decodable frame at all: loud failure at open, capture released
    bad = tmp_path / 'bad.mp4'
    bad.write_bytes(b'0')
    with pytest.raises(RuntimeError, match='could not read a frame'):
        PV.OpenCVFrameSource(str(bad))
    with pytest.raises(FileNotFoundError):
        PV.OpenCVFrameSource(str(tmp_path / 'missing.mp4'))
    # through the video driver with the default source: same detections as the in-memory source, failed file reported
    md = PV.run_detector_on_videos(PipelinedStub(), [('clip.mp4', str(good)), ('bad.mp4', str(bad))], every_n_frames=2, batch_size=3,
                                   detection_threshold=0.0)
    ref = PV.run_detector_on_frames(PipelinedStub(), PV.ArrayFrameSource(rgb, frame_rate=30.0), every_n_frames=2,
                                    batch_size=3, detection_threshold=0.0)
    assert md['frame_rates'] == [30.0, -1.0]
    images = PV.video_results_to_md_format(md)
    assert images[0]['frames_processed'] == [0, 2, 4, 6] and images[1]['detections'] is None
    got_dets = sorted((d['frame_number'], d['conf'], tuple(d['bbox'])) for d in images[0]['detections'])
    ref_dets = sorted((int(r['file'][5:11]), d['conf'], tuple(d['bbox'])) for r in ref['results'] for d in (r.get('detections') or []))
    assert got_dets == ref_dets and len(got_dets) > 0


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[3]: videos across the GPUs of a node (reference notebooks/manage_video_batch.py:51-67,201-218)
# ---------------------------------------------------------------------------------------------------------------------
def _open_array(what):
    if what is None:
        raise RuntimeError('cannot open')
    return PV.ArrayFrameSource(what['frames'], frame_rate=what['fps'])


def _stub_video_shard_worker(gpu, model_file, videos, opts, run_kwargs, n_gpus, out_q):
    """stands in for the GPU process of one shard: same driver code, stub detector; reports which device it was given"""
    try:
        md = PV.run_detector_on_videos(PipelinedStub(), videos, **run_kwargs)
        md['_gpu'] = gpu
        out_q.put((gpu, md, None))
    except Exception as e:
        out_q.put((gpu, None, repr(e)))


def _video_set():
    lens = [9, 3, 14, 5, 1, 7, 11]
    vids = [('cam{}/clip{}.mp4'.format(i % 3, i), {'frames': _frames(n, seed=10 + i), 'fps': 10.0 + i})
            for i, n in enumerate(lens)]
    vids.insert(4, ('cam1/broken.mp4', None))
    return vids


def test_video_sharding_is_balanced_and_deterministic():
    vids = _video_set()
    cost = lambda w: 1 if w is None else len(w['frames'])          # noqa: E731
    for g in (2, 3, 8):
        shards = PV.shard_videos(vids, g, cost=cost)
        assert sorted(i for sh in shards for i in sh) == list(range(len(vids)))         # every video exactly once
        assert shards == PV.shard_videos(vids, g, cost=cost)
        loads = [sum(cost(vids[i][1]) for i in sh) for sh in shards]
        assert max(loads) - min(l for l in loads if l) <= max(cost(v[1]) for v in vids)  # LPT bound
    two = PV.shard_videos(vids, 2, cost=cost)
    loads = [sum(cost(vids[i][1]) for i in sh) for sh in two]
    assert abs(loads[0] - loads[1]) <= 1                                                   # 51 frames: 25 / 26
    # default cost: frame count of in-memory lists, file size for paths
    assert PV.video_cost([1, 2, 3]) == 3 and PV.video_cost('/nonexistent/file.mp4') == 1


def test_videos_across_two_gpu_processes_equal_the_one_process_output(tmp_path):
    """world 2, spawned shard processes (stub detector): the merged per-video JSON is byte-identical to the
    one-process run, a video that cannot be opened fails alone, a dead shard is noticed"""
    vids = _video_set()
    one = str(tmp_path / 'one.json')
    two = str(tmp_path / 'two.json')
    kw = dict(frame_sample=2, batch_size=4, videos=vids, open_source=_open_array, json_confidence_threshold=0.0)
    im1 = PV.process_videos('md_v5a.0.0.pt', 'unused', one, detector=PipelinedStub(), **kw)
    im2 = PV.process_videos('md_v5a.0.0.pt', 'unused', two, n_gpus=2, shard_worker=_stub_video_shard_worker, **kw)
    assert json.loads(json.dumps(im1)) == json.loads(json.dumps(im2))
    strip = lambda p: __import__('re').sub(r'"detection_completion_time": "[^"]*"', '', open(p).read())   # noqa: E731
    assert strip(one) == strip(two)
    assert [im['file'] for im in im2] == [v[0] for v in vids]
    broken = [im for im in im2 if im['file'] == 'cam1/broken.mp4'][0]
    assert broken['detections'] is None and 'Failure processing video' in broken['failure']
    # more GPUs than videos: never an empty shard process
    im3 = PV.process_videos('md_v5a.0.0.pt', 'unused', str(tmp_path / 'three.json'), n_gpus=16,
                            shard_worker=_stub_video_shard_worker, **kw)
    assert json.loads(json.dumps(im3)) == json.loads(json.dumps(im1))


def _dying_video_worker(gpu, model_file, videos, opts, run_kwargs, n_gpus, out_q):
    if gpu == 1:
        os._exit(3)
    _stub_video_shard_worker(gpu, model_file, videos, opts, run_kwargs, n_gpus, out_q)


def test_dead_video_shard_raises(tmp_path):
    vids = _video_set()
    with pytest.raises(RuntimeError, match='exited with code 3'):
        PV.process_videos('md_v5a.0.0.pt', 'unused', str(tmp_path / 'x.json'), n_gpus=2, videos=vids,
                          open_source=_open_array, shard_worker=_dying_video_worker, batch_size=4)
