"""
Batched video path.

The reference runs the detector on one frame at a time from inside the decode loop
(`process_video.py:158-198` builds a per-frame callback around `generate_detections_one_image`,
`video_utils.py:332-470` `run_callback_on_frames` calls it between two `vidcap.read()`s; its TODO.md:537
asks for batching within a video).  All sampled frames of a video have one shape, which is the ideal
case for fixed-shape batches: here the sampled frames are collected into batches of `batch_size` and go
through the detector's pipelined `start_batch` / `finish_batch` (decode of batch i+1 overlaps the GPU
work of batch i).  Frame sampling (`every_n_frames`, negative = seconds, `frames_to_process`), frame
identifiers (`frame000123.jpg`, `video_utils.py:274`), failure handling per video and the per-video JSON
shape (`frame_rate`, `frames_processed`, `frame_number` on every detection, `process_video.py:211-258`)
are the reference's.

Decoding is behind a small frame-source interface (`frame_rate`, `n_frames`, iteration over RGB HxWx3
uint8 frames in order): `OpenCVFrameSource` is the reference's decoder (cv2.VideoCapture + BGR->RGB)
and needs `opencv-python`, which this image does not have; `ArrayFrameSource` serves in-memory frames
(tests, or frames decoded elsewhere).
"""

import os
import re

import numpy as np

from . import run_detector, run_detector_batch
from .constants import DEFAULT_OUTPUT_CONFIDENCE_THRESHOLD

default_video_extensions = ('.mp4', '.avi', '.mpeg', '.mpg', '.mov', '.mkv', '.flv')      # reference video_utils.py:32


# --------------------------------------------------------------------------------------------
# frame identifiers (reference video_utils.py:274-304)
# --------------------------------------------------------------------------------------------
def frame_number_to_filename(frame_number):
    return 'frame{:06d}.jpg'.format(frame_number)


def filename_to_frame_number(filename):
    match = re.search(r'frame(\d+)\.jpg', os.path.basename(filename))
    if match is None:
        raise ValueError('{} does not appear to be a frame file'.format(filename))
    return int(match.group(1))


# --------------------------------------------------------------------------------------------
# frame sources
# --------------------------------------------------------------------------------------------
class ArrayFrameSource:
    """frames: sequence (or iterable with n_frames given) of RGB HxWx3 uint8 arrays"""

    def __init__(self, frames, frame_rate=30.0, n_frames=None):
        self.frames = frames
        self.frame_rate = float(frame_rate)
        self.n_frames = int(n_frames if n_frames is not None else len(frames))

    def __iter__(self):
        return iter(self.frames)

    def close(self):
        pass


class OpenCVFrameSource:
    """The reference's decoder: cv2.VideoCapture, first frame read at open (video_utils.py:130-195,:377-440)."""

    def __init__(self, path):
        try:
            import cv2
        except ImportError as e:
            raise RuntimeError('decoding video files needs opencv-python (cv2), which is not installed: {}'.format(e))
        if not os.path.isfile(path):
            raise FileNotFoundError(path)
        self.cv2 = cv2
        self.cap = cv2.VideoCapture(path)
        ok, self.first = self.cap.read()
        if not ok:
            self.cap.release()
            raise RuntimeError('could not read a frame from {}'.format(path))
        self.n_frames = int(self.cap.get(cv2.CAP_PROP_FRAME_COUNT))
        self.frame_rate = float(self.cap.get(cv2.CAP_PROP_FPS))

    def __iter__(self):
        image = self.first
        for i in range(self.n_frames):
            if i != 0:
                ok, image = self.cap.read()
                if not ok:
                    break
            yield self.cv2.cvtColor(image, self.cv2.COLOR_BGR2RGB)

    def close(self):
        try:
            self.cap.release()
        except Exception:
            pass


def find_videos(folder, recursive=True):
    out = []
    for root, _, files in os.walk(folder):
        for f in files:
            if f.lower().endswith(default_video_extensions):
                out.append(os.path.join(root, f).replace('\\', '/'))
        if not recursive:
            break
    return sorted(out)


# --------------------------------------------------------------------------------------------
# one video
# --------------------------------------------------------------------------------------------
def _frame_interval(every_n_frames, frame_rate):
    """reference video_utils.py:391-405"""
    if every_n_frames is None:
        return None
    if every_n_frames < 0:
        return int(abs(every_n_frames) * frame_rate)
    if every_n_frames == 0:
        return 1
    return int(every_n_frames)


def run_detector_on_frames(detector, source, every_n_frames=None, frames_to_process=None, batch_size=8,
                           detection_threshold=DEFAULT_OUTPUT_CONFIDENCE_THRESHOLD, image_size=None,
                           allow_empty_videos=False, verbose=False):
    """
    The batched equivalent of reference `run_callback_on_frames` (video_utils.py:332-470) with the detector
    as the callback.  Returns {'frame_filenames', 'frame_rate', 'results'}; `results[i]` is what
    `detector.generate_detections_one_image(frame, 'frameNNNNNN.jpg', detection_threshold=...)` returns for
    the i-th sampled frame.
    """
    if isinstance(frames_to_process, int):
        frames_to_process = [frames_to_process]
    if frames_to_process is not None and every_n_frames is not None:
        raise ValueError('frames_to_process and every_n_frames are mutually exclusive')
    interval = _frame_interval(every_n_frames, source.frame_rate)
    if interval is not None and interval < 1:
        interval = 1
    wanted = set(frames_to_process) if frames_to_process is not None else None
    last_wanted = max(wanted) if wanted else None
    batch_size = max(1, int(batch_size))
    pipelined = hasattr(detector, 'start_batch') and hasattr(detector, 'finish_batch') and batch_size > 1

    frame_filenames, results, inflight = [], [], []

    def finish_oldest():
        ticket, frames = inflight.pop(0)
        results.extend(detector.finish_batch(ticket))
        del frames

    def flush(frames, ids):
        if not frames:
            return
        if batch_size == 1:
            results.append(detector.generate_detections_one_image(frames[0], ids[0], detection_threshold=detection_threshold,
                                                                  image_size=image_size, verbose=verbose))
        elif pipelined:
            inflight.append((detector.start_batch(list(frames), list(ids), detection_threshold=detection_threshold,
                                                  image_size=image_size, verbose=verbose), list(frames)))
            while len(inflight) >= 2:
                finish_oldest()
        else:
            results.extend(detector.generate_detections_one_batch(list(frames), list(ids),
                                                                  detection_threshold=detection_threshold,
                                                                  image_size=image_size, verbose=verbose))

    frames, ids = [], []
    for frame_number, image in enumerate(source):
        if frame_number >= source.n_frames:
            break
        if interval is not None and (frame_number % interval) != 0:
            continue
        if wanted is not None:
            if frame_number > last_wanted:
                break
            if frame_number not in wanted:
                continue
        name = frame_number_to_filename(frame_number)
        frame_filenames.append(name)
        frames.append(np.ascontiguousarray(image))
        ids.append(name)
        if len(frames) >= batch_size:
            flush(frames, ids)
            frames, ids = [], []
    flush(frames, ids)
    while inflight:
        finish_oldest()
    if len(frame_filenames) == 0:
        if allow_empty_videos:
            print('Warning: found no frames')
        else:
            raise Exception('Error: found no frames')
    assert [r['file'] for r in results] == frame_filenames
    return {'frame_filenames': frame_filenames, 'frame_rate': source.frame_rate, 'results': results}


# --------------------------------------------------------------------------------------------
# many videos -> MegaDetector video results
# --------------------------------------------------------------------------------------------
def run_detector_on_videos(detector, videos, open_source=OpenCVFrameSource, error_on_empty_video=False, **kwargs):
    """
    videos: list of (relative_name, thing handed to open_source).  Same return value as the reference's
    `run_callback_on_frames_for_folder` (video_utils.py:473-583): failed videos get frame rate -1 and a
    {'failure': ...} dict instead of a result list.
    """
    ret = {'video_filenames': [], 'frame_rates': [], 'results': []}
    for rel, what in videos:
        rel = rel.replace('\\', '/')
        ret['video_filenames'].append(rel)
        source = None
        try:
            source = open_source(what)
            r = run_detector_on_frames(detector, source, **kwargs)
        except Exception as e:
            if error_on_empty_video:
                raise
            print('Warning: error processing video {}: {}'.format(rel, str(e)))
            ret['frame_rates'].append(-1.0)
            ret['results'].append({'failure': 'Failure processing video: {}'.format(str(e))})
            continue
        finally:
            if source is not None:
                source.close()
        ret['frame_rates'].append(r['frame_rate'])
        for x in r['results']:
            assert x['file'].startswith('frame')
            x['file'] = rel + '/' + x['file']
        ret['results'].append(r['results'])
    return ret


def video_results_to_md_format(md_results):
    """reference process_video.py:211-258: one entry per video, detections carry `frame_number`"""
    out = []
    for video_fn, rate, res in zip(md_results['video_filenames'], md_results['frame_rates'], md_results['results']):
        im = {'file': video_fn, 'frame_rate': rate, 'frames_processed': []}
        if isinstance(res, dict):
            assert 'failure' in res
            im['failure'] = res['failure']
            im['detections'] = None
        else:
            im['detections'] = []
            for one in res:
                assert one['file'].startswith(video_fn)
                n = filename_to_frame_number(one['file'])
                assert n not in im['frames_processed'], 'Received the same frame twice for video {}'.format(video_fn)
                im['frames_processed'].append(n)
                for det in (one.get('detections') or []):
                    det['frame_number'] = n
                im['detections'].extend(one.get('detections') or [])
        im['frames_processed'] = sorted(im['frames_processed'])
        out.append(im)
    return out


# --------------------------------------------------------------------------------------------
# many videos on several GPUs (BASELINE.json configs[3]): shard the VIDEO list, one process per GPU, no collectives
# --------------------------------------------------------------------------------------------
def video_cost(what):
    """what a video costs, for balancing: its frame count when the source says so, else its file size"""
    n = getattr(what, 'n_frames', None)
    if n is not None:
        return int(n)
    if isinstance(what, (list, tuple)):
        return len(what)
    try:
        return int(os.path.getsize(what))
    except Exception:
        return 1


def shard_videos(videos, n_shards, cost=video_cost):
    """
    Assigns whole videos to shards so that the summed cost is balanced: largest first, each to the lightest shard
    (ties: lowest shard index) -- deterministic.  The reference's out-of-process recipe splits the extracted FRAME
    list into equal chunks (notebooks/manage_video_batch.py:51-67 -> manage_local_batch.py:496); whole videos keep
    the decode of a file and its frames' fixed shape on one GPU.  Returns [[index into videos, ...] per shard], each
    in the original order.
    """
    order = sorted(range(len(videos)), key=lambda i: (-cost(videos[i][1]), i))
    load = [0] * n_shards
    shards = [[] for _ in range(n_shards)]
    for i in order:
        g = min(range(n_shards), key=lambda k: (load[k], k))
        shards[g].append(i)
        load[g] += max(1, cost(videos[i][1]))
    return [sorted(sh) for sh in shards]


def _video_shard_worker(gpu, model_file, videos, opts, run_kwargs, n_gpus, out_q):
    try:
        from . import placement
        placement.pin_worker(gpu, n_gpus)
        opts = dict(opts)
        opts['device'] = 'cuda:{}'.format(gpu)
        detector = run_detector.load_detector(model_file, detector_options=opts)
        import gc
        gc.collect()
        gc.freeze()
        out_q.put((gpu, run_detector_on_videos(detector, videos, **run_kwargs), None))
    except Exception as e:
        out_q.put((gpu, None, repr(e)))


def merge_video_shards(videos, shards, shard_results):
    """the per-shard returns of run_detector_on_videos back in the order of `videos`, with the duplicate / omission
    checks of the image path's merge (reference notebooks/manage_local_batch.py:930-964)"""
    n = len(videos)
    slots = [None] * n
    for idx, md in zip(shards, shard_results):
        assert len(md['video_filenames']) == len(idx), 'a shard returned {} of its {} videos'.format(
            len(md['video_filenames']), len(idx))
        for k, i in enumerate(idx):
            if slots[i] is not None:
                raise ValueError('duplicate result for video {}'.format(videos[i][0]))
            assert md['video_filenames'][k] == videos[i][0].replace('\\', '/')
            slots[i] = (md['video_filenames'][k], md['frame_rates'][k], md['results'][k])
    missing = [videos[i][0] for i in range(n) if slots[i] is None]
    if missing:
        raise ValueError('{} videos have no result (first: {})'.format(len(missing), missing[0]))
    return {'video_filenames': [s_[0] for s_ in slots], 'frame_rates': [s_[1] for s_ in slots],
            'results': [s_[2] for s_ in slots]}


def run_detector_on_videos_sharded(model_file, videos, n_gpus, detector_options=None, worker=None, **run_kwargs):
    """
    run_detector_on_videos on n_gpus GPUs: whole videos are assigned to GPUs (shard_videos), every shard runs in a
    spawned process with its own detector (device cuda:g) and CPU set (placement.py); the result equals the
    one-process result.  `worker` replaces the shard process body in the CPU tests.
    """
    from .run_detector_batch import run_spawned_shards, require_saved_fp8_scales_for_shards
    if n_gpus > 1:
        require_saved_fp8_scales_for_shards(detector_options)
    shards = shard_videos(videos, n_gpus)
    args = [(model_file, [videos[i] for i in shards[g]], dict(detector_options or {}), dict(run_kwargs), n_gpus)
            for g in range(n_gpus)]
    return merge_video_shards(videos, shards, run_spawned_shards(worker or _video_shard_worker, args, n_gpus))


def process_videos(model_file, input_video_file, output_json_file=None, frame_sample=None, time_sample=None,
                   json_confidence_threshold=DEFAULT_OUTPUT_CONFIDENCE_THRESHOLD, image_size=None, recursive=True,
                   batch_size=8, detector_options=None, detector=None, exit_on_empty_video=False, verbose=False,
                   n_gpus=1, videos=None, open_source=None, shard_worker=None):
    """
    reference process_video.py:123-272 (the options that touch this path).  n_gpus > 1 (BASELINE.json configs[3],
    reference notebooks/manage_video_batch.py:51-67,201-218 does it out of process): the video list is sharded over
    the GPUs of the node, the per-video JSON is the one-process one.  `videos` ([(relative name, source argument)]) and
    `open_source` replace the folder scan and the cv2 decoder (tests, frames decoded elsewhere).
    """
    if frame_sample is not None and time_sample is not None:
        raise ValueError('frame_sample and time_sample are mutually exclusive')
    if output_json_file is None:
        v = input_video_file.replace('\\', '/')
        output_json_file = (v[:-1] if v.endswith('/') else v) + '.json'
        print('Output file not specified, defaulting to {}'.format(output_json_file))
    assert output_json_file.endswith('.json'), 'Illegal output file {}'.format(output_json_file)
    every = -1 * time_sample if time_sample is not None else frame_sample
    opts = dict(detector_options or {})
    if batch_size > 1:
        opts['batch_size'] = batch_size
    if image_size is not None and int(image_size) > int(opts.get('max_image_size', 1280) or 1280):
        opts['max_image_size'] = int(image_size)     # the device arena is planned at construction
    if videos is None:
        if os.path.isfile(input_video_file):
            videos = [(os.path.basename(input_video_file), input_video_file)]
        else:
            assert os.path.isdir(input_video_file), '{} is neither a file nor a folder'.format(input_video_file)
            folder = input_video_file
            videos = [(os.path.relpath(f, folder).replace('\\', '/'), f) for f in find_videos(folder, recursive=recursive)]
    run_kwargs = dict(every_n_frames=every, batch_size=batch_size, detection_threshold=json_confidence_threshold,
                      image_size=image_size, error_on_empty_video=exit_on_empty_video, verbose=verbose)
    if open_source is not None:
        run_kwargs['open_source'] = open_source
    if n_gpus > 1 and len(videos) > 1:
        assert detector is None, 'a detector object cannot be shared between GPU processes: pass the model file'
        md = run_detector_on_videos_sharded(model_file, videos, min(n_gpus, len(videos)), detector_options=opts,
                                            worker=shard_worker, **run_kwargs)
    else:
        if detector is None:
            detector = run_detector.load_detector(model_file, detector_options=opts)
        import gc
        gc.collect()
        gc.freeze()              # see run_detector_batch.load_and_run_detector_batch
        md = run_detector_on_videos(detector, videos, **run_kwargs)
    print('Finished running MD on videos')
    images = video_results_to_md_format(md)
    run_detector_batch.write_results_to_file(images, output_json_file, relative_path_base=None, detector_file=model_file)
    return images
