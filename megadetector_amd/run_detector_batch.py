"""
Batch driver: the image loop of the reference (megadetector/detection/run_detector_batch.py) on
top of the HIP detector, emitting the same MegaDetector JSON.

Mirrored entry points (same names, argument meaning and failure conventions):
  load_and_run_detector_batch   reference :1062-1439
  _process_batch                reference :680-831   (batched path; threshold applied afterwards :766)
  _process_image                reference :937-1056  (per-image path; threshold forwarded :988-994)
  write_checkpoint / load_checkpoint   reference :1465 / :1497
  write_results_to_file         reference :1546-1662
  main (CLI)                    reference :1763-2186 (the options that touch this path)
New here (spec = reference notebooks/manage_local_batch.py:496-964):
  run_sharded                   one spawned worker process per GPU, balanced `i % G` sharding of the
                                image list, host-side merge with duplicate/missing checks -- no
                                collectives; output identical to the 1-GPU result.

Loader workers are threads (`use_threads_for_queue=True`, reference :1814) or *spawned* processes that
decode into a page-locked shared-memory ring (`use_threads_for_queue=False`; feed.py, SURVEY.md 8(f)
N1) -- never forked processes: a forked child of a process that has touched HIP is undefined behaviour
(the reference forks before loading the model, run_detector_batch.py:545-557).  In both queue modes the
batches go through the detector's pipelined start_batch / finish_batch when it has them (the GPU works
on batch i while the host formats batch i-1 and assembles batch i+1); results are identical to the
plain loop.
"""

import argparse
import copy
import json
import os
import queue
import shutil
import sys
import threading
import time
from datetime import datetime

from . import feed, placement, run_detector
from .feed import load_image, EXIF_IMAGE_ROTATIONS          # noqa: F401  (re-exported)
from .constants import FAILURE_IMAGE_OPEN, FAILURE_INFER, DEFAULT_OUTPUT_CONFIDENCE_THRESHOLD
from .constants import DEFAULT_DETECTOR_LABEL_MAP

# reference run_detector_batch.py:86-119
default_loaders = 4
default_preprocess_on_image_queue = False
max_queue_size = 10
current_format_version = '1.6'
verbose = False

image_extensions = ('.jpg', '.jpeg', '.gif', '.png')           # reference ct_utils.py:30


# --------------------------------------------------------------------------------------------
# host I/O helpers
# --------------------------------------------------------------------------------------------
def is_image_file(s):
    return os.path.splitext(s)[1].lower() in image_extensions


def find_images(dirname, recursive=False):
    """reference path_utils.py:525 (sorted list of image files)"""
    found = []
    if recursive:
        for root, _, files in os.walk(dirname):
            found += [os.path.join(root, f) for f in files if is_image_file(f)]
    else:
        found = [os.path.join(dirname, f) for f in os.listdir(dirname) if is_image_file(f)]
    return sorted(found)


def parse_kvp_list(items, kv_separator='='):
    """reference ct_utils.py:921: ['a=b','c=d'] (or 'a=b,c=d') -> {'a':'b','c':'d'}"""
    if items is None:
        return {}
    if isinstance(items, str):
        items = [s for s in items.split(',') if s]
    d = {}
    for item in items:
        if kv_separator not in item:
            raise ValueError('Illegal key-value pair: {}'.format(item))
        k, v = item.split(kv_separator, 1)
        d[k.strip()] = v.strip()
    return d


def write_json(path, content, indent=1):
    """reference ct_utils.py:210-251 (force_str=True)"""
    parent = os.path.dirname(path)
    if parent:
        os.makedirs(parent, exist_ok=True)
    with open(path, 'w', newline='\n', encoding='utf-8') as f:
        json.dump(content, f, indent=indent, default=str, ensure_ascii=True)


def _sort_by_key(items, key, reverse=False):
    """reference ct_utils.py:509 (None sorts as smallest)"""
    return sorted(items, key=lambda d: (d[key] is not None, d[key]), reverse=reverse)


# --------------------------------------------------------------------------------------------
# per-batch / per-image processing
# --------------------------------------------------------------------------------------------
def _group_into_batches(items, batch_size):
    """reference :657"""
    if batch_size <= 0:
        raise ValueError('Batch size must be positive')
    return [items[i:i + batch_size] for i in range(0, len(items), batch_size)]


def _add_image_metadata(result, image, include_image_size, include_image_timestamp):
    if isinstance(image, dict):
        image = image['img_original_pil']
    if image is None:
        return
    if include_image_size:
        result['width'] = image.width
        result['height'] = image.height
    if include_image_timestamp:
        dt = None
        try:
            exif = image.getexif()
            dt = exif.get(36867) or exif.get(306)       # DateTimeOriginal / DateTime
        except Exception:
            pass
        result['datetime'] = dt


def _filter_batch_output(dets, names, images, confidence_threshold, include_image_size, include_image_timestamp):
    """reference :760-792: the batched detector call gets no threshold; it is applied to its output"""
    assert len(dets) == len(names)
    out = []
    for i, r in enumerate(dets):
        assert names[i] == r['file']
        if 'failure' not in r:
            r['detections'] = [d for d in r['detections'] if d['conf'] >= confidence_threshold]
            if include_image_size or include_image_timestamp:
                _add_image_metadata(r, images[i], include_image_size, include_image_timestamp)
        else:
            print('Warning: within-batch processing failure for image {}'.format(r['file']))
        out.append(r)
    return out


class _BatchPipeline:
    """
    Feeds batches of (file, image, meta_image, release) to the detector.  With a detector that has
    start_batch / finish_batch (HIPDetector) two batches are kept in flight; otherwise every batch is
    processed synchronously.  `release` (or None) is called once the detector no longer reads the pixels.
    """

    def __init__(self, detector, confidence_threshold, include_image_size, include_image_timestamp, on_results,
                 depth=2):
        self.det = detector
        self.thr = confidence_threshold
        self.inc_size, self.inc_time = include_image_size, include_image_timestamp
        self.on_results = on_results
        self.async_ok = hasattr(detector, 'start_batch') and hasattr(detector, 'finish_batch')
        self.depth = depth
        self.inflight = []

    def submit(self, items):
        if not items:
            return
        names = [it[0] for it in items]
        images = [it[1] for it in items]
        metas = [it[2] if it[2] is not None else it[1] for it in items]
        releases = [it[3] for it in items if it[3] is not None]
        if not self.async_ok:
            try:
                dets = self.det.generate_detections_one_batch(images, names, verbose=verbose)
                res = _filter_batch_output(dets, names, metas, self.thr, self.inc_size, self.inc_time)
            except Exception as e:
                print('Batch processing failure for {} images: {}'.format(len(images), str(e)))
                res = [{'file': n, 'failure': FAILURE_INFER} for n in names]
            for r in releases:
                r()
            self.on_results(res)
            return
        try:
            ticket = self.det.start_batch(images, names, verbose=verbose)
        except Exception as e:
            print('Batch processing failure for {} images: {}'.format(len(images), str(e)))
            for r in releases:
                r()
            self.on_results([{'file': n, 'failure': FAILURE_INFER} for n in names])
            return
        self.inflight.append((ticket, names, metas, releases))
        while len(self.inflight) >= self.depth:
            self._finish_oldest()

    def _finish_oldest(self):
        ticket, names, metas, releases = self.inflight.pop(0)
        try:
            dets = self.det.finish_batch(ticket)
            res = _filter_batch_output(dets, names, metas, self.thr, self.inc_size, self.inc_time)
        except Exception as e:
            print('Batch processing failure for {} images: {}'.format(len(names), str(e)))
            res = [{'file': n, 'failure': FAILURE_INFER} for n in names]
        for r in releases:
            r()
        self.on_results(res)

    def drain(self):
        while self.inflight:
            self._finish_oldest()


def _process_batch(image_items_batch, detector, confidence_threshold, quiet=False, image_size=None,
                   include_image_size=False, include_image_timestamp=False, include_exif_tags=None,
                   augment=False):
    """
    reference :680-831.  Items are file names or (file, image, producer_id) tuples.  As in the
    reference, the batched detector call receives neither the threshold nor image_size/augment
    (:751-754); the confidence threshold is applied to its output (:766-767).
    """
    valid_images, valid_names, batch_results = [], [], []
    for item in image_items_batch:
        if isinstance(item, str):
            try:
                image = load_image(item)
            except Exception as e:
                print('Image {} cannot be loaded: {}'.format(item, str(e)))
                batch_results.append({'file': item, 'failure': FAILURE_IMAGE_OPEN})
                continue
            name = item
        else:
            assert len(item) == 3
            name, image, _ = item
        valid_images.append(image)
        valid_names.append(name)

    valid_results = []
    if valid_images:
        try:
            dets = detector.generate_detections_one_batch(valid_images, valid_names, verbose=verbose)
            valid_results = _filter_batch_output(dets, valid_names, valid_images, confidence_threshold,
                                                 include_image_size, include_image_timestamp)
        except Exception as e:
            print('Batch processing failure for {} images: {}'.format(len(valid_images), str(e)))
            valid_results = [{'file': n, 'failure': FAILURE_INFER} for n in valid_names]
    batch_results.extend(valid_results)
    return batch_results


def _process_image(im_file, detector, confidence_threshold, image=None, quiet=False, image_size=None,
                   include_image_size=False, include_image_timestamp=False, include_exif_tags=None,
                   augment=False):
    """reference :937-1056 (the un-batched path: threshold, image_size and augment ARE forwarded)"""
    if not quiet:
        print('Processing image {}'.format(im_file))
    if image is None:
        try:
            image = load_image(im_file)
        except Exception as e:
            if not quiet:
                print('Image {} cannot be loaded: {}'.format(im_file, str(e)))
            return {'file': im_file, 'failure': FAILURE_IMAGE_OPEN}
    try:
        result = detector.generate_detections_one_image(image, im_file, detection_threshold=confidence_threshold,
                                                        image_size=image_size, augment=augment, verbose=verbose)
    except Exception as e:
        if not quiet:
            print('Image {} cannot be processed: {}'.format(im_file, str(e)))
        return {'file': im_file, 'failure': FAILURE_INFER}
    if 'failure' not in result or result.get('failure') is None:
        _add_image_metadata(result, image, include_image_size, include_image_timestamp)
    return result


# --------------------------------------------------------------------------------------------
# checkpoints (reference :1465-1520)
# --------------------------------------------------------------------------------------------
def write_checkpoint(checkpoint_path, results):
    assert checkpoint_path is not None
    tmp = None
    if os.path.isfile(checkpoint_path):
        tmp = checkpoint_path + '_tmp'
        shutil.copyfile(checkpoint_path, tmp)
    write_json(checkpoint_path, {'checkpoint': results})
    if tmp is not None:
        try:
            os.remove(tmp)
        except Exception as e:
            print('Warning: error removing backup checkpoint file {}:\n{}'.format(tmp, str(e)))


def load_checkpoint(checkpoint_path):
    print('Loading previous results from checkpoint file {}'.format(checkpoint_path))
    with open(checkpoint_path, 'r') as f:
        data = json.load(f)
    if 'checkpoint' not in data:
        raise ValueError('Checkpoint file {} is missing "checkpoint" field'.format(checkpoint_path))
    print('Restored {} entries from the checkpoint {}'.format(len(data['checkpoint']), checkpoint_path))
    return data['checkpoint']


# --------------------------------------------------------------------------------------------
# image queue (threads) -- reference :124-200 producers, :203-455 consumer, :461-650 driver
# --------------------------------------------------------------------------------------------
def _producer(q, file_q, preprocessor, image_size, producer_id):
    while True:
        try:
            im_file = file_q.get_nowait()
        except queue.Empty:
            break
        try:
            image = load_image(im_file)
            if preprocessor is not None:
                image = preprocessor.preprocess_image(image, image_id=im_file, image_size=image_size)
        except Exception as e:
            print('Producer process: image {} cannot be loaded:\n{}'.format(im_file, str(e)))
            image = FAILURE_IMAGE_OPEN
        q.put((im_file, image, producer_id))
    q.put(None)


def _make_preprocessor(detector, detector_options):
    """A weight-free, HIP-free twin of `detector` for the producer side: same options, same geometry attributes."""
    from .detector import HIPDetector
    opts = copy.deepcopy(dict(detector_options or {}))
    if getattr(detector, 'compatibility_mode', None) is not None:
        opts['compatibility_mode'] = detector.compatibility_mode
    opts['preprocess_only'] = True
    pre = HIPDetector('synthetic', opts)
    for attr in ('default_image_size', 'letterbox_stride', 'compatibility_mode'):
        if hasattr(detector, attr):
            setattr(pre, attr, getattr(detector, attr))
    return pre


def _run_detector_with_image_queue(image_files, detector, confidence_threshold, quiet, image_size,
                                   include_image_size, include_image_timestamp, augment, loader_workers,
                                   preprocess_on_image_queue, batch_size, on_results, detector_options=None):
    q = queue.Queue(max_queue_size)
    file_q = queue.Queue()
    for f in image_files:
        file_q.put(f)
    preprocessor = None
    if preprocess_on_image_queue:
        # reference _producer_func (:143-152): load_detector(..., deepcopy(detector_options) + preprocess_only), so the
        # producers letterbox in the consumer's compatibility mode
        preprocessor = _make_preprocessor(detector, detector_options)
    n_workers = max(1, min(loader_workers, len(image_files)))
    threads = [threading.Thread(target=_producer, args=(q, file_q, preprocessor, image_size, i), daemon=True)
               for i in range(n_workers)]
    for t in threads:
        t.start()
    finished = 0
    pending = []
    pipe = _BatchPipeline(detector, confidence_threshold, include_image_size, include_image_timestamp, on_results)

    def flush():
        if pending:
            if batch_size > 1:
                pipe.submit([(f, im, None, None) for f, im, _ in pending])
            else:
                on_results([_process_image(f, detector, confidence_threshold, image=im, quiet=quiet,
                                           image_size=image_size, include_image_size=include_image_size,
                                           include_image_timestamp=include_image_timestamp, augment=augment)
                            for f, im, _ in pending])
            pending.clear()

    while finished < n_workers:
        item = q.get()
        if item is None:
            finished += 1
            continue
        if isinstance(item[1], str):
            on_results([{'file': item[0], 'failure': item[1]}])
            continue
        pending.append(item)
        if len(pending) >= max(1, batch_size):
            flush()
    flush()
    pipe.drain()
    for t in threads:
        t.join()


# the shared-memory ring: slot size (bytes) and slots per image of the batch size
ring_slot_bytes = 48 * 1024 * 1024              # a 16-megapixel RGB frame; larger images travel through the queue
ring_slots_per_batch_image = 3                  # one batch being filled, two in flight


def _run_detector_with_shared_ring(image_files, detector, confidence_threshold, quiet, image_size,
                                   include_image_size, include_image_timestamp, augment, loader_workers,
                                   batch_size, on_results):
    """
    SURVEY.md 8(f) N1 (feed.py): spawned loader processes decode into a page-locked shared-memory ring,
    the batches go through the detector's pipelined interface.  Same results as every other mode.
    """
    bs = max(1, batch_size)
    n_workers = max(1, min(loader_workers, len(image_files)))
    n_slots = ring_slots_per_batch_image * bs + n_workers
    try:
        loader = feed.ProcessLoader(image_files, n_workers, n_slots, ring_slot_bytes,
                                    want_meta=include_image_size or include_image_timestamp)
    except (OSError, MemoryError) as e:
        # e.g. a container whose /dev/shm is smaller than the ring: the thread queue needs no shared memory
        print('Warning: cannot create the shared-memory ring ({} slots of {} MB: {}); using loader threads'.format(
            n_slots, ring_slot_bytes >> 20, str(e)))
        return _run_detector_with_image_queue(image_files, detector, confidence_threshold, quiet, image_size,
                                              include_image_size, include_image_timestamp, augment, loader_workers,
                                              False, batch_size, on_results)
    ring = loader.ring
    try:
        if hasattr(detector, 'start_batch'):
            ring.pin()
        pipe = _BatchPipeline(detector, confidence_threshold, include_image_size, include_image_timestamp, on_results)
        pending = []

        def flush():
            if not pending:
                return
            if bs > 1:
                pipe.submit(list(pending))
            else:
                for f, im, meta_img, release in pending:
                    r = _process_image(f, detector, confidence_threshold, image=im, quiet=quiet,
                                       image_size=image_size, include_image_size=False,
                                       include_image_timestamp=False, augment=augment)
                    if r.get('failure') is None and meta_img is not None:
                        _add_image_metadata(r, meta_img, include_image_size, include_image_timestamp)
                    if release is not None:
                        release()
                    on_results([r])
            pending.clear()

        for kind, im_file, payload, shape, meta in loader:
            if kind == 'fail':
                on_results([{'file': im_file, 'failure': FAILURE_IMAGE_OPEN}])
                continue
            meta_img = feed.ImageMeta(meta) if meta is not None else None
            if kind == 'slot':
                slot = payload
                pending.append((im_file, ring.view(slot, shape), meta_img, (lambda s=slot: ring.release(s))))
            else:
                pending.append((im_file, payload, meta_img, None))
            if len(pending) >= bs:
                flush()
        flush()
        pipe.drain()
    finally:
        loader.close()


# --------------------------------------------------------------------------------------------
# main entry point
# --------------------------------------------------------------------------------------------
def _resolve_image_list(image_file_names):
    """reference :1150-1190: list | single image | folder | .json / .txt list file"""
    if isinstance(image_file_names, str):
        s = image_file_names
        if os.path.isdir(s):
            return find_images(s, recursive=True)
        if is_image_file(s):
            return [s]
        if s.lower().endswith('.json'):
            with open(s, 'r') as f:
                return list(json.load(f))
        if s.lower().endswith('.txt'):
            with open(s, 'r') as f:
                return [ln.strip() for ln in f if ln.strip()]
        raise ValueError('Illegal image_file_names value {}'.format(s))
    return list(image_file_names)


def load_and_run_detector_batch(model_file, image_file_names, checkpoint_path=None,
                                confidence_threshold=DEFAULT_OUTPUT_CONFIDENCE_THRESHOLD,
                                checkpoint_frequency=-1, results=None, n_cores=1, use_image_queue=False,
                                quiet=False, image_size=None, class_mapping_filename=None,
                                include_image_size=False, include_image_timestamp=False,
                                include_exif_tags=None, augment=False, force_model_download=False,
                                detector_options=None, loader_workers=default_loaders,
                                preprocess_on_image_queue=default_preprocess_on_image_queue, batch_size=1,
                                verbose_output=False, use_threads_for_queue=True, detector=None):
    """
    reference :1062-1439.  `detector` (extra, optional) injects an already constructed detector
    object -- used by run_sharded and by the CPU tests of the loop with a stub detector.
    Returns the list of per-image result dicts.
    """
    global verbose
    verbose = bool(verbose_output)
    if detector_options is None:
        detector_options = {}
    elif isinstance(detector_options, (list, str)):
        detector_options = parse_kvp_list(detector_options)
    else:
        detector_options = dict(detector_options)
    if class_mapping_filename is not None or include_exif_tags is not None:
        raise NotImplementedError('class_mapping_filename / include_exif_tags are not part of the HIP hot path')
    if n_cores is not None and n_cores > 1:
        print('Warning: n_cores is ignored when running on a GPU (reference :1204)')
    if confidence_threshold is None:
        confidence_threshold = DEFAULT_OUTPUT_CONFIDENCE_THRESHOLD
    if checkpoint_frequency is None or checkpoint_path is None:
        checkpoint_frequency = -1
    if results is None:
        results = []
    already_processed = set(r['file'] for r in results)
    image_files = [f for f in _resolve_image_list(image_file_names) if f not in already_processed]
    batch_size = max(1, int(batch_size))
    if batch_size > 1:
        detector_options['batch_size'] = batch_size           # reference :1227-1228
    if image_size is not None and int(image_size) > int(detector_options.get('max_image_size', 1280) or 1280):
        # the device arena is planned at construction (default: the model's own size, 1280): a user-supplied size
        # above it (the reference accepts any, :988-994) must be known there, or every image would come back as an
        # inference failure
        detector_options['max_image_size'] = int(image_size)

    if detector is None:
        t0 = time.time()
        detector = run_detector.load_detector(model_file, force_model_download=force_model_download,
                                              detector_options=detector_options, verbose=verbose)
        print('Loaded model in {:.2f} seconds'.format(time.time() - t0))
    # Building the per-image result dicts allocates tens of thousands of containers per batch; every full pass of
    # the cyclic collector they trigger walks the start-up heap (torch, numpy, the model).  Freezing that heap once
    # (the collector stays on) is worth ~7 % of the loop at MI355X speeds.
    import gc
    gc.collect()
    gc.freeze()

    # Checkpoint cadence, as the reference has it: the in-line loops write when the number of images handed to the
    # detector in THIS run is a multiple of the frequency (:1314, :1338 -- with batches, only when a batch boundary
    # lands on a multiple); the queue consumer writes whenever a multiple has been crossed (:272-288).
    counts = {'images': 0, 'last_checkpoint': 0}
    crossing_rule = bool(use_image_queue)

    def on_results(new_results, n_images=None):
        results.extend(new_results)
        before = counts['images']
        counts['images'] = before + (len(new_results) if n_images is None else n_images)
        if checkpoint_path is None or checkpoint_frequency is None or checkpoint_frequency <= 0:
            return
        if crossing_rule:
            due = counts['images'] // checkpoint_frequency > counts['last_checkpoint'] // checkpoint_frequency
        else:
            due = counts['images'] % checkpoint_frequency == 0
        if due:
            print('Writing a new checkpoint after having processed {} images since last restart'.format(
                counts['images']))
            write_checkpoint(checkpoint_path, results)
            counts['last_checkpoint'] = counts['images']

    if use_image_queue and not use_threads_for_queue and len(image_files) > 0:
        _run_detector_with_shared_ring(image_files, detector, confidence_threshold, quiet, image_size,
                                       include_image_size, include_image_timestamp, augment, loader_workers,
                                       batch_size, on_results)
    elif use_image_queue:
        _run_detector_with_image_queue(image_files, detector, confidence_threshold, quiet, image_size,
                                       include_image_size, include_image_timestamp, augment, loader_workers,
                                       preprocess_on_image_queue, batch_size, on_results,
                                       detector_options=detector_options)
    elif batch_size > 1:
        for batch in _group_into_batches(image_files, batch_size):
            on_results(_process_batch(batch, detector, confidence_threshold, quiet, image_size,
                                      include_image_size, include_image_timestamp, None, augment), len(batch))
    else:
        for im_file in image_files:
            on_results([_process_image(im_file, detector, confidence_threshold, quiet=quiet, image_size=image_size,
                                       include_image_size=include_image_size,
                                       include_image_timestamp=include_image_timestamp, augment=augment)])
    # a loader process that died mid-list, or a result dropped anywhere above, must not pass silently
    have = set(r['file'] for r in results)
    missing = [f for f in image_files if f not in have]
    if missing:
        raise RuntimeError('{} images have no result (first: {})'.format(len(missing), missing[0]))
    return results


def write_results_to_file(results, output_file, relative_path_base=None, detector_file=None, info=None,
                          include_max_conf=False, custom_metadata=None, force_forward_slashes=True):
    """reference :1546-1662: MegaDetector batch output format 1.6"""
    out = []
    for r in results:
        r = copy.copy(r)
        if relative_path_base is not None:
            r['file'] = os.path.relpath(r['file'], start=relative_path_base)
        if force_forward_slashes:
            r['file'] = r['file'].replace('\\', '/')
        if not include_max_conf:
            r.pop('max_detection_conf', None)
        out.append(r)
    if info is None:
        info = {'detection_completion_time': datetime.now().strftime('%Y-%m-%d %H:%M:%S'),
                'format_version': current_format_version}
        if detector_file is not None:
            name = os.path.basename(detector_file)
            info['detector'] = name
            info['detector_metadata'] = run_detector.get_detector_metadata_from_version_string(
                run_detector.get_detector_version_from_filename(name, verbose=True))
        else:
            info['detector'] = 'unknown'
            info['detector_metadata'] = run_detector.get_detector_metadata_from_version_string('unknown')
    elif detector_file is not None:
        print('Warning (write_results_to_file): info struct and detector file supplied, ignoring detector file')
    if custom_metadata is not None:
        info['custom_metadata'] = custom_metadata
    out = _sort_by_key(out, 'file')
    for im in out:
        if im.get('detections') is not None:
            im['detections'] = _sort_by_key(im['detections'], 'conf', reverse=True)
        if 'failure' in im:
            assert im.get('detections') is None, 'Illegal failure/detection combination'
            im['detections'] = None
    final_output = {'images': out, 'detection_categories': DEFAULT_DETECTOR_LABEL_MAP, 'info': info}
    write_json(output_file, final_output)
    print('Output file saved at {}'.format(output_file))
    return final_output


# --------------------------------------------------------------------------------------------
# multi-GPU: shard the image queue, one process per GPU, no collectives
# --------------------------------------------------------------------------------------------
def shard_image_list(image_files, n_shards):
    """balanced split (reference ct_utils.py:499-503 'balanced' strategy: shard i gets files i, i+n, ...)"""
    return [list(image_files[i::n_shards]) for i in range(n_shards)]


def merge_shard_results(shard_results, expected_files=None):
    """reference notebooks/manage_local_batch.py:930-964: concatenate, reject duplicates / omissions"""
    merged, seen = [], set()
    for res in shard_results:
        for r in res:
            if r['file'] in seen:
                raise ValueError('duplicate result for {}'.format(r['file']))
            seen.add(r['file'])
            merged.append(r)
    if expected_files is not None:
        missing = [f for f in expected_files if f not in seen]
        if missing:
            raise ValueError('{} images have no result (first: {})'.format(len(missing), missing[0]))
    return merged


def shard_checkpoint_path(checkpoint_path, gpu):
    """every shard process writes its own checkpoint file: G processes must not rewrite one file"""
    return None if checkpoint_path is None else '{}.shard{}'.format(checkpoint_path, gpu)


def _shard_worker(gpu, model_file, files, kwargs, out_q):
    try:
        n_gpus = kwargs.pop('_n_gpus', 1)
        # CPUs of this GPU's NUMA node, disjoint from the other shards'; the loader processes spawned below inherit
        # the mask and are sized to it (placement.py)
        cpus = placement.pin_worker(gpu, n_gpus)
        if cpus and kwargs.get('use_image_queue'):
            kwargs['loader_workers'] = placement.loader_workers_for(kwargs.get('loader_workers', default_loaders), len(cpus))
        opts = dict(kwargs.pop('detector_options', None) or {})
        opts['device'] = 'cuda:{}'.format(gpu)
        kwargs['checkpoint_path'] = shard_checkpoint_path(kwargs.get('checkpoint_path'), gpu)
        res = load_and_run_detector_batch(model_file, files, detector_options=opts, **kwargs)
        out_q.put((gpu, res, None))
    except Exception as e:        # the parent reports it; a dead shard must not hang the join
        out_q.put((gpu, None, repr(e)))


def run_spawned_shards(target, shard_args, n_shards):
    """
    One *spawned* process per shard (a forked child of a process that has touched HIP is undefined behaviour):
    `target(g, *shard_args[g], out_q)` posts (g, result, error-or-None) on out_q.  Returns [result of shard 0, ...];
    raises when a shard reports an error or dies without a result (its exit code is polled: a process killed by a
    signal -- HIP fault, OOM killer -- never posts).  Shared by the image path (run_sharded) and the video path
    (process_video.process_videos).
    """
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    out_q = ctx.Queue()
    procs = []
    for g in range(n_shards):
        p = ctx.Process(target=target, args=(g,) + tuple(shard_args[g]) + (out_q,))
        p.start()
        procs.append(p)
    got = {}
    try:
        while len(got) < n_shards:
            try:
                g, res, err = out_q.get(timeout=2.0)
            except queue.Empty:
                dead = [g for g, p in enumerate(procs) if g not in got and not p.is_alive()]
                if dead and out_q.empty():
                    time.sleep(0.5)                  # its last message may still be in the pipe
                    if out_q.empty():
                        raise RuntimeError('shard {} exited with code {} without a result'.format(
                            dead[0], procs[dead[0]].exitcode))
                continue
            if err is not None:
                raise RuntimeError('shard {} failed: {}'.format(g, err))
            got[g] = res
    except BaseException:
        for p in procs:
            if p.is_alive():
                p.terminate()
        raise
    for p in procs:
        p.join()
    return [got[g] for g in range(n_shards)]


def require_saved_fp8_scales_for_shards(detector_options):
    """A multi-GPU run (image shards here, video shards in process_video.py) in the fp8 mode must start from SAVED
    activation scales: with fp8_calibrate_on_first_batch every shard would calibrate its e4m3 scales on its own first
    batch (and race to write the same fp8_scales_file), so the same image would get different detections depending
    on n_gpus and the shard it lands in."""
    dopts = detector_options or {}
    if str(dopts.get('dtype', '')).lower() == 'fp8' and not dopts.get('fp8_scales') and not (
            dopts.get('fp8_scales_file') and os.path.isfile(dopts['fp8_scales_file'])):
        raise ValueError("dtype 'fp8' on several GPUs needs saved scales (detector_options fp8_scales or an existing "
                         "fp8_scales_file): calibrate once on one GPU first")


def run_sharded(model_file, image_file_names, n_gpus, results=None, worker=None, **kwargs):
    """
    Runs load_and_run_detector_batch on n_gpus GPUs of one node: the image list is split
    `i % n_gpus`, each shard runs in its own *spawned* process pinned to one GPU
    (detector_options['device'] = 'cuda:g', reference pytorch_detector.py:853-858) and to that GPU's share of the
    CPUs (placement.py), results are merged on the host.  There is no inter-GPU traffic.

    Checkpoints: shard g writes `<checkpoint_path>.shard<g>` (shard_checkpoint_path).  `results` (restored from a
    checkpoint, single-GPU or merged from shard files by load_sharded_checkpoints) are kept as they are; only the
    files without a result are sharded, so a resumed run recomputes nothing.
    `worker` replaces _shard_worker in the CPU tests.
    """
    files = _resolve_image_list(image_file_names)
    results = list(results) if results else []
    done = set(r['file'] for r in results)
    todo = [f for f in files if f not in done]
    if n_gpus <= 1:
        return load_and_run_detector_batch(model_file, files, results=results, **kwargs)
    dopts = kwargs.get('detector_options') or {}
    if isinstance(dopts, (list, str)):
        dopts = parse_kvp_list(dopts)
    require_saved_fp8_scales_for_shards(dopts)
    ck = kwargs.get('checkpoint_path')
    if ck is not None and results and (kwargs.get('checkpoint_frequency') or -1) > 0:
        # The shards only know their own new results, and their `<ck>.shard<g>` files are about to be overwritten by
        # this run's first checkpoints (a resumed run is split i % n again): keep everything restored so far in the
        # plain file, which load_sharded_checkpoints unions with the shard files -- a second crash loses nothing.
        write_checkpoint(ck, results)
    shards = shard_image_list(todo, n_gpus)
    shard_kwargs = dict(kwargs)
    if worker is None:
        shard_kwargs['_n_gpus'] = n_gpus
    got = run_spawned_shards(worker or _shard_worker, [(model_file, shards[g], dict(shard_kwargs)) for g in range(n_gpus)],
                             n_gpus)
    return merge_shard_results([results] + got, expected_files=files)


def load_sharded_checkpoints(checkpoint_path, n_gpus):
    """what a sharded run left behind: the union of `<checkpoint_path>.shard<g>` (and of the plain file, if any)"""
    results, seen = [], set()
    paths = [checkpoint_path] + [shard_checkpoint_path(checkpoint_path, g) for g in range(max(1, n_gpus))]
    for p in paths:
        if p and os.path.isfile(p):
            for r in load_checkpoint(p):
                if r['file'] not in seen:
                    seen.add(r['file'])
                    results.append(r)
    return results


def load_previous_results(previous_results_file, image_folder):
    """
    reference :2056-2096: results of an earlier pass over the same folder (relative paths), made absolute the way
    the final output stage expects them.  Returns the list of image entries.
    """
    assert os.path.isfile(previous_results_file), 'Could not find previous results file {}'.format(previous_results_file)
    with open(previous_results_file, 'r') as f:
        previous = json.load(f)
    assert previous['detection_categories'] == DEFAULT_DETECTOR_LABEL_MAP, \
        "Can't merge previous results when those results use a different set of detection categories"
    print('Loaded previous results for {} images from {}'.format(len(previous['images']), previous_results_file))
    assert os.path.isdir(image_folder)
    for im in previous['images']:
        assert not os.path.isabs(im['file']), 'When processing previous results, relative paths are required'
        im['file'] = os.path.join(image_folder, im['file']).replace('\\', '/')
    return previous['images']


# --------------------------------------------------------------------------------------------
# CLI (reference :1763-2186, the options that concern this path)
# --------------------------------------------------------------------------------------------
def main(argv=None):
    ap = argparse.ArgumentParser(description='Run MegaDetector (HIP / MI355X path) on a batch of images')
    ap.add_argument('detector_file', help='.pt checkpoint, known model name (MDV5A ...; resolved through the '
                                          'environment variable of the same name) or "synthetic"')
    ap.add_argument('image_file', help='image file, folder, or .json/.txt list of image paths')
    ap.add_argument('output_file', help='output .json')
    ap.add_argument('--recursive', action='store_true', default=True)
    ap.add_argument('--output_relative_filenames', action='store_true')
    ap.add_argument('--quiet', action='store_true')
    ap.add_argument('--image_size', type=int, default=None)
    ap.add_argument('--augment', action='store_true')
    ap.add_argument('--use_image_queue', action='store_true')
    ap.add_argument('--preprocess_on_image_queue', action='store_true')
    ap.add_argument('--use_threads_for_queue', action='store_true',
                    help='loader threads instead of loader processes (reference :1814); without it the image queue '
                         'uses spawned processes and the page-locked shared-memory ring (feed.py)')
    ap.add_argument('--loader_workers', type=int, default=default_loaders)
    ap.add_argument('--batch_size', type=int, default=1)
    ap.add_argument('--threshold', type=float, default=DEFAULT_OUTPUT_CONFIDENCE_THRESHOLD)
    ap.add_argument('--checkpoint_frequency', type=int, default=-1)
    ap.add_argument('--checkpoint_path', type=str, default=None)
    ap.add_argument('--resume_from_checkpoint', type=str, default=None)
    ap.add_argument('--include_max_conf', action='store_true')
    ap.add_argument('--include_image_size', action='store_true')
    ap.add_argument('--include_image_timestamp', action='store_true')
    ap.add_argument('--detector_options', nargs='*', metavar='KEY=VALUE', default='')
    ap.add_argument('--previous_results_file', type=str, default=None,
                    help='results of a previous run over the same folder: images in it are skipped and its entries '
                         'are merged into the output (needs a folder and --output_relative_filenames; reference :1894)')
    ap.add_argument('--overwrite_handling', type=str, default='overwrite', choices=['skip', 'overwrite', 'error'])
    ap.add_argument('--n_gpus', type=int, default=1, help='shard the image list over this many GPUs')
    ap.add_argument('--verbose', action='store_true')
    args = ap.parse_args(argv)

    assert 0.0 <= args.threshold <= 1.0, 'Confidence threshold needs to be between 0 and 1'
    assert args.output_file.endswith('.json'), 'output_file specified needs to end with .json'
    if args.checkpoint_frequency != -1:
        assert args.checkpoint_frequency > 0, 'Checkpoint_frequency needs to be > 0 or == -1'
    if args.output_relative_filenames:
        assert os.path.isdir(args.image_file), \
            'Could not find folder {}, must supply a folder when --output_relative_filenames is set'.format(args.image_file)
    if args.previous_results_file is not None:
        assert os.path.isdir(args.image_file) and args.output_relative_filenames, \
            'Can only process previous results when using relative paths'
    if os.path.exists(args.output_file):
        if args.overwrite_handling == 'overwrite':
            print('Warning: output file {} already exists and will be overwritten'.format(args.output_file))
        elif args.overwrite_handling == 'skip':
            print('Output file {} exists, returning'.format(args.output_file))
            return
        else:
            raise Exception('Output file {} exists'.format(args.output_file))
    results = None
    checkpoint_path = args.checkpoint_path
    if args.checkpoint_frequency > 0 and checkpoint_path is None:
        checkpoint_path = os.path.join(os.path.dirname(os.path.abspath(args.output_file)),
                                       'md_checkpoint_{}.json'.format(datetime.now().strftime('%Y%m%d%H%M%S')))
    if args.resume_from_checkpoint:
        results = load_sharded_checkpoints(args.resume_from_checkpoint, args.n_gpus) if args.n_gpus > 1 \
            else load_checkpoint(args.resume_from_checkpoint)
    files = _resolve_image_list(args.image_file)
    print('{} image files found in the input'.format(len(files)))
    previous_images = None
    if args.previous_results_file is not None:
        previous_images = load_previous_results(args.previous_results_file, args.image_file)
        previous_set = set(im['file'] for im in previous_images)
        keep = [f for f in files if f.replace('\\', '/') not in previous_set]
        print('Based on previous results file, processing {} of {} images'.format(len(keep), len(files)))
        files = keep
    kwargs = dict(checkpoint_path=checkpoint_path, confidence_threshold=args.threshold,
                  checkpoint_frequency=args.checkpoint_frequency, use_image_queue=args.use_image_queue,
                  quiet=args.quiet, image_size=args.image_size, include_image_size=args.include_image_size,
                  include_image_timestamp=args.include_image_timestamp, augment=args.augment,
                  detector_options=parse_kvp_list(args.detector_options), loader_workers=args.loader_workers,
                  preprocess_on_image_queue=args.preprocess_on_image_queue, batch_size=args.batch_size,
                  verbose_output=args.verbose, use_threads_for_queue=args.use_threads_for_queue)
    t0 = time.time()
    if args.n_gpus > 1:
        results = run_sharded(args.detector_file, files, args.n_gpus, results=results, **kwargs)
    else:
        results = load_and_run_detector_batch(args.detector_file, files, results=results, **kwargs)
    elapsed = time.time() - t0
    print('Finished inference for {} images in {:.1f} s ({:.2f} images per second)'.format(
        len(results), elapsed, len(results) / max(elapsed, 1e-9)))
    base = os.path.abspath(args.image_file) if (args.output_relative_filenames and os.path.isdir(args.image_file)) else None
    if previous_images is not None:                      # reference :2166-2171
        previous_set = set(im['file'] for im in previous_images)
        assert not previous_set.intersection(r['file'] for r in results), \
            'Previous results handling error: redundant image filenames'
        results.extend(previous_images)
    write_results_to_file(results, args.output_file, relative_path_base=base, detector_file=args.detector_file,
                          include_max_conf=args.include_max_conf)
    for cp in [checkpoint_path] + [shard_checkpoint_path(checkpoint_path, g) for g in range(max(1, args.n_gpus))]:
        if cp and os.path.isfile(cp):
            os.remove(cp)
            print('Deleted checkpoint file {}'.format(cp))


if __name__ == '__main__':
    main()
