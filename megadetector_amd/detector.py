"""
HIPDetector: drop-in for the reference's PTDetector
(megadetector/detection/pytorch_detector.py:739-1480) with the per-batch hot path running as
hand-written HIP on one MI355X (libmdhip.so).

Same constructor, same three public methods, same return dicts and failure conventions
(SURVEY.md section 8(b)):
  preprocess_image(img, image_id, image_size, verbose)            -> dict      (:964)
  generate_detections_one_batch(imgs, ids, threshold, ...)        -> list[dict] (:1124)
  generate_detections_one_image(img, id, threshold, ...)          -> dict      (:1428)

Differences, all deliberate and documented in DESIGN.md:
  * the letterbox resize runs on the GPU, so `preprocess_image` only computes the letterbox
    geometry; 'img_processed' is a LetterboxSpec placeholder exposing `.shape` (dicts carrying
    a real letterboxed ndarray, e.g. from the reference's own preprocessing workers, are
    accepted too);
  * in the non-classic compatibility modes the pre-resize (cv2.INTER_AREA / INTER_LINEAR to the long side) runs on
    the GPU as well; 'img_original' stays the caller's array and 'resized_shape' carries the resized size.
augment=True runs yolov5's augmented inference (three scaled / flipped passes) on the device
(mdhip_forward_tta).
"""

import json
import os

import numpy as np

from . import weights_io
from .constants import (FAILURE_IMAGE_OPEN, FAILURE_INFER, DEFAULT_COMPATIBILITY_MODE)
from .postprocess import letterbox_geometry, modern_geometry, format_detections


def parse_bool_string(s):
    """reference megadetector/utils/ct_utils.py:1000"""
    if isinstance(s, bool):
        return s
    s = str(s).lower().strip()
    if s in ('true', '1', 'yes', 'y'):
        return True
    if s in ('false', '0', 'no', 'n'):
        return False
    raise ValueError('Cannot convert "{}" to bool'.format(s))


class LetterboxSpec:
    """Placeholder for the letterboxed image: the pixels are produced on the GPU."""

    __slots__ = ('shape', 'geometry')

    def __init__(self, shape, geometry):
        self.shape = tuple(shape)
        self.geometry = tuple(geometry)    # (src_h, src_w, resized_h, resized_w, top, left)

    def __reduce__(self):
        return (LetterboxSpec, (self.shape, self.geometry))


def _device_ordinal(device):
    s = str(device).lower()
    if s.startswith('cuda'):
        return int(s.split(':')[1]) if ':' in s else 0
    raise ValueError('HIPDetector needs a GPU device ("cuda:N"), got {}'.format(device))


# storage type when detector_options does not name one (see HIPDetector.__init__)
DEFAULT_DTYPE = 'fp16'


class HIPDetector:

    def __init__(self, model_path, detector_options=None, verbose=False):
        """
        model_path: a YOLOv5 .pt checkpoint (md_v5a.0.0.pt ...), a YoloWeights object, or the
        string 'synthetic[:yaml_name[:seed]]' for seeded weights on the MDv5 topology.
        detector_options keys honoured: force_cpu (must be false), use_model_native_classes,
        compatibility_mode, preprocess_only, device, batch_size, max_image_size, dtype, hip_graph ('auto' | 'on' | 'off').
        dtype: storage type of activations and packed weights (accumulation is fp32 either way).  Default 'fp16':
        |d conf| against the fp32 evaluation the reference performs stays below the reference's own bar between
        environments (0.005-0.01, md_tests.py:96-100,1779) with an 8x margin on every model measured
        (profiles/r2a_accuracy_x6.txt); 'bf16' is the throughput configuration BASELINE.json names (3 % faster,
        8 significant bits); 'fp8' is BASELINE.json configs[4]: bf16 storage, the bottleneck 3x3 convs on e4m3 operands;
        its static activation scales come from `fp8_scales` (a list saved from an earlier run, HipContext.fp8_scales)
        or, without it, from the FIRST batch this detector processes (mdhip_calibrate) -- results then depend on that
        batch; a stated-tolerance throughput mode, not a reference-grade one (tests/test_gpu_fp8.py).
        """
        opts = dict(detector_options or {})
        self.use_model_native_classes = parse_bool_string(opts.get('use_model_native_classes', False))
        compat = opts.get('compatibility_mode') or DEFAULT_COMPATIBILITY_MODE
        self.compatibility_mode = compat
        preprocess_only = bool(opts.get('preprocess_only', False))
        if verbose or not preprocess_only:
            print('Loading HIP detector with compatibility mode {}'.format(compat))

        self.model_metadata = None
        if isinstance(model_path, str) and not model_path.startswith('synthetic'):
            self.model_metadata = weights_io.read_metadata_from_megadetector_model_file(model_path)
        if self.model_metadata is not None and 'image_size' in self.model_metadata:
            self.default_image_size = self.model_metadata['image_size']
            print('Loaded image size {} from model metadata'.format(self.default_image_size))
        else:
            # reference pytorch_detector.py:802-805
            if not preprocess_only:
                print('No image size available in model metadata, defaulting to 1280')
            self.default_image_size = 1280
        self.device = 'cpu'
        self.printed_image_size_warning = False
        # reference :827-845
        self.letterbox_stride = 64 if self.default_image_size == 1280 else 32
        self.half_precision = False
        self.model = None
        self._ctx = None
        if preprocess_only:
            return                      # never touches HIP: safe in forked producer processes

        if parse_bool_string(opts.get('force_cpu', False)):
            raise RuntimeError('HIPDetector has no CPU path (force_cpu requested)')
        device = opts.get('device') or 'cuda:0'
        self.device = device
        # AddaxAI parses this line from the reference (pytorch_detector.py:885)
        print('PTDetector using device {}'.format(str(self.device).lower()))

        if isinstance(model_path, str):
            if model_path.startswith('synthetic'):
                parts = model_path.split(':')
                from . import yolo_yaml
                yaml = getattr(yolo_yaml, parts[1]) if len(parts) > 1 and parts[1] else yolo_yaml.YOLOV5X6_MD
                seed = int(parts[2]) if len(parts) > 2 else 0
                weights = weights_io.synthetic_weights(yaml, seed=seed)
            else:
                weights = weights_io.load_checkpoint(model_path)
        else:
            weights = model_path
        self.weights = weights
        if weights.max_stride != self.letterbox_stride and verbose:
            print('*** Warning: model stride is {}, letterbox stride is {} ***'.format(
                weights.max_stride, self.letterbox_stride))
        # fp8 mode: where do the static activation scales come from?  Decided before anything touches the GPU, so that
        # a missing calibration is an immediate, clear error (also in the parent of a multi-GPU run)
        self._fp8_pending = False
        self._fp8_scales_file = opts.get('fp8_scales_file') or None
        fp8_scales = None
        if str(opts.get('dtype') or DEFAULT_DTYPE).lower() == 'fp8':
            fp8_scales = opts.get('fp8_scales')
            if isinstance(fp8_scales, str):                   # "a;b;c" from a key=value command line
                fp8_scales = [v for v in fp8_scales.replace(';', ' ').split() if v]
            if not fp8_scales and self._fp8_scales_file and os.path.isfile(self._fp8_scales_file):
                with open(self._fp8_scales_file, 'r') as f:
                    fp8_scales = json.load(f)['fp8_scales']
            if not fp8_scales:
                if parse_bool_string(opts.get('fp8_calibrate_on_first_batch', False)):
                    # explicit opt-in: the scales then come from whatever batch arrives first, i.e. the output depends
                    # on file order / batch size / shard; with fp8_scales_file they are saved for the next run
                    self._fp8_pending = True
                else:
                    raise ValueError(
                        "dtype 'fp8' needs its activation scales: pass detector_options['fp8_scales'] (a list saved "
                        "from HipContext.fp8_scales) or ['fp8_scales_file'] (json written by an earlier calibration), or "
                        "opt into calibrating on the first batch with ['fp8_calibrate_on_first_batch']=True -- results "
                        "then depend on that batch")
        from .hip_backend import HipContext
        self.max_batch = int(opts.get('batch_size', 1)) if int(opts.get('batch_size', 1)) > 1 else int(opts.get('max_batch', 8))
        max_size = int(opts.get('max_image_size', self.default_image_size))
        max_size = -(-max_size // weights.max_stride) * weights.max_stride
        if 'classic' not in compat:
            max_size += weights.max_stride      # the modern target shape is ceil(size / stride + 0.5) * stride
        self._ctx = HipContext(weights, device=_device_ordinal(device), dtype=opts.get('dtype') or DEFAULT_DTYPE,
                               max_batch=self.max_batch, max_h=max_size, max_w=max_size)
        self.model = self._ctx
        # launch plumbing, off by default: replaying the forward from a captured graph ('on'; 'auto' = batches <= 8) was
        # measured and changes nothing -- at batch 1 .. 8 the step is bound by its kernels, not by their launches
        self._ctx.set_graph(opts.get('hip_graph', 'off'))
        if fp8_scales:
            self._ctx.set_fp8_scales([float(v) for v in fp8_scales])

    # -----------------------------------------------------------------------------------
    def preprocess_image(self, img_original, image_id='unknown', image_size=None, verbose=False):
        """reference pytorch_detector.py:964-1119 ('classic'): geometry only, pixels stay put."""
        result = {'file': image_id}
        img_original_pil = None
        if not isinstance(img_original, np.ndarray):
            img_original_pil = img_original
            img_original = np.asarray(img_original)
        if img_original.ndim != 3 or img_original.shape[2] != 3 or img_original.dtype != np.uint8:
            raise ValueError('expected an HxWx3 uint8 RGB image, got {} {}'.format(
                img_original.shape, img_original.dtype))
        scaling_shape = img_original.shape
        if image_size is not None:
            assert isinstance(image_size, int)
            if not self.printed_image_size_warning:
                print('Using user-supplied image size {}'.format(image_size))
                self.printed_image_size_warning = True
        else:
            image_size = self.default_image_size
            self.printed_image_size_warning = False
        if 'classic' in self.compatibility_mode:
            g = letterbox_geometry(img_original.shape[:2], new_shape=image_size, stride=self.letterbox_stride,
                                   auto=True, scaleup=True)
            geometry = (img_original.shape[0], img_original.shape[1], g['new_unpad'][1], g['new_unpad'][0],
                        g['top'], g['left'], 0)
            target_shape = image_size
        else:
            # 'modern' (reference :1036-1101): resize to the long side (INTER_AREA when shrinking) and pad into the
            # target shape -- both on the device; the reference replaces img_original by the resized image, here
            # the pixels stay put and 'resized_shape' carries what the box rescaling needs
            m = modern_geometry(img_original.shape[:2], image_size, self.letterbox_stride,
                                use_ceil='use_ceil_for_resize' in self.compatibility_mode)
            g = m['letterbox']
            geometry = (img_original.shape[0], img_original.shape[1], m['resized_hw'][0], m['resized_hw'][1],
                        g['top'], g['left'], m['interp'])
            target_shape = m['target_shape']
            result['resized_shape'] = (m['resized_hw'][0], m['resized_hw'][1], 3)
        result['img_processed'] = LetterboxSpec((g['out_hw'][0], g['out_hw'][1], 3), geometry)
        result['img_original'] = img_original
        result['img_original_pil'] = img_original_pil
        result['target_shape'] = target_shape
        result['scaling_shape'] = scaling_shape
        result['letterbox_ratio'] = g['ratio']
        result['letterbox_pad'] = g['pad']
        return result

    # -----------------------------------------------------------------------------------
    def generate_detections_one_batch(self, img_original, image_id=None, detection_threshold=0.00001,
                                      image_size=None, augment=False, verbose=False):
        """reference pytorch_detector.py:1124-1252"""
        if not isinstance(img_original, list):
            raise ValueError('img_original must be a list for batch processing')
        if len(img_original) == 0:
            return []
        if isinstance(img_original[0], dict):
            for i, img in enumerate(img_original):
                if not isinstance(img, dict):
                    raise ValueError('Mixed input types in batch: item {} is not a dict, but item 0 is a dict'.format(i))
        else:
            if image_id is None:
                raise ValueError('image_id must be a list when img_original contains PIL/numpy images')
            if not isinstance(image_id, list):
                raise ValueError('image_id must be a list for batch processing')
            if len(image_id) != len(img_original):
                raise ValueError('Length mismatch: img_original has {} items, image_id has {} items'.format(
                    len(img_original), len(image_id)))
            for i_img, img in enumerate(img_original):
                if isinstance(img, dict):
                    raise ValueError('Mixed input types in batch: item {} is a dict, but item 0 is not a dict'.format(i_img))
        if detection_threshold is None:
            detection_threshold = 0.0
        if self._ctx is None:
            raise RuntimeError('this HIPDetector was created with preprocess_only')
        results, shape_groups = self._prepare_batch(img_original, image_id, image_size, verbose)
        for shape, items in shape_groups.items():
            try:
                for start in range(0, len(items), self.max_batch):
                    self._process_batch_group(items[start:start + self.max_batch], results,
                                              detection_threshold, augment, verbose)
            except Exception as e:
                print('Warning: batch inference failed for shape {}: {}'.format(shape, str(e)))
                for original_idx, _, current_id in items:
                    results[original_idx] = {'file': current_id, 'detections': None, 'failure': FAILURE_INFER}
        return results

    def _prepare_batch(self, img_original, image_id, image_size, verbose):
        """per-image preprocessing with failure capture (reference :1194-1222) and grouping by processed
        shape (:1226-1233); returns (results with the failed slots filled in, {shape: [(idx, info, id)]})"""
        results = [None] * len(img_original)
        preprocessed = []
        for i_img, img in enumerate(img_original):
            try:
                if isinstance(img, dict):
                    info = img
                    current_id = info['file']
                else:
                    current_id = image_id[i_img]
                    info = self.preprocess_image(img, image_id=current_id, image_size=image_size, verbose=verbose)
                preprocessed.append((i_img, info, current_id))
            except Exception as e:
                current_id = image_id[i_img] if image_id else 'index_{}'.format(i_img)
                print('Warning: preprocessing failed for image {}: {}'.format(current_id, str(e)))
                results[i_img] = {'file': current_id, 'detections': None, 'failure': FAILURE_IMAGE_OPEN}
        shape_groups = {}
        for item in preprocessed:
            shape_groups.setdefault(tuple(item[1]['img_processed'].shape), []).append(item)
        return results, shape_groups

    def _nms_iou(self):
        return 0.45 if 'classic' in self.compatibility_mode else 0.6        # reference :1318-1321

    @staticmethod
    def _group_inputs(group_items):
        images, geoms = [], []
        for _, info, _ in group_items:
            ip = info['img_processed']
            if isinstance(ip, LetterboxSpec):
                images.append(np.ascontiguousarray(info['img_original']))
                geoms.append(ip.geometry)
            else:                      # an already letterboxed HWC u8 array
                ip = np.ascontiguousarray(ip)
                images.append(ip)
                geoms.append((ip.shape[0], ip.shape[1], ip.shape[0], ip.shape[1], 0, 0))
        return images, geoms

    def _format_group(self, group_items, det_all, counts, h, w, results, detection_threshold):
        for i, (original_idx, info, current_id) in enumerate(group_items):
            det = det_all[i, :counts[i]]
            modern = 'classic' not in self.compatibility_mode
            detections, max_conf = format_detections(
                det, (h, w), info.get('resized_shape', info['img_original'].shape) if modern else info['img_original'].shape,
                info['scaling_shape'], detection_threshold,
                use_model_native_classes=self.use_model_native_classes, modern=modern,
                letterbox_pad=info.get('letterbox_pad'))
            results[original_idx] = {'file': current_id, 'detections': detections,
                                     'max_detection_conf': max_conf}

    def _fp8_calibrated(self):
        self._fp8_pending = False
        if self._fp8_scales_file:
            tmp = '{}.{}.tmp'.format(self._fp8_scales_file, os.getpid())
            with open(tmp, 'w') as f:
                json.dump({'fp8_scales': [float(sc) for sc, _, _ in self._ctx.fp8_scales()]}, f)
            os.replace(tmp, self._fp8_scales_file)

    def _process_batch_group(self, group_items, results, detection_threshold, augment, verbose):
        """reference pytorch_detector.py:1257-1426 with the device work in libmdhip.so"""
        if len(group_items) == 0:
            return
        h, w = group_items[0][1]['img_processed'].shape[:2]
        images, geoms = self._group_inputs(group_items)
        n = len(group_items)
        ctx = self._ctx
        ctx.preprocess(images, geoms, h, w)
        if self._fp8_pending:               # fp8 mode, explicit opt-in: this batch calibrates the scales
            ctx.calibrate(n, h, w)
            self._fp8_calibrated()
        if augment:
            ctx.forward_tta(n, h, w)        # yolov5 _forward_augment: 3 passes, concatenated predictions
        else:
            ctx.forward(n, h, w)
        det_all, counts = ctx.nms(n, detection_threshold, self._nms_iou(), max_det=300)
        self._format_group(group_items, det_all, counts, h, w, results, detection_threshold)

    # -----------------------------------------------------------------------------------
    # Pipelined variant of generate_detections_one_batch for the batch driver (feed.py): the device work
    # of a batch is enqueued on a private stream and the call returns; finish_batch() waits for it and
    # formats.  Host images are copied to the device on a copy stream into one of two staging buffers
    # (asynchronously when they live in page-locked memory, e.g. feed.SharedImageRing), so the copy of
    # batch i+1 overlaps the kernels of batch i.  Same kernels, same results as the synchronous call.
    # -----------------------------------------------------------------------------------
    def _pipeline(self):
        if getattr(self, '_pl', None) is None:
            import torch
            dev = torch.device('cuda', _device_ordinal(self.device))
            with torch.cuda.device(dev):
                self._pl = {'torch': torch, 'dev': dev, 'copy_s': torch.cuda.Stream(), 'comp_s': torch.cuda.Stream(),
                            'nms_s': torch.cuda.Stream(), 'nms_done': [None] * 4,
                            'stage': [None, None], 'copied': [torch.cuda.Event(), torch.cuda.Event()],
                            'consumed': [None, None], 'count': 0}
        return self._pl

    def _submit_group(self, group_items, detection_threshold, augment=False):
        pl = self._pipeline()
        torch = pl['torch']
        h, w = group_items[0][1]['img_processed'].shape[:2]
        images, geoms = self._group_inputs(group_items)
        n = len(group_items)
        k = pl['count'] % 2
        nms_slot = pl['count'] % 4
        pl['count'] += 1
        offs, total = [], 0
        for im in images:
            offs.append(total)
            total += (im.nbytes + 255) // 256 * 256
        with torch.cuda.device(pl['dev']):
            if pl['stage'][k] is None or pl['stage'][k].numel() < total:
                if pl['consumed'][k] is not None:
                    pl['consumed'][k].synchronize()
                pl['stage'][k] = torch.empty(max(total, 1), dtype=torch.uint8, device=pl['dev'])
            stage = pl['stage'][k]
            with torch.cuda.stream(pl['copy_s']):
                if pl['consumed'][k] is not None:
                    pl['copy_s'].wait_event(pl['consumed'][k])      # the letterbox kernel that read this buffer is done
                for im, off in zip(images, offs):
                    flat = im.reshape(-1)
                    if not flat.flags.writeable:          # torch warns on read-only arrays; the copy only reads
                        flat = flat.view()
                        try:
                            flat.flags.writeable = True
                        except ValueError:
                            flat = np.array(flat)
                    stage[off:off + im.nbytes].copy_(torch.from_numpy(flat), non_blocking=True)
                pl['copied'][k].record(pl['copy_s'])
            comp = pl['comp_s']
            base = stage.data_ptr()
            ctx = self._ctx
            comp.wait_event(pl['copied'][k])
            # (the letterbox on the copy stream next to the previous batch's forward -- mdhip_preprocess waits for that
            # forward's stem inside the library -- was measured with bench.py --pre-own-stream: 0.6 % slower, its workgroups
            # keep the 8-wave conv workgroups off their CUs; it stays on the compute stream)
            ctx.preprocess([base + off for off in offs], geoms, h, w, stream=comp.cuda_stream)
            if self._fp8_pending:
                ctx.calibrate(n, h, w, stream=comp.cuda_stream)
                self._fp8_calibrated()
            ev = torch.cuda.Event()
            ev.record(comp)
            pl['consumed'][k] = ev
            # NMS + D2H on their own stream, next to the following batch's letterbox and first layers (the library
            # alternates between two prediction buffers; the forward that reuses a buffer waits for the NMS that read it).
            # History: own stream in round 1; in line behind the forward in rounds 2-3 (the 1024-thread NMS workgroups kept
            # the persistent conv workgroups off their CUs: 37.6 vs 37.2 ms / step); since round 4 the NMS sorts only the
            # confidence band it needs and is gone before the next forward reaches its 8-wave kernels: own stream again
            # (+0.4 .. 0.7 % at batch 32, profiles/r4_bench_nms_stream.txt).
            prev = pl['nms_done'][(pl['count'] - 3) % 4]             # (count was incremented above: the batch two before this one)
            if prev is not None:
                comp.wait_event(prev)
            if augment:
                ctx.forward_tta(n, h, w, stream=comp.cuda_stream)
            else:
                ctx.forward(n, h, w, stream=comp.cuda_stream)
            fwd_done = torch.cuda.Event()
            fwd_done.record(comp)
            nms_s = pl['nms_s']
            nms_s.wait_event(fwd_done)
            ctx.nms_enqueue(n, detection_threshold, self._nms_iou(), 300, slot=nms_slot, stream=nms_s.cuda_stream)
            done = torch.cuda.Event()
            done.record(nms_s)
            pl['nms_done'][nms_slot] = done
        return {'items': group_items, 'h': h, 'w': w, 'slot': nms_slot, 'copied': pl['copied'][k], 'images': images}

    def _collect_group(self, handle, results, detection_threshold):
        det_all, counts = self._ctx.nms_wait(slot=handle['slot'])
        self._format_group(handle['items'], det_all, counts, handle['h'], handle['w'], results, detection_threshold)

    def start_batch(self, img_original, image_id, detection_threshold=0.00001, image_size=None, augment=False,
                    verbose=False):
        """Enqueues a batch; returns a ticket for finish_batch().  At most two tickets may be outstanding.
        Same arguments as generate_detections_one_batch (augment = yolov5's three-pass augmented inference)."""
        if self._ctx is None:
            raise RuntimeError('this HIPDetector was created with preprocess_only')
        if detection_threshold is None:
            detection_threshold = 0.0
        results, shape_groups = self._prepare_batch(img_original, image_id, image_size, verbose)
        chunks = []
        for shape, items in shape_groups.items():
            for start in range(0, len(items), self.max_batch):
                chunks.append(items[start:start + self.max_batch])
        pending = None
        for ci, chunk in enumerate(chunks):
            try:
                handle = self._submit_group(chunk, detection_threshold, augment)
                if ci == len(chunks) - 1:
                    pending = handle                     # the last group stays in flight
                else:
                    self._collect_group(handle, results, detection_threshold)
            except Exception as e:
                print('Warning: batch inference failed for shape {}: {}'.format(chunk[0][1]['img_processed'].shape, str(e)))
                for original_idx, _, current_id in chunk:
                    results[original_idx] = {'file': current_id, 'detections': None, 'failure': FAILURE_INFER}
        return {'results': results, 'pending': pending, 'threshold': detection_threshold}

    def batch_inputs_consumed(self, ticket):
        """Blocks until the host images of the ticket's in-flight group have been copied to the device
        (their buffers -- e.g. shared-ring slots -- may then be reused)."""
        if ticket['pending'] is not None:
            ticket['pending']['copied'].synchronize()

    def finish_batch(self, ticket):
        results = ticket['results']
        handle = ticket['pending']
        if handle is not None:
            try:
                self._collect_group(handle, results, ticket['threshold'])
            except Exception as e:
                print('Warning: batch inference failed: {}'.format(str(e)))
                for original_idx, _, current_id in handle['items']:
                    results[original_idx] = {'file': current_id, 'detections': None, 'failure': FAILURE_INFER}
            ticket['pending'] = None
        return results

    # -----------------------------------------------------------------------------------
    def generate_detections_one_image(self, img_original, image_id='unknown', detection_threshold=0.00001,
                                      image_size=None, augment=False, verbose=False):
        """reference pytorch_detector.py:1428-1478"""
        if isinstance(img_original, dict):
            res = self.generate_detections_one_batch([img_original], None, detection_threshold,
                                                     image_size, augment, verbose)
        else:
            res = self.generate_detections_one_batch([img_original], [image_id], detection_threshold,
                                                     image_size, augment, verbose)
        return res[0]
