"""
Host feed for real image files.

The reference moves decoded images from its loader processes to the GPU process by pickling the
arrays through a `JoinableQueue(maxsize=10)` (run_detector_batch.py:92,:124-200,:518) -- at MI355X
speeds that queue, not the GPU, sets the rate (decode is ~7 ms per 3-megapixel JPEG and core, the GPU
takes ~1 ms per image).  Here:

  * loader *processes* (spawned, never forked: a forked child of a process that has touched HIP is
    undefined behaviour) decode with PIL -- same EXIF-rotation and failure semantics as the reference's
    `load_image` (visualization_utils.py:103-175,:306) -- straight into a slot of ONE shared-memory ring;
  * only (file name, slot, shape) crosses the queue; the GPU process sees the pixels as a zero-copy
    NumPy view of the slot;
  * the ring is page-locked in the GPU process (hipHostRegister), so the host-to-device copy of a batch
    is an asynchronous DMA at PCIe rate on a copy stream, overlapped with the previous batch's kernels
    (detector.HIPDetector.start_batch / finish_batch);
  * an image that does not fit a slot falls back to travelling through the queue as an array.

This module must stay import-light (numpy + PIL only): it is what the spawned loader processes import.
"""

import multiprocessing as mp
import os
import queue
import traceback

import numpy as np

EXIF_IMAGE_ROTATIONS = {3: 180, 6: 270, 8: 90}                  # reference visualization_utils.py


def load_image(input_file, ignore_exif_rotation=False):
    """PIL decode to RGB with EXIF rotation (reference visualization_utils.py:103-175, :306)."""
    from PIL import Image
    image = Image.open(input_file)
    if image.mode not in ('RGBA', 'RGB', 'L', 'I;16'):
        raise AttributeError('Image {} uses unsupported mode {}'.format(input_file, image.mode))
    if image.mode in ('RGBA', 'L'):
        image = image.convert(mode='RGB')
    if not ignore_exif_rotation:
        try:
            exif = image._getexif()
            orientation = exif.get(274, None)
            if orientation is not None and orientation != 1:
                assert orientation in EXIF_IMAGE_ROTATIONS, 'Mirrored rotations are not supported'
                image = image.rotate(EXIF_IMAGE_ROTATIONS[orientation], expand=True)
        except Exception:
            pass
    image.load()
    return image


def image_metadata(image):
    """what _add_image_metadata needs from the PIL image, as plain data (reference :769-792)"""
    dt = None
    try:
        exif = image.getexif()
        dt = exif.get(36867) or exif.get(306)       # DateTimeOriginal / DateTime
    except Exception:
        pass
    return {'width': image.width, 'height': image.height, 'datetime': dt}


class ImageMeta:
    """Stands in for the PIL image where only width / height / datetime are read."""

    def __init__(self, meta):
        self.width, self.height, self._dt = meta['width'], meta['height'], meta['datetime']

    def getexif(self):
        return {36867: self._dt} if self._dt is not None else {}


class SharedImageRing:
    """n_slots x slot_bytes of shared memory + the queue of free slot numbers."""

    def __init__(self, n_slots, slot_bytes, ctx):
        from multiprocessing import shared_memory
        self.n_slots, self.slot_bytes = int(n_slots), int(slot_bytes)
        self.shm = shared_memory.SharedMemory(create=True, size=self.n_slots * self.slot_bytes)
        self.free_q = ctx.Queue()
        for i in range(self.n_slots):
            self.free_q.put(i)
        self._registered = False
        self._all = np.ndarray((self.n_slots * self.slot_bytes,), dtype=np.uint8, buffer=self.shm.buf)

    @property
    def name(self):
        return self.shm.name

    def view(self, slot, shape):
        n = int(np.prod(shape))
        off = slot * self.slot_bytes
        return self._all[off:off + n].reshape(shape)

    def pin(self):
        """Page-locks the ring for asynchronous H2D copies (no-op without a HIP device).  Returns bool."""
        try:
            import torch
            if not torch.cuda.is_available():
                return False
            rc = torch.cuda.cudart().cudaHostRegister(self._all.ctypes.data, self._all.nbytes, 0)
            self._registered = int(rc) == 0
        except Exception:
            self._registered = False
        return self._registered

    def release(self, slot):
        self.free_q.put(slot)

    def close(self):
        if self._registered:
            try:
                import torch
                torch.cuda.cudart().cudaHostUnregister(self._all.ctypes.data)
            except Exception:
                pass
            self._registered = False
        self._all = None
        try:
            self.shm.close()
            self.shm.unlink()
        except Exception:
            pass


def _loader_process_main(shm_name, slot_bytes, file_q, free_q, ready_q, want_meta, worker_id):
    """Body of a loader process: file names in, (file, slot, shape) out."""
    from multiprocessing import shared_memory
    shm = shared_memory.SharedMemory(name=shm_name)
    try:
        buf = np.ndarray((shm.size,), dtype=np.uint8, buffer=shm.buf)
        while True:
            im_file = file_q.get()
            if im_file is None:
                break
            try:
                image = load_image(im_file)
                meta = image_metadata(image) if want_meta else None
                arr = np.asarray(image)
                if arr.ndim != 3 or arr.shape[2] != 3 or arr.dtype != np.uint8:
                    raise ValueError('unexpected decoded layout {} {}'.format(arr.shape, arr.dtype))
                if arr.nbytes <= slot_bytes:
                    slot = free_q.get()
                    off = slot * slot_bytes
                    np.copyto(buf[off:off + arr.nbytes].reshape(arr.shape), arr)
                    ready_q.put(('slot', im_file, slot, arr.shape, meta, worker_id))
                else:
                    ready_q.put(('array', im_file, np.ascontiguousarray(arr), arr.shape, meta, worker_id))
            except Exception as e:
                print('Producer process: image {} cannot be loaded:\n{}'.format(im_file, str(e)))
                ready_q.put(('fail', im_file, None, None, None, worker_id))
        del buf
    except Exception:
        traceback.print_exc()
    finally:
        ready_q.put(('done', None, None, None, None, worker_id))
        shm.close()


class ProcessLoader:
    """
    Spawns the loader processes and yields ('slot'|'array'|'fail', file, payload, shape, meta) in
    completion order.  `payload` is the slot number (pixels: ring.view(slot, shape)) or the array.
    """

    def __init__(self, image_files, n_workers, n_slots, slot_bytes, want_meta=False):
        self.ctx = mp.get_context('spawn')
        self.ring = SharedImageRing(n_slots, slot_bytes, self.ctx)
        self.file_q = self.ctx.Queue()
        self.ready_q = self.ctx.Queue()
        self.n_workers = max(1, min(int(n_workers), max(1, len(image_files))))
        for f in image_files:
            self.file_q.put(f)
        for _ in range(self.n_workers):
            self.file_q.put(None)
        self.procs = [self.ctx.Process(target=_loader_process_main,
                                       args=(self.ring.name, self.ring.slot_bytes, self.file_q, self.ring.free_q,
                                             self.ready_q, bool(want_meta), i), daemon=True)
                      for i in range(self.n_workers)]
        for p in self.procs:
            p.start()

    def __iter__(self):
        finished = 0
        while finished < self.n_workers:
            try:
                item = self.ready_q.get(timeout=5.0)
            except queue.Empty:
                if not any(p.is_alive() for p in self.procs):      # all loaders died without saying so
                    break
                continue
            if item[0] == 'done':
                finished += 1
                continue
            yield item[:5]

    def close(self):
        for p in self.procs:
            p.join(timeout=5.0)
            if p.is_alive():
                p.terminate()
        self.ring.close()
