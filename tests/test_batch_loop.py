"""
CPU tests of the host side of the path: the image loop, checkpoint/resume, the JSON writer and
the multi-GPU sharding logic -- with a deterministic stub detector standing in for the GPU, and
a world_size-2 gloo run of the one-process-per-GPU path.
"""

import json
import os
import socket
import sys

import numpy as np
import pytest

from conftest import REPO
from megadetector_amd import run_detector_batch as RDB
from megadetector_amd import run_detector


from stub_detector import StubDetector  # noqa: E402


@pytest.fixture()
def image_dir(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    names = []
    for i in range(11):
        sub = tmp_path / ('cam%d' % (i % 3))
        sub.mkdir(exist_ok=True)
        p = sub / ('img_%02d.jpg' % i)
        Image.fromarray(rng.integers(0, 256, (40 + i, 64, 3), dtype=np.uint8)).save(p, quality=95)
        names.append(str(p))
    bad = tmp_path / 'cam0' / 'broken.jpg'
    bad.write_bytes(b'not a jpeg')
    names.append(str(bad))
    return tmp_path, sorted(names)


def _strip(results):
    return json.loads(json.dumps(sorted(results, key=lambda r: r['file'])))


def test_orchestration_modes_give_identical_results(image_dir):
    """reference md_tests.py:1251,1267,1283: queue / preprocess-queue / batched runs must equal the plain run"""
    root, names = image_dir
    plain = RDB.load_and_run_detector_batch('stub', str(root), detector=StubDetector(), quiet=True)
    assert sorted(r['file'] for r in plain) == names
    broken = [r for r in plain if r['file'].endswith('broken.jpg')][0]
    assert broken == {'file': broken['file'], 'failure': run_detector.FAILURE_IMAGE_OPEN}
    for kw in (dict(batch_size=4), dict(use_image_queue=True), dict(use_image_queue=True, batch_size=3),
               dict(use_image_queue=True, batch_size=3, preprocess_on_image_queue=True, loader_workers=2)):
        det = StubDetector()
        got = RDB.load_and_run_detector_batch('stub', names, detector=det, quiet=True, **kw)
        assert _strip(got) == _strip(plain), kw
        if kw.get('batch_size', 1) > 1:
            assert max(det.batches) <= kw['batch_size']


def test_shared_memory_ring_loader_processes(image_dir, monkeypatch):
    """SURVEY.md 8(f) N1 (megadetector_amd/feed.py): spawned loader processes decoding into the shared-memory
    ring give the plain run's results -- batched and unbatched, with image size/timestamp metadata, with a
    broken file, and with images that do not fit a ring slot (they travel through the queue instead)."""
    root, names = image_dir
    plain = RDB.load_and_run_detector_batch('stub', names, detector=StubDetector(), quiet=True,
                                            include_image_size=True, include_image_timestamp=True)
    assert any('width' in r for r in plain)
    for kw in (dict(batch_size=4, loader_workers=3), dict(batch_size=1, loader_workers=2)):
        got = RDB.load_and_run_detector_batch('stub', names, detector=StubDetector(), quiet=True,
                                              use_image_queue=True, use_threads_for_queue=False,
                                              include_image_size=True, include_image_timestamp=True, **kw)
        assert _strip(got) == _strip(plain), kw
    monkeypatch.setattr(RDB, 'ring_slot_bytes', 45 * 64 * 3)        # only the smaller images fit a slot
    got = RDB.load_and_run_detector_batch('stub', names, detector=StubDetector(), quiet=True, batch_size=4,
                                          use_image_queue=True, use_threads_for_queue=False, loader_workers=2,
                                          include_image_size=True, include_image_timestamp=True)
    assert _strip(got) == _strip(plain)


def test_shared_ring_falls_back_to_threads_when_shared_memory_is_unavailable(image_dir, monkeypatch):
    root, names = image_dir
    plain = RDB.load_and_run_detector_batch('stub', names, detector=StubDetector(), quiet=True)

    def no_shm(*a, **k):
        raise OSError(28, 'No space left on device')
    monkeypatch.setattr(RDB.feed, 'ProcessLoader', no_shm)
    got = RDB.load_and_run_detector_batch('stub', names, detector=StubDetector(), quiet=True, batch_size=4,
                                          use_image_queue=True, use_threads_for_queue=False, loader_workers=2)
    assert _strip(got) == _strip(plain)


class PipelinedStub(StubDetector):
    """StubDetector with the start_batch / finish_batch interface of HIPDetector: results must not depend
    on which interface the driver uses, and at most two tickets may be outstanding."""

    def __init__(self):
        super().__init__()
        self.outstanding = 0
        self.max_outstanding = 0

    def start_batch(self, imgs, names, detection_threshold=1e-5, image_size=None, verbose=False):
        self.outstanding += 1
        self.max_outstanding = max(self.max_outstanding, self.outstanding)
        # the pixels must stay valid until finish_batch: keep views, not copies
        return {'imgs': imgs, 'names': names}

    def finish_batch(self, ticket):
        self.outstanding -= 1
        return self.generate_detections_one_batch(ticket['imgs'], ticket['names'])


def test_pipelined_detector_interface_gives_identical_results(image_dir):
    root, names = image_dir
    plain = RDB.load_and_run_detector_batch('stub', names, detector=StubDetector(), quiet=True)
    for kw in (dict(use_threads_for_queue=True), dict(use_threads_for_queue=False)):
        det = PipelinedStub()
        got = RDB.load_and_run_detector_batch('stub', names, detector=det, quiet=True, batch_size=2,
                                              use_image_queue=True, loader_workers=2, **kw)
        assert _strip(got) == _strip(plain), kw
        assert det.outstanding == 0 and 1 <= det.max_outstanding <= 2


def test_threshold_applied_after_batched_detector(image_dir):
    root, names = image_dir
    res = RDB.load_and_run_detector_batch('stub', names, detector=StubDetector(), batch_size=4,
                                          confidence_threshold=0.5, quiet=True)
    for r in res:
        if 'failure' not in r:
            assert all(d['conf'] >= 0.5 for d in r['detections'])


def test_batch_failure_marks_only_that_batch(image_dir):
    root, names = image_dir
    good = [n for n in names if not n.endswith('broken.jpg')]
    det = StubDetector(fail_on={good[5]})
    res = RDB.load_and_run_detector_batch('stub', good, detector=det, batch_size=4, quiet=True)
    failed = [r['file'] for r in res if r.get('failure') == run_detector.FAILURE_INFER]
    assert failed == good[4:8]
    assert all('detections' in r for r in res if r['file'] not in failed)


def test_checkpoint_resume_and_json_format(image_dir, tmp_path):
    root, names = image_dir
    ck = str(tmp_path / 'ck.json')
    first = RDB.load_and_run_detector_batch('stub', names[:6], detector=StubDetector(), checkpoint_path=ck,
                                            checkpoint_frequency=2, quiet=True)
    restored = RDB.load_checkpoint(ck)
    assert [r['file'] for r in restored] == [r['file'] for r in first]
    det = StubDetector()
    full = RDB.load_and_run_detector_batch('stub', names, detector=det, results=restored, quiet=True)
    plain = RDB.load_and_run_detector_batch('stub', names, detector=StubDetector(), quiet=True)
    assert _strip(full) == _strip(plain)
    out = str(tmp_path / 'out' / 'results.json')
    written = RDB.write_results_to_file(full, out, relative_path_base=str(root), detector_file='md_v5a.0.0.pt')
    on_disk = json.load(open(out))
    assert on_disk == json.loads(json.dumps(written))
    assert on_disk['info']['format_version'] == '1.6'
    assert on_disk['info']['detector_metadata']['megadetector_version'] == 'v5a.0.1'
    assert on_disk['detection_categories'] == {'1': 'animal', '2': 'person', '3': 'vehicle'}
    files = [im['file'] for im in on_disk['images']]
    assert files == sorted(files) and not any(f.startswith('/') for f in files)
    for im in on_disk['images']:
        assert 'max_detection_conf' not in im
        if 'failure' in im:
            assert im['detections'] is None
        else:
            confs = [d['conf'] for d in im['detections']]
            assert confs == sorted(confs, reverse=True)


def test_sharding_is_balanced_and_merge_checks(image_dir):
    root, names = image_dir
    shards = RDB.shard_image_list(names, 8)
    assert sorted(sum(shards, [])) == names
    assert max(map(len, shards)) - min(map(len, shards)) <= 1
    res = [RDB.load_and_run_detector_batch('stub', s, detector=StubDetector(), quiet=True) for s in shards]
    merged = RDB.merge_shard_results(res, expected_files=names)
    plain = RDB.load_and_run_detector_batch('stub', names, detector=StubDetector(), quiet=True)
    assert _strip(merged) == _strip(plain)
    with pytest.raises(ValueError, match='duplicate'):
        RDB.merge_shard_results([res[0], res[0]])
    with pytest.raises(ValueError, match='no result'):
        RDB.merge_shard_results(res[:-1], expected_files=names)


def test_model_name_resolution(monkeypatch, tmp_path):
    assert run_detector.get_detector_version_from_filename('/x/md_v5a.0.0.pt') == 'v5a.0.1'
    assert run_detector.get_detector_version_from_filename('whatever.pt') == 'unknown'
    fake = tmp_path / 'md_v5a.0.0.pt'
    fake.write_bytes(b'')
    monkeypatch.setenv('MDV5A', str(fake))
    assert run_detector.try_download_known_detector('MDV5A') == str(fake)
    monkeypatch.delenv('MDV5A')
    with pytest.raises(FileNotFoundError):
        run_detector.try_download_known_detector('MDV5A')
    assert run_detector.try_download_known_detector('/some/file.pt') == '/some/file.pt'
    with pytest.raises(ValueError):
        run_detector.load_detector('model.pb')


# ---------------------------------------------------------------------------------------------
# world_size-2 gloo run of the one-process-per-GPU path (no GPU needed: stub detector)
# ---------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _gloo_worker(rank, world, port, names, out_path):
    import torch.distributed as dist
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    from megadetector_amd import sharded
    from megadetector_amd import run_detector_batch as rdb
    from test_batch_loop import StubDetector as Stub
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:{}'.format(port), rank=rank, world_size=world)
    mine = sharded.my_shard(names, rank, world)
    res = rdb.load_and_run_detector_batch('stub', mine, detector=Stub(), quiet=True, batch_size=3)
    dist.barrier()
    slowest = sharded.max_over_ranks(1.0 + rank, dist)
    merged = sharded.gather_results(res, dist, expected_files=names)
    if rank == 0:
        with open(out_path, 'w') as f:
            json.dump({'merged': merged, 'slowest': slowest}, f)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_run_matches_single_process(image_dir, tmp_path):
    import torch.multiprocessing as mp
    root, names = image_dir
    out_path = str(tmp_path / 'merged.json')
    mp.spawn(_gloo_worker, args=(2, _free_port(), names, out_path), nprocs=2, join=True)
    got = json.load(open(out_path))
    plain = RDB.load_and_run_detector_batch('stub', names, detector=StubDetector(), quiet=True, batch_size=3)
    assert _strip(got['merged']) == _strip(plain)
    assert got['slowest'] == 2.0


def test_load_image_modes_and_exif_rotation(tmp_path):
    """reference visualization_utils.py:103-175: RGBA / L are converted to RGB, EXIF orientations 3 / 6 / 8 are
    applied (180 / 270 / 90 degrees, expand), mirrored orientations are left alone (the reference asserts and
    swallows), unsupported modes raise"""
    from PIL import Image
    from megadetector_amd import feed
    rng = np.random.default_rng(3)
    base = rng.integers(0, 256, (30, 50, 3), dtype=np.uint8)
    Image.fromarray(base).save(tmp_path / 'rgb.png')
    Image.fromarray(np.dstack([base, np.full((30, 50), 200, np.uint8)]), 'RGBA').save(tmp_path / 'rgba.png')
    Image.fromarray(base[..., 0], 'L').save(tmp_path / 'gray.png')
    for name in ('rgb.png', 'rgba.png', 'gray.png'):
        im = feed.load_image(str(tmp_path / name))
        a = np.asarray(im)
        assert im.mode == 'RGB' and a.shape == (30, 50, 3) and a.dtype == np.uint8
    np.testing.assert_array_equal(np.asarray(feed.load_image(str(tmp_path / 'rgb.png'))), base)
    for orientation, shape in ((3, (30, 50)), (6, (50, 30)), (8, (50, 30)), (2, (30, 50))):
        ex = Image.Exif()
        ex[274] = orientation
        p = tmp_path / 'o{}.jpg'.format(orientation)
        Image.fromarray(base).save(p, exif=ex, quality=95)
        im = feed.load_image(str(p))
        assert np.asarray(im).shape[:2] == shape, orientation
        plain = feed.load_image(str(p), ignore_exif_rotation=True)
        assert np.asarray(plain).shape[:2] == (30, 50)
    Image.fromarray(base).convert('CMYK').save(tmp_path / 'cmyk.jpg')
    with pytest.raises(AttributeError, match='unsupported mode'):
        feed.load_image(str(tmp_path / 'cmyk.jpg'))
    meta = feed.image_metadata(feed.load_image(str(tmp_path / 'rgb.png')))
    assert meta['width'] == 50 and meta['height'] == 30 and meta['datetime'] is None


# --------------------------------------------------------------------------------------------
# round 2: sharded checkpoints / resume / liveness, --previous_results_file, option forwarding
# --------------------------------------------------------------------------------------------
def _stub_shard_worker(gpu, model_file, files, kwargs, out_q):
    """_shard_worker with the stub detector instead of a GPU (same checkpoint-path rule)"""
    try:
        kwargs = dict(kwargs)
        kwargs.pop('detector_options', None)
        kwargs['checkpoint_path'] = RDB.shard_checkpoint_path(kwargs.get('checkpoint_path'), gpu)
        res = RDB.load_and_run_detector_batch(model_file, files, detector=StubDetector(), **kwargs)
        out_q.put((gpu, res, None))
    except Exception as e:
        out_q.put((gpu, None, repr(e)))


def _dying_shard_worker(gpu, model_file, files, kwargs, out_q):
    if gpu == 1:
        os._exit(3)                # what a HIP fault or the OOM killer looks like from the parent
    _stub_shard_worker(gpu, model_file, files, kwargs, out_q)


def test_sharded_run_writes_one_checkpoint_per_shard_and_resumes(image_dir, tmp_path):
    root, names = image_dir
    ck = str(tmp_path / 'ck.json')
    plain = RDB.load_and_run_detector_batch('stub', names, detector=StubDetector(), quiet=True)
    part = RDB.run_sharded('stub', names[:8], 2, worker=_stub_shard_worker, checkpoint_path=ck,
                           checkpoint_frequency=2, quiet=True)
    assert sorted(r['file'] for r in part) == names[:8]
    assert not os.path.exists(ck)
    shard_files = [RDB.shard_checkpoint_path(ck, g) for g in range(2)]
    assert all(os.path.isfile(p) for p in shard_files)
    per_shard = [set(r['file'] for r in RDB.load_checkpoint(p)) for p in shard_files]
    assert per_shard[0] == set(names[0:8:2]) and per_shard[1] == set(names[1:8:2])     # no shard saw the other's files
    restored = RDB.load_sharded_checkpoints(ck, 2)
    assert sorted(r['file'] for r in restored) == names[:8]
    # resume: only the files without a result are sharded again
    full = RDB.run_sharded('stub', names, 2, results=restored, worker=_stub_shard_worker, quiet=True)
    assert _strip(full) == _strip(plain)
    # (a resumed result is kept as it is, not recomputed)
    marked = [dict(r, marker=1) for r in restored]
    full = RDB.run_sharded('stub', names, 2, results=marked, worker=_stub_shard_worker, quiet=True)
    assert sum(1 for r in full if r.get('marker') == 1) == 8
    # a resumed run with the same checkpoint path overwrites the shard files of the first run; what it
    # restored must survive a SECOND crash (kept in the plain file, which load_sharded_checkpoints unions)
    second = RDB.run_sharded('stub', names[:10], 2, results=restored, worker=_stub_shard_worker,
                             checkpoint_path=ck, checkpoint_frequency=1, quiet=True)
    assert len(second) == 10
    after_second = RDB.load_sharded_checkpoints(ck, 2)
    assert set(r['file'] for r in after_second) == set(names[:10])
    assert len(after_second) == 10                                   # no duplicates in the union


def test_dead_shard_raises_instead_of_hanging(image_dir):
    root, names = image_dir
    import time
    t0 = time.time()
    with pytest.raises(RuntimeError, match='exited with code 3'):
        RDB.run_sharded('stub', names, 2, worker=_dying_shard_worker, quiet=True)
    assert time.time() - t0 < 60


def test_previous_results_file_merge(image_dir, tmp_path, monkeypatch):
    """reference run_detector_batch.py:2056-2096, :2166-2171"""
    root, names = image_dir
    monkeypatch.setattr(RDB.run_detector, 'load_detector', lambda *a, **k: StubDetector())
    first = str(tmp_path / 'first.json')
    sub = [n for n in names if '/cam0/' in n]
    lst = str(tmp_path / 'list.json')
    json.dump(sub, open(lst, 'w'))
    # a first pass over cam0 only, written with paths relative to the folder
    res = RDB.load_and_run_detector_batch('stub', sub, detector=StubDetector(), quiet=True)
    RDB.write_results_to_file(res, first, relative_path_base=str(root), detector_file='md_v5a.0.0.pt')
    for im in json.load(open(first))['images']:
        assert not os.path.isabs(im['file'])
    seen = []
    orig = RDB.load_and_run_detector_batch

    def spy(model_file, image_file_names, **kw):
        seen.extend(image_file_names)
        return orig(model_file, image_file_names, **kw)
    monkeypatch.setattr(RDB, 'load_and_run_detector_batch', spy)
    out = str(tmp_path / 'second.json')
    RDB.main(['stub', str(root), out, '--output_relative_filenames', '--previous_results_file', first, '--quiet'])
    assert sorted(seen) == sorted(n for n in names if n not in sub)           # cam0 was not processed again
    merged = json.load(open(out))
    monkeypatch.setattr(RDB, 'load_and_run_detector_batch', orig)
    plain = str(tmp_path / 'plain.json')
    RDB.main(['stub', str(root), plain, '--output_relative_filenames', '--quiet'])
    want = json.load(open(plain))
    assert merged['images'] == want['images']
    with pytest.raises(AssertionError, match='relative paths'):
        RDB.main(['stub', str(root), out, '--previous_results_file', first])
    RDB.main(['stub', str(root), out, '--output_relative_filenames', '--overwrite_handling', 'skip'])
    with pytest.raises(Exception, match='exists'):
        RDB.main(['stub', str(root), out, '--output_relative_filenames', '--overwrite_handling', 'error'])


class _ModeRecordingStub(StubDetector):
    compatibility_mode = 'modern'

    def __init__(self):
        super().__init__()
        self.seen = []

    def generate_detections_one_batch(self, imgs, names, **kw):
        self.seen.extend(imgs)
        return super().generate_detections_one_batch(imgs, names, **kw)


def test_preprocess_queue_letterboxes_in_the_detectors_compatibility_mode(image_dir):
    """the producers' preprocessor is built from the consumer's options (reference _producer_func :143-152): under
    compatibility_mode=modern the preprocessed dicts carry the modern geometry ('resized_shape'), not the classic one"""
    root, names = image_dir
    from megadetector_amd.detector import HIPDetector
    for mode in ('modern', 'classic'):
        det = _ModeRecordingStub()
        det.compatibility_mode = mode
        RDB.load_and_run_detector_batch('stub', names, detector=det, quiet=True, use_image_queue=True, batch_size=3,
                                        preprocess_on_image_queue=True, loader_workers=2,
                                        detector_options={'compatibility_mode': mode})
        assert det.seen and all(isinstance(d, dict) for d in det.seen)
        want = HIPDetector('synthetic', {'preprocess_only': True, 'compatibility_mode': mode})
        for d in det.seen:
            ref = want.preprocess_image(d['img_original'], image_id=d['file'])
            assert ('resized_shape' in d) == (mode == 'modern') == ('resized_shape' in ref)
            assert d['img_processed'].geometry == ref['img_processed'].geometry
            assert d['target_shape'] == ref['target_shape']


def test_user_image_size_reaches_the_device_capacity(image_dir, monkeypatch):
    root, names = image_dir
    got = {}

    def fake_load(model_file, force_model_download=False, detector_options=None, verbose=False):
        got.update(detector_options or {})
        return StubDetector()
    monkeypatch.setattr(RDB.run_detector, 'load_detector', fake_load)
    RDB.load_and_run_detector_batch('stub', names[:2], image_size=1600, quiet=True)
    assert got.get('max_image_size') == 1600
    got.clear()
    RDB.load_and_run_detector_batch('stub', names[:2], image_size=640, quiet=True)
    assert 'max_image_size' not in got
