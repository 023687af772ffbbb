"""
Generates tests/golden/host_path_reference.json by running the *real* host path of the reference
(/root/reference/megadetector/detection/run_detector_batch.py and run_detector.py, read-only) in the build
container.  The reference cannot travel to the GPU box, so what it emits is committed as a fixture together with this
script; tests/test_host_path_reference.py asserts that megadetector_amd reproduces it byte for byte.

What runs from the reference itself:
  * write_results_to_file (:1546-1662) -> the output FILE TEXT (known model, unknown model, failures, '\\' paths,
    include_max_conf, relative_path_base, custom_metadata, caller-supplied info)
  * write_checkpoint / load_checkpoint (:1465-1520) -> file text, round trip
  * _group_into_batches (:657)
  * load_and_run_detector_batch (:1062-1439): the plain loop, the batched loop, both with checkpoints, a resumed run,
    an unreadable image, a failing batch -- driven by the deterministic stub detector of tests/stub_detector.py
    (load_detector / try_download_known_detector / is_gpu_available of the reference module are bound to the stub,
    nothing else is touched)
  * run_detector.get_detector_version_from_filename / get_detector_metadata_from_version_string (:276-350) and the
    known_models / model_string_to_model_version tables (:88-248)

Third-party modules absent from this container are stubbed as empty modules: cv2 (attribute access returns an int),
jsonpickle, humanfriendly (format_timespan only), torchvision.  None of them computes anything on this path.

Run (build container only):  python tests/golden/gen_host_golden_from_reference.py
"""

import copy
import io
import json
import os
import re
import sys
import tempfile
import types
from contextlib import redirect_stdout

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
sys.path.insert(0, '/root/reference')

from stub_detector import StubDetector, write_test_images, IMAGE_SPECS, sample_results  # noqa: E402

TIME_RE = re.compile(r'("detection_completion_time": )"[^"]*"')
TIME_PLACEHOLDER = r'\1"<time>"'


class _AnyAttr(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return 0


def import_reference():
    sys.modules.setdefault('cv2', _AnyAttr('cv2'))
    sys.modules.setdefault('jsonpickle', types.ModuleType('jsonpickle'))
    hf = types.ModuleType('humanfriendly')
    hf.format_timespan = lambda s: '{:.2f} seconds'.format(s)
    sys.modules.setdefault('humanfriendly', hf)
    tv = types.ModuleType('torchvision')
    tv.ops = types.ModuleType('torchvision.ops')
    sys.modules.setdefault('torchvision', tv)
    sys.modules.setdefault('torchvision.ops', tv.ops)
    import megadetector.detection.run_detector as rd
    import megadetector.detection.run_detector_batch as rdb
    return rd, rdb


def _read(path):
    with open(path, 'r', encoding='utf-8') as f:
        return TIME_RE.sub(TIME_PLACEHOLDER, f.read())


def writer_cases(rdb, tmp):
    base = sample_results()
    cases = [
        dict(name='known_model_v5a', kwargs=dict(detector_file='/models/md_v5a.0.0.pt')),
        dict(name='known_model_v5b_name', kwargs=dict(detector_file='MDV5B')),
        dict(name='known_model_redwood', kwargs=dict(detector_file='md_v1000.0.0-redwood.pt')),
        dict(name='known_model_spruce', kwargs=dict(detector_file='/x/md_v1000.0.0-spruce.pt')),
        dict(name='unknown_model', kwargs=dict(detector_file='/models/my_finetuned_yolo.pt')),
        dict(name='no_detector_file', kwargs=dict()),
        dict(name='include_max_conf', kwargs=dict(detector_file='md_v5a.0.0.pt', include_max_conf=True)),
        dict(name='relative_paths', kwargs=dict(detector_file='md_v5a.0.0.pt', relative_path_base='/data/cam')),
        dict(name='keep_backslashes', kwargs=dict(detector_file='md_v5a.0.0.pt', force_forward_slashes=False)),
        dict(name='custom_metadata', kwargs=dict(detector_file='md_v5a.0.0.pt',
                                                 custom_metadata={'site': 'A1', 'n': 3})),
        dict(name='caller_info', kwargs=dict(info={'format_version': '1.6', 'detector': 'x', 'note': 'given'})),
    ]
    for c in cases:
        out = os.path.join(tmp, 'w_{}.json'.format(c['name']))
        results = copy.deepcopy(base)
        with redirect_stdout(io.StringIO()):
            rdb.write_results_to_file(results, out, **copy.deepcopy(c['kwargs']))
        c['output_text'] = _read(out)
    return cases


def checkpoint_cases(rdb, tmp):
    base = sample_results()
    path = os.path.join(tmp, 'ckpt.json')
    texts = []
    with redirect_stdout(io.StringIO()):
        rdb.write_checkpoint(path, copy.deepcopy(base[:2]))
        texts.append(_read(path))
        rdb.write_checkpoint(path, copy.deepcopy(base))        # the overwrite path (backup + remove)
        texts.append(_read(path))
        loaded = rdb.load_checkpoint(path)
    return {'texts': texts, 'loaded': loaded, 'tmp_left_behind': os.path.exists(path + '_tmp')}


def loop_cases(rd, rdb, tmp):
    names = write_test_images(tmp)
    bad = os.path.join(tmp, 'broken.jpg')
    with open(bad, 'wb') as f:
        f.write(b'this is not a jpeg')
    cwd = os.getcwd()
    os.chdir(tmp)
    rel = [os.path.basename(n) for n in names]
    cases = []
    try:
        def run(case_name, files, fail_on=None, results=None, **kw):
            det = StubDetector(fail_on=set(fail_on or ()))
            rdb.load_detector = lambda *a, **k: det
            rdb.try_download_known_detector = lambda m, **k: m
            rdb.is_gpu_available = lambda *a, **k: True
            ck = kw.get('checkpoint_path')
            if ck and os.path.exists(ck) and results is None:
                os.remove(ck)
            with redirect_stdout(io.StringIO()):
                res = rdb.load_and_run_detector_batch('stub.pt', list(files), results=copy.deepcopy(results),
                                                      quiet=True, **kw)
            case = {'name': case_name, 'files': list(files), 'fail_on': sorted(fail_on or ()), 'kwargs': kw,
                    'results_in': results, 'results': json.loads(json.dumps(res, default=str)),
                    'batches': det.batches}
            if ck:
                case['checkpoint_text'] = _read(ck) if os.path.exists(ck) else None
            out = 'out_{}.json'.format(case_name)
            with redirect_stdout(io.StringIO()):
                rdb.write_results_to_file(copy.deepcopy(res), out, detector_file='md_v5a.0.0.pt')
            case['output_text'] = _read(out)
            cases.append(case)
            return res

        run('plain', rel)
        run('plain_threshold', rel, confidence_threshold=0.4)
        run('plain_with_unreadable', rel[:5] + ['broken.jpg'] + rel[5:])
        run('plain_size_timestamp', rel, include_image_size=True, include_image_timestamp=True)
        run('plain_checkpoints', rel, checkpoint_path='ck_plain.json', checkpoint_frequency=4)
        run('batched', rel, batch_size=4)
        run('batched_threshold', rel, batch_size=4, confidence_threshold=0.4)
        run('batched_ragged_with_unreadable', rel[:5] + ['broken.jpg'] + rel[5:], batch_size=3)
        run('batched_size_timestamp', rel, batch_size=5, include_image_size=True, include_image_timestamp=True)
        run('batched_failing_batch', rel, batch_size=4, fail_on=[rel[5]])
        run('batched_checkpoints_multiple', rel, batch_size=3, checkpoint_path='ck_b3.json', checkpoint_frequency=6)
        run('batched_checkpoints_never_aligned', rel[:10], batch_size=4, checkpoint_path='ck_b4.json',
            checkpoint_frequency=3)
        first = run('resume_part1', rel[:6], batch_size=2)
        run('resume_part2', rel, batch_size=2, results=json.loads(json.dumps(first)))
    finally:
        os.chdir(cwd)
    return cases


def main():
    rd, rdb = import_reference()
    fixture = {'image_specs': IMAGE_SPECS}
    with tempfile.TemporaryDirectory() as tmp:
        fixture['writer_cases'] = writer_cases(rdb, tmp)
        fixture['checkpoint'] = checkpoint_cases(rdb, tmp)
        fixture['loop_cases'] = loop_cases(rd, rdb, tmp)
    fixture['group_into_batches'] = [
        {'n': n, 'batch_size': bs, 'out': rdb._group_into_batches(list(range(n)), bs)}
        for n, bs in ((0, 4), (1, 4), (4, 4), (9, 4), (10, 1), (7, 32))]
    names = ['md_v5a.0.0.pt', 'md_v5b.0.0.pt', 'MD_V5A.0.1.PT', '/a/b/md_v5b.0.1.pt', 'mdv5a', 'MDV5B', 'megadetector',
             'default', 'md_v1000.0.0-redwood.pt', 'md_v1000.0.0-spruce.pt', 'md_v1000.0.0-larch.pt',
             'md_v1000.0.0-cedar.pt', 'md_v1000.0.0-sorrel.pt', 'my_model.pt', 'md_v4.1.0.pb', 'megadetector_v3.pb',
             'mdv5-custom.pt', 'redwood-finetune.pt', 'c:\\models\\md_v5a.0.0.pt']
    with redirect_stdout(io.StringIO()):
        fixture['version_from_filename'] = {n: rd.get_detector_version_from_filename(n) for n in names}
        fixture['version_from_filename_strict'] = {
            n: rd.get_detector_version_from_filename(n, accept_first_match=False) for n in names}
        versions = sorted(set(fixture['version_from_filename'].values()) | set(rd.known_models) | {'unknown'})
        # json text, not a dict: the key ORDER is part of what the writer emits
        fixture['metadata_text'] = {
            v: json.dumps(rd.get_detector_metadata_from_version_string(v), indent=1)
            for v in versions if isinstance(v, str)}
    fixture['model_string_to_model_version'] = list(rd.model_string_to_model_version.items())
    fixture['constants'] = {
        'current_format_version': rdb.current_format_version,
        'default_loaders': rdb.default_loaders,
        'max_queue_size': rdb.max_queue_size,
        'DEFAULT_OUTPUT_CONFIDENCE_THRESHOLD': rd.DEFAULT_OUTPUT_CONFIDENCE_THRESHOLD,
        'DEFAULT_DETECTOR_LABEL_MAP': rd.DEFAULT_DETECTOR_LABEL_MAP,
        'FAILURE_INFER': rd.FAILURE_INFER, 'FAILURE_IMAGE_OPEN': rd.FAILURE_IMAGE_OPEN,
        'CONF_DIGITS': rd.CONF_DIGITS, 'COORD_DIGITS': rd.COORD_DIGITS,
    }
    out = os.path.join(REPO, 'tests', 'golden', 'host_path_reference.json')
    with open(out, 'w') as f:
        json.dump(fixture, f, indent=0)
    print('wrote', out, os.path.getsize(out), 'bytes;', len(fixture['writer_cases']), 'writer cases,',
          len(fixture['loop_cases']), 'loop cases')


if __name__ == '__main__':
    main()
