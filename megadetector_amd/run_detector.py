"""
Detector factory and model bookkeeping: the plugin seam of the reference
(megadetector/detection/run_detector.py) with the HIP detector behind it.

  load_detector(model_file, force_cpu, force_model_download, detector_options, verbose)
      reference run_detector.py:601-681 -- dispatches on the file extension; '.pt' models (and the
      'synthetic...' pseudo-model used when no checkpoint is available offline) are served by
      megadetector_amd.detector.HIPDetector.  TensorFlow (.pb) and RF-DETR (.pth) models are other
      model families and out of scope.
  try_download_known_detector(...)   reference :1051-1091 -- resolves "MDV5A"-style names through
      the environment variable of the same name; this build never touches the network.
"""

import os

from .constants import (FAILURE_INFER, FAILURE_IMAGE_OPEN, CONF_DIGITS, COORD_DIGITS,          # noqa: F401
                        DEFAULT_DETECTOR_LABEL_MAP, DEFAULT_OUTPUT_CONFIDENCE_THRESHOLD)

# reference run_detector.py:88-137: filename substrings -> canonical version.  Order matters (the first matching
# key wins), so the table is generated in the reference's order: exact versions, the three spellings of the MDv1000
# family, bare tree names (no "sorrel": the reference has none), opinionated defaults.
_V1000 = ('redwood', 'cedar', 'larch', 'sorrel', 'spruce')
model_string_to_model_version = {}
model_string_to_model_version.update({'mdv2': 'v2.0.0', 'mdv3': 'v3.0.0', 'mdv4': 'v4.1.0',
                                      'mdv5a': 'v5a.0.1', 'mdv5b': 'v5b.0.1',
                                      'v2': 'v2.0.0', 'v3': 'v3.0.0', 'v4': 'v4.1.0', 'v4.1': 'v4.1.0',
                                      'v5a.0.0': 'v5a.0.1', 'v5b.0.0': 'v5b.0.1',
                                      'v5a.0.1': 'v5a.0.1', 'v5b.0.1': 'v5b.0.1'})
for _prefix in ('md1000-', 'mdv1000-', 'v1000-'):
    model_string_to_model_version.update({_prefix + t: 'v1000.0.0-' + t for t in _V1000})
model_string_to_model_version.update({t: 'v1000.0.0-' + t for t in ('redwood', 'spruce', 'cedar', 'larch')})
model_string_to_model_version.update({'mdv5': 'v5a.0.1', 'md5': 'v5a.0.1', 'mdv1000': 'v1000.0.0-redwood',
                                      'md1000': 'v1000.0.0-redwood', 'default': 'v5a.0.1',
                                      'megadetector': 'v5a.0.1'})

# reference run_detector.py:140-248: what write_results_to_file copies -- key for key, in this key order -- into
# info.detector_metadata (pinned against the real table by tests/test_host_path_reference.py)
_GH = 'https://github.com/agentmorris/MegaDetector/releases/download/'
_LILA = 'https://lila.science/public/models/megadetector/'
model_url_base = os.environ.get('MD_MODEL_URL_BASE') or (_GH + 'v1000.0/')
if not model_url_base.endswith('/'):
    model_url_base += '/'


def _tf(url):
    return {'url': url, 'typical_detection_threshold': 0.8, 'conservative_detection_threshold': 0.3,
            'model_type': 'tf', 'normalized_typical_inference_speed': 1.0 / 3.5}


def _v5(name, md5):
    return {'url': _GH + 'v5.0/' + name, 'typical_detection_threshold': 0.2, 'conservative_detection_threshold': 0.05,
            'image_size': 1280, 'model_type': 'yolov5', 'normalized_typical_inference_speed': 1.0, 'md5': md5}


def _v1000(tree, speed, md5, **extra):
    d = {'url': model_url_base + 'md_v1000.0.0-{}.pt'.format(tree), 'normalized_typical_inference_speed': speed,
         'md5': md5}
    d.update(extra)
    return d


known_models = {
    'v2.0.0': _tf(_LILA + 'megadetector_v2.pb'),
    'v3.0.0': _tf(_LILA + 'megadetector_v3.pb'),
    'v4.1.0': _tf(_GH + 'v4.1/md_v4.1.0.pb'),
    'v5a.0.0': _v5('md_v5a.0.0.pt', 'ec1d7603ec8cf642d6e0cd008ba2be8c'),
    'v5b.0.0': _v5('md_v5b.0.0.pt', 'bc235e73f53c5c95e66ea0d1b2cbf542'),
    'v5a.0.1': _v5('md_v5a.0.1.pt', '60f8e7ec1308554df258ed1f4040bc4f'),
    'v5b.0.1': _v5('md_v5b.0.1.pt', 'f17ed6fedfac2e403606a08c89984905'),
    'v1000.0.0-redwood': _v1000('redwood', 1.0, '74474b3aec9cf1a990da38b37ddf9197', typical_detection_threshold=0.3),
    'v1000.0.0-spruce': _v1000('spruce', 12.7, '1c9d1d2b3ba54931881471fdd508e6f2'),
    'v1000.0.0-larch': _v1000('larch', 2.4, 'cab94ebd190c2278e12fb70ffd548b6d'),
    'v1000.0.0-cedar': _v1000('cedar', 2.0, '3d6472c9b95ba687b59ebe255f7c576b'),
    'v1000.0.0-sorrel': _v1000('sorrel', 7.0, '4339a2c8af7a381f18ded7ac2a4df03e'),
}


def get_detector_version_from_filename(detector_filename, accept_first_match=True, verbose=False):
    """reference run_detector.py:303-350"""
    fn = os.path.basename(detector_filename).lower()
    matches = [s for s in model_string_to_model_version if s in fn]
    if not matches:
        return 'unknown'
    if len(matches) > 1 and not accept_first_match:
        return 'multiple'
    return model_string_to_model_version[matches[0]]


def get_detector_metadata_from_version_string(detector_version):
    """reference run_detector.py:276-300"""
    if detector_version not in known_models:
        print('Warning: no metadata for unknown detector version {}'.format(detector_version))
        return {'megadetector_version': 'unknown', 'typical_detection_threshold': 0.2,
                'conservative_detection_threshold': 0.1}
    meta = dict(known_models[detector_version])
    meta['megadetector_version'] = detector_version
    return meta


def is_gpu_available(model_file=None):
    """reference run_detector.py:554-598 (ROCm torch reports HIP devices through torch.cuda)"""
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return False


def try_download_known_detector(detector_file, force_download=False, verbose=False):
    """
    reference run_detector.py:1051-1091.  A known model name is resolved through the environment
    variable of the same name (`MDV5A=/path/md_v5a.0.0.pt`, reference :1083-1087); there is no
    network here, so an unresolved name is an error instead of a download.
    """
    model_string = detector_file.lower()
    if model_string in model_string_to_model_version:
        model_string = model_string_to_model_version[model_string]
    if model_string in known_models:
        if detector_file in os.environ:
            fn = os.environ[detector_file]
            print('Reading MD location from environment variable {}: {}'.format(detector_file, fn))
            return fn
        raise FileNotFoundError(
            'model name "{0}" needs a download, which this offline build does not do: point the '
            'environment variable {0} at the checkpoint file'.format(detector_file))
    return detector_file


def load_detector(model_file, force_cpu=False, force_model_download=False, detector_options=None,
                  verbose=False):
    """reference run_detector.py:601-681"""
    from .detector import HIPDetector
    if isinstance(model_file, str) and not model_file.startswith('synthetic'):
        model_file = try_download_known_detector(model_file, force_download=force_model_download,
                                                 verbose=verbose)
    if verbose:
        print('GPU available: {}'.format(is_gpu_available(model_file)))
    opts = dict(detector_options or {})
    if 'force_cpu' not in opts:
        opts['force_cpu'] = force_cpu
    name = model_file if isinstance(model_file, str) else 'weights-object'
    if isinstance(model_file, str) and not (name.endswith('.pt') or name.startswith('synthetic')):
        if name.endswith('.pb') or name.endswith('.pth'):
            raise ValueError('{}: TensorFlow (.pb) and RF-DETR (.pth) models are not part of the HIP '
                             'hot path; use the reference implementation for them'.format(name))
        raise ValueError('Unrecognized model format: {}'.format(name))
    return HIPDetector(model_file, detector_options=opts, verbose=verbose)
