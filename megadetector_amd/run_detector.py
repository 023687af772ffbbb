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

# reference run_detector.py:88-137 (order matters: first match wins)
model_string_to_model_version = {
    'mdv5a': 'v5a.0.1', 'mdv5b': 'v5b.0.1',
    'v5a.0.0': 'v5a.0.1', 'v5b.0.0': 'v5b.0.1', 'v5a.0.1': 'v5a.0.1', 'v5b.0.1': 'v5b.0.1',
    'md1000-redwood': 'v1000.0.0-redwood', 'md1000-spruce': 'v1000.0.0-spruce',
    'mdv1000-redwood': 'v1000.0.0-redwood', 'mdv1000-spruce': 'v1000.0.0-spruce',
    'v1000-redwood': 'v1000.0.0-redwood', 'v1000-spruce': 'v1000.0.0-spruce',
    'redwood': 'v1000.0.0-redwood', 'spruce': 'v1000.0.0-spruce',
    'mdv5': 'v5a.0.1', 'md5': 'v5a.0.1', 'mdv1000': 'v1000.0.0-redwood', 'md1000': 'v1000.0.0-redwood',
    'default': 'v5a.0.1', 'megadetector': 'v5a.0.1',
}

# the YOLOv5-family entries of reference run_detector.py:152-248 (what write_results_to_file
# copies into info.detector_metadata)
known_models = {
    'v5a.0.0': {'typical_detection_threshold': 0.2, 'conservative_detection_threshold': 0.05,
                'image_size': 1280, 'model_type': 'yolov5', 'md5': 'ec1d7603ec8cf642d6e0cd008ba2be8c'},
    'v5b.0.0': {'typical_detection_threshold': 0.2, 'conservative_detection_threshold': 0.05,
                'image_size': 1280, 'model_type': 'yolov5', 'md5': 'bc235e73f53c5c95e66ea0d1b2cbf542'},
    'v5a.0.1': {'typical_detection_threshold': 0.2, 'conservative_detection_threshold': 0.05,
                'image_size': 1280, 'model_type': 'yolov5', 'md5': '60f8e7ec1308554df258ed1f4040bc4f'},
    'v5b.0.1': {'typical_detection_threshold': 0.2, 'conservative_detection_threshold': 0.05,
                'image_size': 1280, 'model_type': 'yolov5', 'md5': 'f17ed6fedfac2e403606a08c89984905'},
    'v1000.0.0-redwood': {'typical_detection_threshold': 0.3, 'md5': '74474b3aec9cf1a990da38b37ddf9197'},
    'v1000.0.0-spruce': {'md5': '1c9d1d2b3ba54931881471fdd508e6f2'},
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
