"""Constants shared with MegaDetector's output format (reference run_detector.py:55-67,:251)."""

# reference megadetector/detection/run_detector.py:55-56
FAILURE_INFER = 'inference failure'
FAILURE_IMAGE_OPEN = 'image access failure'

# reference run_detector.py:59-60
CONF_DIGITS = 3
COORD_DIGITS = 4

# reference run_detector.py:63-67
DEFAULT_DETECTOR_LABEL_MAP = {'1': 'animal', '2': 'person', '3': 'vehicle'}

# reference run_detector.py:251
DEFAULT_OUTPUT_CONFIDENCE_THRESHOLD = 0.005

# reference pytorch_detector.py:733
DEFAULT_COMPATIBILITY_MODE = 'classic'
