"""
ctypes binding of libmdhip.so (C ABI: include/mdhip.h).

There is NO CPU fallback: if the shared library is missing or no HIP device is visible the
product path raises.  (The CPU restatement under oracle/ is test infrastructure only.)
"""

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libmdhip.so')

MDHIP_OK = 0
MDHIP_DTYPE_BF16 = 0
MDHIP_DTYPE_FP8 = 1
MDHIP_DTYPE_FP16 = 2


class mdhip_conv(C.Structure):
    _fields_ = [('weight', C.POINTER(C.c_float)), ('bias', C.POINTER(C.c_float)),
                ('c_out', C.c_int32), ('c_in', C.c_int32), ('kh', C.c_int32), ('kw', C.c_int32)]


class mdhip_layer(C.Structure):
    _fields_ = [('type', C.c_int32), ('n_from', C.c_int32), ('from_', C.c_int32 * 4),
                ('c_out', C.c_int32), ('k', C.c_int32), ('s', C.c_int32), ('p', C.c_int32),
                ('n', C.c_int32), ('shortcut', C.c_int32), ('first_conv', C.c_int32)]


class mdhip_model(C.Structure):
    _fields_ = [('n_layers', C.c_int32), ('layers', C.POINTER(mdhip_layer)),
                ('n_convs', C.c_int32), ('convs', C.POINTER(mdhip_conv)),
                ('nc', C.c_int32), ('na', C.c_int32), ('nl', C.c_int32),
                ('anchors_px', C.POINTER(C.c_float)), ('strides', C.POINTER(C.c_float))]


class mdhip_letterbox(C.Structure):
    _fields_ = [('src_h', C.c_int32), ('src_w', C.c_int32), ('resized_h', C.c_int32),
                ('resized_w', C.c_int32), ('top', C.c_int32), ('left', C.c_int32), ('interp', C.c_int32)]


class mdhip_op_info(C.Structure):
    _fields_ = [('name', C.c_char * 48), ('kind', C.c_int32), ('layer', C.c_int32),
                ('m', C.c_int32), ('n', C.c_int32), ('k', C.c_int32),
                ('flops', C.c_double), ('bytes', C.c_double), ('cfg', C.c_int32),
                ('ntaps', C.c_int32), ('stride', C.c_int32), ('has_res', C.c_int32)]


class mdhip_tuned(C.Structure):
    _fields_ = [('m', C.c_int32), ('n', C.c_int32), ('k', C.c_int32), ('ntaps', C.c_int32),
                ('stride', C.c_int32), ('has_res', C.c_int32), ('cfg', C.c_int32), ('batch', C.c_int32)]


#: every symbol include/mdhip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    'mdhip_create': (C.c_int, [C.POINTER(mdhip_model), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    'mdhip_destroy': (None, [_P]),
    'mdhip_last_error': (C.c_char_p, [_P]),
    'mdhip_preprocess': (C.c_int, [_P, C.POINTER(_P), C.POINTER(mdhip_letterbox), C.c_int, C.c_int, C.c_int, _P]),
    'mdhip_forward': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P]),
    'mdhip_forward_tta': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P]),
    'mdhip_last_num_anchors': (C.c_int, [_P]),
    'mdhip_calibrate': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P]),
    'mdhip_fp8_num_tensors': (C.c_int, [_P]),
    'mdhip_fp8_get_scales': (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int]),
    'mdhip_fp8_set_scales': (C.c_int, [_P, C.POINTER(C.c_float), C.c_int]),
    'mdhip_f32_to_e4m3': (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.c_int]),
    'mdhip_nms': (C.c_int, [_P, C.c_int, C.c_float, C.c_float, C.c_int, _P, _P, _P]),
    'mdhip_nms_enqueue': (C.c_int, [_P, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, _P]),
    'mdhip_nms_wait': (C.c_int, [_P, C.c_int, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_int32))]),
    'mdhip_nms_on': (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, _P, _P, _P]),
    'mdhip_num_anchors': (C.c_int, [_P, C.c_int, C.c_int]),
    'mdhip_max_stride': (C.c_int, [_P]),
    'mdhip_read_predictions': (C.c_int, [_P, C.c_int, _P, _P]),
    'mdhip_read_input': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P]),
    'mdhip_read_layer': (C.c_int, [_P, C.c_int, C.c_int, _P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), _P]),
    'mdhip_num_ops': (C.c_int, [_P]),
    'mdhip_get_op_info': (C.c_int, [_P, C.c_int, C.POINTER(mdhip_op_info)]),
    'mdhip_forward_timed': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P]),
    'mdhip_time_forwards': (C.c_int, [_P, C.c_int]),
    'mdhip_forward_times': (C.c_int, [_P, _P, C.c_int]),
    'mdhip_set_op_cfg': (C.c_int, [_P, C.c_int, C.c_int]),
    'mdhip_num_conv_cfgs': (C.c_int, []),
    'mdhip_op_supports_cfg': (C.c_int, [_P, C.c_int, C.c_int]),
    'mdhip_cfg_is_bitwise': (C.c_int, [C.c_int]),
    'mdhip_conv_cfg_name': (C.c_char_p, [C.c_int]),
    'mdhip_set_tuned': (C.c_int, [_P, C.POINTER(mdhip_tuned), C.c_int]),
    'mdhip_set_fuse': (C.c_int, [_P, C.c_int]),
    'mdhip_set_graph': (C.c_int, [_P, C.c_int, C.c_int]),
    'mdhip_set_option': (C.c_int, [_P, C.c_char_p, C.c_int]),
    'mdhip_time_op': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), _P]),
    'mdhip_version': (C.c_char_p, []),
}

_lib = None


def load():
    """Loads libmdhip.so (once).  Raises RuntimeError when the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'libmdhip.so not found at {}: build it with `python -c "import __graft_entry__ as g; '
            'g.build()"` or `make -C megadetector_amd/csrc`. There is no CPU fallback.'.format(LIB_PATH))
    # torch bundles its own libamdhip64.so.7; importing it first makes the process use ONE HIP
    # runtime (ours resolves the already-loaded SONAME).
    if os.environ.get('MDHIP_NO_TORCH_PRELOAD', '0') != '1':
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class HipError(RuntimeError):
    pass
