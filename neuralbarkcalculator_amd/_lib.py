"""ctypes binding of libnbc_hip.so (C ABI declared in include/nbc.h).

The library is the product: there is no Python/torch fallback.  ``load()`` raises if the
shared object has not been built (``python -m neuralbarkcalculator_amd.build``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libnbc_hip.so")

NBC_OK = 0
NBC_ERR_INVALID, NBC_ERR_KEYS, NBC_ERR_HIP, NBC_ERR_STATE, NBC_ERR_NOMEM = -1, -2, -3, -4, -5
PREC_FP32, PREC_BF16, PREC_F16X2 = 0, 1, 2
IN_F32_NCHW, IN_U8_NHWC = 0, 1
LABEL_U8, LABEL_I64 = 0, 1
PACK_ROW_CLAMPED, PACK_SCALE_RANGE = 1, 2      # NBC_PACK_* of include/nbc.h


class NbcTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("shape", C.c_int64 * 4),
                ("ndim", C.c_int32), ("dtype", C.c_int32)]


class NbcConvDesc(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("bn", C.c_char * 64),
                ("cin", C.c_int32), ("cout", C.c_int32), ("k", C.c_int32), ("stride", C.c_int32),
                ("pad", C.c_int32), ("dil", C.c_int32),
                ("relu", C.c_int32), ("bias", C.c_int32), ("residual", C.c_int32)]


class NbcOpRecord(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("kernel", C.c_char * 32), ("ms", C.c_float), ("calls", C.c_int32),
                ("flops", C.c_double), ("bytes", C.c_double), ("kh", C.c_int32), ("kw", C.c_int32),
                ("cout", C.c_int32), ("launches", C.c_int32)]


# every symbol include/nbc.h declares: (restype, argtypes)
SIGNATURES = {
    "nbc_last_error": (C.c_char_p, []),
    "nbc_version": (C.c_char_p, []),
    "nbc_num_convs": (C.c_int, []),
    "nbc_conv_info": (C.c_int, [C.c_int, C.POINTER(NbcConvDesc)]),
    "nbc_num_state_keys": (C.c_int, []),
    "nbc_state_key": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64 * 4),
                                C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "nbc_lowres_size": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "nbc_packed_weights_bytes": (C.c_size_t, [C.c_int]),
    "nbc_split_f16x2": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "nbc_pack_weights": (C.c_int, [C.POINTER(NbcTensor), C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "nbc_packed_weights_flags": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int]),
    "nbc_weights_flags": (C.c_int, [C.c_void_p]),
    "nbc_activation_exponent": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int32)]),
    "nbc_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "nbc_destroy": (C.c_int, [C.c_void_p]),
    "nbc_attach_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "nbc_load_weights": (C.c_int, [C.c_void_p, C.POINTER(NbcTensor), C.c_int, C.c_int]),
    "nbc_set_normalization": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "nbc_reserve": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "nbc_nonfinite_seen": (C.c_int, [C.c_void_p, C.c_int]),
    "nbc_nonfinite_peek_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "nbc_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                              C.c_void_p]),
    "nbc_upsample_argmax": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "nbc_remove_small_zones": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p]),
    "nbc_resize_cubic_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "nbc_preprocess_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_void_p]),
    "nbc_autotune": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_void_p]),
    "nbc_get_plan_tiles": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int]),
    "nbc_set_plan_tiles": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int]),
    "nbc_default_conv_tile": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "nbc_bcast_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "nbc_set_conv_tile": (C.c_int, [C.c_void_p, C.c_int]),
    "nbc_set_keep_activations": (C.c_int, [C.c_void_p, C.c_int]),
    "nbc_activation_peaks": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]),
    "nbc_read_activation": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t,
                                      C.POINTER(C.c_int64 * 4)]),
    "nbc_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "nbc_num_op_records": (C.c_int, [C.c_void_p]),
    "nbc_get_op_record": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(NbcOpRecord)]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libnbc_hip.so; raise loudly when it is missing (no fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP library is the only implementation of this path. "
            "Build it with `python -m neuralbarkcalculator_amd.build` (needs hipcc).")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().nbc_last_error().decode("utf-8", "replace")


def check(rc: int, what: str = "") -> None:
    """Map the C ABI's integer error convention to Python exceptions (models.py callers expect
    ``RuntimeError`` from a bad ``load_state_dict``)."""
    if rc == NBC_OK:
        return
    msg = last_error()
    raise RuntimeError(f"{what + ': ' if what else ''}{msg} (nbc error {rc})")
