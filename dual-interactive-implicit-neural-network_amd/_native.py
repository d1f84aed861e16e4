"""ctypes binding of libdiinn_hip.so (the C ABI in include/diinn_hip.h).

There is deliberately NO fallback: if the shared library is missing or a symbol
cannot be resolved, importing/using this module raises.  The HIP path is the
product; nothing here ever routes to a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# DIINN_HIP_LIB lets kernel experiments load an alternative build of the same ABI
LIB_PATH = os.environ.get("DIINN_HIP_LIB") or os.path.join(PKG_DIR, "libdiinn_hip.so")

DIINN_OK = 0
ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_TOO_LARGE = 1, 2, 3, 4
SIN_ACCURATE = 0
SIN_HW = 1
SIN_HW_REDUCED = 2
# default: 2-term reduction in revolutions + v_sin_f32 (max abs error 2.5e-7 for |x| <= 1e4, measured in
# tests/test_gpu_parity.py::test_device_sine_accuracy); SIN_ACCURATE (1e-7) costs ~4 % more time
SIN_DEFAULT = SIN_HW_REDUCED
ABI_VERSION = 9
PACKED_MAGIC = 0x44493038
PACKED_MAGIC_WPU = 0x44495750          # a training image: permutation sections + section 13 (WPU) filled on the device
P_ALGO_DIRECT, P_ALGO_WINOGRAD, P_ALGO_DIRECT_BF16, P_ALGO_DIRECT_BF16X3 = 0, 1, 2, 3
RDN_ALGO_AUTO, RDN_ALGO_DIRECT, RDN_ALGO_WINO, RDN_ALGO_WINO4, RDN_ALGO_X3 = 0, 1, 2, 3, 4
COMPUTE_F32 = 0
COMPUTE_BF16 = 1
COMPUTE_F32_QONLY = 2
COMPUTE_BF16_FULL = 3
COMPUTE_BF16X3 = 4
COMPUTE = {"f32": COMPUTE_F32, "fp32": COMPUTE_F32, "bf16": COMPUTE_BF16, "bf16_full": COMPUTE_BF16_FULL,
           "bf16x3": COMPUTE_BF16X3}

_f = C.POINTER(C.c_float)
_i32 = C.POINTER(C.c_int32)
_ip = C.POINTER(C.c_int)
_f3 = _f * 3

# name -> (restype, argtypes): every symbol include/diinn_hip.h declares
SIGNATURES = {
    "diinn_abi_version": (C.c_int, []),
    "diinn_status_string": (C.c_char_p, [C.c_int]),
    "diinn_last_hip_error": (C.c_int, []),
    "diinn_packed_weight_floats": (C.c_size_t, []),
    "diinn_packed_section": (C.c_int, [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "diinn_pack_weights": (C.c_int, [_f, _f, _f3, _f3, _f, _f, _f3, _f3, _f, _f, _f]),
    "diinn_make_axis_tables": (C.c_int, [C.c_int, C.c_int, C.c_int, _i32, _f]),
    "diinn_uses_small_output_kernel": (C.c_int, [C.c_int, C.c_int]),
    "diinn_make_axis_tables_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "diinn_eval_sin_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "diinn_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "diinn_lr_rows_for_band": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip, _ip]),
    "diinn_precompute_P": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "diinn_precompute_P_wpu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "diinn_precompute_P_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "diinn_decode_band": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "diinn_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "diinn_decode_band_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "diinn_decode_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "diinn_window_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip, _ip, _ip, _ip]),
    "diinn_precompute_P_win": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "diinn_decode_band_win": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "diinn_decode_win": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                   C.c_void_p, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "diinn_decode_tile_win": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_longlong, C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "diinn_cell_chain": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "diinn_decode_launch_info": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip, _ip, _ip, _ip]),
    "diinn_p_launch_info": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip]),
    "diinn_decode_kernel_info": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip]),
    "diinn_debug_set": (C.c_int, [C.c_char_p, C.c_longlong]),
    "diinn_debug_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_longlong)]),
    "diinn_debug_clock_probe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_uint]),
    "diinn_metasr_packed_floats": (C.c_size_t, []),
    "diinn_metasr_pack_weights": (C.c_int, [_f, _f, _f, _f, _f]),
    "diinn_metasr_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "diinn_metasr_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "diinn_metasr_make_axis_tables": (C.c_int, [C.c_int, C.c_int, _i32, _f, _f]),
    "diinn_conv_ksplit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong,
                                   C.c_int, C.c_int, C.c_int, C.c_int]),
    "diinn_rdn_packed_floats": (C.c_size_t, []),
    "diinn_rdn_planes_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "diinn_rdn_forward_ex": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "diinn_conv_wino4_ws_status": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _ip]),
    "diinn_rdn_workspace_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "diinn_rdn_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int]),
    "diinn_sfe1_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int]),
    "diinn_conv_wino": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int]),
    "diinn_conv3x3_x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int]),
    "diinn_rdn_x3_packed_floats": (C.c_size_t, []),
    "diinn_conv_t16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int]),
    "diinn_conv_t16_applies": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "diinn_conv_t16_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "diinn_conv1x1_t16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong,
                                   C.c_int, C.c_int, C.c_int, C.c_int]),
    "diinn_rdn_x3_workspace_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "diinn_rdn_forward_x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "diinn_rdn_wino_packed_floats": (C.c_size_t, []),
    "diinn_conv_wino4": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int]),
    "diinn_conv_wino4_ws": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_size_t]),
    "diinn_conv_wino4_workspace_floats": (C.c_size_t, []),
    "diinn_conv_wino4_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "diinn_rdn_wino4_packed_floats": (C.c_size_t, []),
    "diinn_rdn_wino4_applies": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "diinn_rdn_forward_wino4": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "diinn_rdn_forward_wino": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "diinn_liif_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "diinn_liif_make_axis_tables": (C.c_int, [C.c_int, C.c_int, C.c_int, _i32, _f, _f]),
    "diinn_backward_data": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_longlong]),
    "diinn_plane_gemm_nt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                      C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int]),
    "diinn_plane_rowdot": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_longlong, C.c_int]),
    "diinn_backward_cell_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "diinn_sum_parts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_longlong]),
    "diinn_backward_cell_sum_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "diinn_unfold_tiled": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "diinn_training_plane_floats": (C.c_longlong, [C.c_longlong, C.c_int]),
    "diinn_decode_train_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
}

_lib = None
_lock = threading.Lock()


class DiinnNativeError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the library (once) and bind every declared symbol; raise if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise DiinnNativeError(
                f"{LIB_PATH} not found: build it with `python __graft_entry__.py` or "
                f"`python -c 'import diinn_amd.build as b; b.build()'` (needs hipcc). "
                f"There is no CPU fallback for the DIINN decode path.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise DiinnNativeError(f"{LIB_PATH} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        ver = lib.diinn_abi_version()
        if ver != ABI_VERSION:
            raise DiinnNativeError(f"ABI version mismatch: library {ver}, binding {ABI_VERSION}")
        _lib = lib
    return _lib


def check(status: int, what: str) -> None:
    if status != DIINN_OK:
        lib = load()
        msg = lib.diinn_status_string(status).decode()
        extra = f" (hipError_t {lib.diinn_last_hip_error()})" if status == 3 else ""
        raise DiinnNativeError(f"{what} failed: {msg}{extra}")


def debug_set(name: str, value: int) -> None:
    """Force a kernel-variant choice in-process (tests, A/B timing): include/diinn_hip.h "diagnostic overrides"."""
    check(load().diinn_debug_set(name.encode(), int(value)), f"diinn_debug_set({name})")


def debug_get(name: str) -> int:
    v = C.c_longlong()
    check(load().diinn_debug_get(name.encode(), C.byref(v)), f"diinn_debug_get({name})")
    return v.value


def fptr(arr):
    """float* of a contiguous float32 numpy array."""
    return arr.ctypes.data_as(_f)
