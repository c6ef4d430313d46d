"""ctypes binding of ``libspecan_hip.so`` (the C ABI declared in ``include/specan.h``).

This is the only place the shared library is loaded.  There is no CPU fallback: if the
library is missing the import of any compute entry point raises, and ``sa_create`` itself
fails when no HIP device is usable.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libspecan_hip.so")
CSRC = os.path.join(_PKG, "csrc")

# error codes / constants (include/specan.h)
SA_OK, SA_EINVAL, SA_ESHAPE, SA_EHIP, SA_ESTATE, SA_ENOMEM = 0, -1, -2, -3, -4, -5
SA_N = 16384
SA_FRAME_BYTES = 65536
SA_FILTER_DEFAULT, SA_FILTER_CUSTOM, SA_FILTER_NONE, SA_FILTER_WIDE = 0x00, 0xA1, 0xB1, 0xA2
SA_WIN_RTL_SIGNED, SA_WIN_HANN_U16 = 0, 1
SA_OUT_MAG_FULL, SA_OUT_MAG_HALF, SA_OUT_SPEC_HALF, SA_OUT_TIME = 0, 1, 2, 3


class CmdEvents(C.Structure):
    """sa_cmd_events of include/specan.h."""
    _fields_ = [("n_start", C.c_int), ("n_uart_request", C.c_int), ("n_reset", C.c_int), ("n_uploads", C.c_int),
                ("control_changed", C.c_int), ("transport", C.c_uint8)]


class SpecanError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"specan error {code}: {msg}")
        self.code = code


def build(verbose: bool = False) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-j4", "-C", CSRC]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    """Load the shared library; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built (run __graft_entry__.build() or "
            f"`make -C {CSRC}`).  This package has no CPU fallback.")
    # torch first: its wheel bundles a HIP runtime under the same SONAME (libamdhip64.so.7) but another file
    # name, so if this library is mapped first the process ends up with two runtimes and the second one to
    # initialise sees no device.  With torch's copy already mapped, the loader binds this library to it and
    # the tensors and the kernels share one runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    H = C.c_void_p
    L.sa_create.argtypes = [C.c_int, C.POINTER(H)]
    L.sa_destroy.argtypes = [H]
    L.sa_abi_version.argtypes = []
    L.sa_last_error.argtypes = [H]
    L.sa_last_error.restype = C.c_char_p
    L.sa_reserve.argtypes = [H, C.c_int]
    L.sa_set_overlap.argtypes = [H, C.c_int]
    L.sa_get_overlap.argtypes = [H, C.POINTER(C.c_int)]
    L.sa_debug_overlap_streams.argtypes = [H, C.c_void_p, C.POINTER(C.c_int)]
    L.sa_flush.argtypes = [H, C.c_void_p]
    L.sa_set_profiling.argtypes = [H, C.c_int]
    L.sa_profile_read.argtypes = [H, C.POINTER(C.c_float), C.c_int]
    L.sa_set_filter_mode.argtypes = [H, C.c_uint8]
    L.sa_get_filter_mode.argtypes = [H, C.POINTER(C.c_uint8)]
    L.sa_load_coeffs_q7.argtypes = [H, C.POINTER(C.c_int8)]
    L.sa_get_coeffs_q7.argtypes = [H, C.POINTER(C.c_int8)]
    L.sa_feed_command_bytes.argtypes = [H, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_int)]
    L.sa_feed_command_bytes_ex.argtypes = [H, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(CmdEvents)]
    L.sa_get_transport.argtypes = [H, C.POINTER(C.c_uint8)]
    L.sa_load_sos_f32.argtypes = [H, C.POINTER(C.c_float), C.c_int]
    L.sa_load_sos_f64.argtypes = [H, C.POINTER(C.c_double), C.c_int]
    L.sa_load_sos_q14.argtypes = [H, C.POINTER(C.c_int16), C.c_int]
    L.sa_set_window_q15.argtypes = [H, C.POINTER(C.c_int16)]
    L.sa_set_window_f32.argtypes = [H, C.POINTER(C.c_float)]
    L.sa_set_window_mode_q15.argtypes = [H, C.c_int]
    L.sa_get_window_q15.argtypes = [H, C.POINTER(C.c_int16)]
    L.sa_process_q15.argtypes = [H, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.sa_filter_q15.argtypes = [H, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.sa_process_f32.argtypes = [H, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.sa_process_f32_i16.argtypes = [H, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.sa_pack_frame.argtypes = [C.POINTER(C.c_int16), C.POINTER(C.c_uint8)]
    L.sa_debug_iir_plan_f32.argtypes = [H, C.POINTER(C.c_float), C.c_int]
    L.sa_iir_plan_from_sos.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_float), C.c_int]
    for name in ("sa_create", "sa_destroy", "sa_abi_version", "sa_reserve", "sa_set_overlap", "sa_get_overlap",
                 "sa_debug_overlap_streams", "sa_flush", "sa_set_profiling", "sa_profile_read", "sa_set_filter_mode",
                 "sa_get_filter_mode", "sa_load_coeffs_q7", "sa_get_coeffs_q7", "sa_feed_command_bytes",
                 "sa_feed_command_bytes_ex", "sa_get_transport",
                 "sa_load_sos_f32", "sa_load_sos_f64", "sa_load_sos_q14", "sa_set_window_q15",
                 "sa_set_window_f32", "sa_set_window_mode_q15", "sa_get_window_q15", "sa_process_q15",
                 "sa_filter_q15", "sa_process_f32", "sa_process_f32_i16", "sa_pack_frame", "sa_debug_iir_plan_f32",
                 "sa_iir_plan_from_sos"):
        getattr(L, name).restype = C.c_int
    _lib = L
    return L
