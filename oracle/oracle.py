"""CPU oracle for the spectrum-analyser signal path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product package (``fpga_real_time_fft_analyzer_amd``) never does.

Two halves:
  * integer path  -> ``libspecan_oracle.so`` (C restatement of the RTL arithmetic,
    ``oracle/specan_oracle.c``; each C function cites the reference file:line it follows);
  * float path    -> stock ``scipy.signal.sosfilt`` + ``numpy.fft`` exactly as BASELINE.json's
    north_star names them (these are third-party libraries, not reference files; versions
    pinned in DESIGN.md: scipy 1.15.3 / numpy 2.2.6).

Parity status: the reference has no DSP test vectors; the integer model is
pinned by the hand KATs of SURVEY.md section 8(a) and the committed ROM ``new/hann.vhd``; the
fixed-point FFT stands where an encrypted Xilinx IP stands => "parity unpinned" there.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

N = 16384
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libspecan_oracle.so")
_lib = None

FILTER_DEFAULT_CMD = 0x00   # gui.py:35
FILTER_CUSTOM_CMD = 0xA1    # gui.py:36
FILTER_NONE_CMD = 0xB1      # gui.py:37
FILTER_WIDE_CMD = 0xA2      # build extension: 6-section Q2.14 SOS (not in the reference)


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (seconds)."""
    src = os.path.join(_HERE, "specan_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libspecan_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.or_win_q15_1.restype = C.c_int16
        _lib.or_win_q15_1.argtypes = [C.c_int16, C.c_int16]
        _lib.or_chain_q15.restype = C.c_int
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


# ----------------------------------------------------------------------------- window
def hann_f64(n: int = N) -> np.ndarray:
    w = np.empty(n, np.float64)
    lib().or_hann_f64(_p(w, C.c_double), C.c_int(n))
    return w


def hann_rom_q15(n: int = N) -> np.ndarray:
    r = np.empty(n, np.int16)
    lib().or_hann_rom_q15(_p(r, C.c_int16), C.c_int(n))
    return r


def win_q15_1(x: int, c: int) -> int:
    return int(lib().or_win_q15_1(C.c_int16(x), C.c_int16(c)))


def window_q15(x: np.ndarray, rom: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.int16)
    rom = np.ascontiguousarray(rom, np.int16)
    y = np.empty_like(x)
    lib().or_window_q15(_p(x, C.c_int16), _p(rom, C.c_int16), _p(y, C.c_int16), C.c_int(x.size))
    return y


def window_u16(x: np.ndarray, rom: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.int16)
    rom = np.ascontiguousarray(rom, np.int16)
    y = np.empty_like(x)
    lib().or_window_u16(_p(x, C.c_int16), _p(rom, C.c_int16), _p(y, C.c_int16), C.c_int(x.size))
    return y


# ----------------------------------------------------------------------------- integer IIR
def default_coeffs_q7() -> np.ndarray:
    c = np.empty(12, np.int8)
    lib().or_default_coeffs_q7(_p(c, C.c_int8))
    return c


def biquad_q7(x: np.ndarray, c6) -> np.ndarray:
    x = np.ascontiguousarray(x, np.int16)
    c6 = np.ascontiguousarray(c6, np.int8)
    assert c6.size == 6
    y = np.empty_like(x)
    lib().or_biquad_q7(_p(x, C.c_int16), _p(y, C.c_int16), C.c_int(x.size), _p(c6, C.c_int8))
    return y


def iir12_q7(x: np.ndarray, c12) -> np.ndarray:
    x = np.ascontiguousarray(x, np.int16)
    c12 = np.ascontiguousarray(c12, np.int8)
    assert c12.size == 12
    y = np.empty_like(x)
    lib().or_iir12_q7(_p(x, C.c_int16), _p(y, C.c_int16), C.c_int(x.size), _p(c12, C.c_int8))
    return y


def quantize_sos_q14(sos: np.ndarray) -> np.ndarray:
    """Wide-mode quantiser (build spec): a0-normalise, round(c*2^14), clip to int16."""
    sos = np.asarray(sos, np.float64)
    sos = sos / sos[:, 3:4]
    return np.clip(np.rint(sos * 16384.0), -32768, 32767).astype(np.int16)


def iir_sos_q14(x: np.ndarray, sos_q14: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.int16)
    s = np.ascontiguousarray(sos_q14, np.int16).reshape(-1, 6)
    y = np.empty_like(x)
    lib().or_iir_sos_q14(_p(x, C.c_int16), _p(y, C.c_int16), C.c_int(x.size), _p(s, C.c_int16),
                         C.c_int(s.shape[0]))
    return y


# ----------------------------------------------------------------------------- fixed-point FFT
def fxfft_twiddles():
    wr = np.empty(N, np.int16)
    wi = np.empty(N, np.int16)
    lib().or_fxfft_twiddles(_p(wr, C.c_int16), _p(wi, C.c_int16))
    return wr, wi


def fxfft16k(x: np.ndarray) -> np.ndarray:
    """[N] int16 real -> [N,2] int16 (re, im), SA-FXFFT-1."""
    x = np.ascontiguousarray(x, np.int16)
    assert x.size == N
    out = np.empty((N, 2), np.int16)
    lib().or_fxfft16k(_p(x, C.c_int16), _p(out, C.c_int16))
    return out


def chain_q15(x: np.ndarray, rom: np.ndarray | None = None, win_mode: int = 0,
              filter_cmd: int = FILTER_NONE_CMD, c12=None, sos_q14=None, want_time: bool = False):
    """[B,N] int16 -> [B,N,2] int16 IQ (and optionally the FFT input [B,N] int16)."""
    x = np.ascontiguousarray(x, np.int16).reshape(-1, N)
    B = x.shape[0]
    rom = hann_rom_q15() if rom is None else np.ascontiguousarray(rom, np.int16)
    c12a = np.zeros(12, np.int8) if c12 is None else np.ascontiguousarray(c12, np.int8)
    sosa = np.zeros((1, 6), np.int16) if sos_q14 is None else np.ascontiguousarray(sos_q14, np.int16).reshape(-1, 6)
    out = np.empty((B, N, 2), np.int16)
    tout = np.empty((B, N), np.int16) if want_time else None
    rc = lib().or_chain_q15(_p(x, C.c_int16), _p(out, C.c_int16),
                            _p(tout, C.c_int16) if want_time else None, C.c_int(B),
                            _p(rom, C.c_int16), C.c_int(win_mode), C.c_int(filter_cmd),
                            _p(c12a, C.c_int8), _p(sosa, C.c_int16), C.c_int(sosa.shape[0]))
    if rc != 0:
        raise ValueError(f"bad filter command 0x{filter_cmd:02X}")
    return (out, tout) if want_time else out


# ----------------------------------------------------------------------------- float path
def sosfilt_f64_c(sos: np.ndarray, x: np.ndarray) -> np.ndarray:
    """C restatement of scipy's DF2T loop (checked against scipy in tests)."""
    sos = np.ascontiguousarray(sos, np.float64).reshape(-1, 6)
    x = np.ascontiguousarray(x, np.float64)
    y = np.empty_like(x)
    lib().or_sosfilt_f64(_p(sos, C.c_double), C.c_int(sos.shape[0]), _p(x, C.c_double),
                         _p(y, C.c_double), C.c_int(x.size))
    return y


def sosfilt_f32_c(sos: np.ndarray, x: np.ndarray) -> np.ndarray:
    sos = np.ascontiguousarray(sos, np.float32).reshape(-1, 6)
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    lib().or_sosfilt_f32(_p(sos, C.c_float), C.c_int(sos.shape[0]), _p(x, C.c_float),
                         _p(y, C.c_float), C.c_int(x.size))
    return y


def chain_f64_c(x: np.ndarray, sos: np.ndarray | None) -> np.ndarray:
    """Self-contained C float chain (double inside): [B,N] f32 -> [B,N] f32 magnitudes."""
    x = np.ascontiguousarray(x, np.float32).reshape(-1, N)
    out = np.empty_like(x)
    if sos is None:
        lib().or_chain_f64(_p(x, C.c_float), _p(out, C.c_float), C.c_int(x.shape[0]), None, C.c_int(0))
    else:
        s = np.ascontiguousarray(sos, np.float64).reshape(-1, 6)
        lib().or_chain_f64(_p(x, C.c_float), _p(out, C.c_float), C.c_int(x.shape[0]),
                           _p(s, C.c_double), C.c_int(s.shape[0]))
    return out


def chain_fp(x: np.ndarray, sos: np.ndarray | None, hann: np.ndarray | None = None):
    """The north-star float oracle:
        y = sosfilt(sos, x * hann); X = rfft(y); mag = |X| mirrored to all N bins.
    x: [B,N] float32 (computed in float64).  Returns (y [B,N] f64, X [B,N/2+1] c128, mag [B,N] f64).
    """
    from scipy.signal import sosfilt
    x = np.asarray(x, np.float32).reshape(-1, N).astype(np.float64)
    h = hann_f64() if hann is None else np.asarray(hann, np.float64)
    y = x * h
    if sos is not None:
        y = sosfilt(np.asarray(sos, np.float64), y, axis=-1)
    X = np.fft.rfft(y, axis=-1)
    m = np.abs(X)
    mag = np.concatenate([m, m[:, -2:0:-1]], axis=-1)
    return y, X, mag


def cpu_baseline_chain(x32: np.ndarray, sos: np.ndarray | None, hann: np.ndarray) -> np.ndarray:
    """Exactly the expression BASELINE.md section 2 times."""
    from scipy.signal import sosfilt
    if sos is None:
        return np.abs(np.fft.rfft(x32 * hann, axis=-1))
    return np.abs(np.fft.rfft(sosfilt(sos, x32 * hann, axis=-1), axis=-1))


def decode_mag(frame_bytes: bytes) -> np.ndarray:
    """Frame decoder restated from gui.py:250-260 (int16 LE re, im -> float32 magnitude)."""
    arr = np.frombuffer(frame_bytes, dtype="<i2").reshape(-1, 2)
    re = arr[:, 0].astype(np.float32)
    im = arr[:, 1].astype(np.float32)
    return np.sqrt(re ** 2 + im ** 2)
