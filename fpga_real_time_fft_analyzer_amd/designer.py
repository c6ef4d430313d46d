"""Filter designer + quantiser with the reference's names and argument meaning.

Mirrors scripts/fft_analyzer_gui.py:108-179 (``design_iir_filter``, ``quantize_coefficients``) and
the two-section pad/truncate rule of gui.py:1186-1192, so host code written against the reference
GUI keeps working.  Adds the wide formats the MI355X path accepts (6-section float SOS, Q2.14).
Pure host logic (scipy); no GPU involved.
"""
from __future__ import annotations

import numpy as np
from scipy import signal as _sig

_PAD_SECTION = (64, 0, 0, 64, 0, 0)      # gui.py:1190: identity-ish section used as padding
Q7_SCALE = 64.0                          # gui.py:168


def design_iir_filter(filter_type: str, order: int, cutoff: float, cutoff2: float | None = None,
                      fs: float = 100.0, *, kind: str = "butter", ripple: float = 1.0,
                      attenuation: float = 40):
    """SOS design, same contract as gui.py:108-157: cutoffs and fs in the same unit, band types
    default ``cutoff2`` to ``2*cutoff``, ``kind`` in butter | cheby1 | cheby2 | ellip | bessel."""
    band = filter_type in ("bandpass", "bandstop")
    if band and cutoff2 is None:
        cutoff2 = 2 * cutoff
    nyq = fs / 2.0
    wn = [cutoff / nyq, cutoff2 / nyq] if band else cutoff / nyq
    k = kind.lower()
    if k == "butter":
        return _sig.butter(order, wn, btype=filter_type, output="sos")
    if k == "cheby1":
        return _sig.cheby1(order, ripple, wn, btype=filter_type, output="sos")
    if k == "cheby2":
        return _sig.cheby2(order, attenuation, wn, btype=filter_type, output="sos")
    if k == "ellip":
        return _sig.ellip(order, ripple, attenuation, wn, btype=filter_type, output="sos")
    if k == "bessel":
        return _sig.bessel(order, wn, btype=filter_type, output="sos", norm="phase")
    raise ValueError(f"Unsupported kind: {kind}")


def quantize_coefficients(sos):
    """int8 x64 quantiser of gui.py:159-179: every entry of [b0,b1,b2,a0,a1,a2] becomes
    ``int8(clip(round(c*64), -128, 127))``; returns a list of 6-element lists of numpy int8."""
    out = []
    for row in np.asarray(sos, dtype=np.float64).reshape(-1, 6):
        q = np.clip(np.round(row * Q7_SCALE), -128, 127).astype(np.int8)
        out.append([q[i] for i in range(6)])
    return out


def two_sections_for_fpga(quantized):
    """gui.py:1186-1192: keep the first two sections, pad with [64,0,0,64,0,0]."""
    secs = [list(int(c) for c in s) for s in quantized][:2]
    while len(secs) < 2:
        secs.append(list(_PAD_SECTION))
    return secs


def int8_to_byte(val: int) -> int:
    """gui.py:101-104."""
    return int(val) & 0xFF


def coefficient_upload_bytes(quantized_two_sections) -> bytes:
    """0xF1 + 12 bytes, the exact stream UartReceiver.send_filter_coefficients writes (gui.py:598-605)."""
    secs = two_sections_for_fpga(quantized_two_sections)
    return bytes([0xF1]) + bytes(int8_to_byte(c) for s in secs for c in s)


def quantize_sos_q14(sos) -> np.ndarray:
    """Wide integer format of this build (not in the reference): a0-normalise, round(c*2^14),
    saturate to int16.  |c| < 2 is representable; Butterworth a1 reaches about -1.5."""
    s = np.asarray(sos, np.float64).reshape(-1, 6)
    s = s / s[:, 3:4]
    return np.clip(np.rint(s * 16384.0), -32768, 32767).astype(np.int16)
