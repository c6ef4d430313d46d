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


# ---------------------------------------------------------------------------------------------
# Response preview.  The reference previews only the *unquantised* design
# (generate_filter_response_plot, gui.py:190-230); what the hardware -- and the Q15 path of this
# build -- actually runs is the cascade the 12 uploaded bytes mean to the RTL, which differs (tap order,
# the x64 / >>7 gain of one half, a2 dropped, two sections used three times each).

def filter_response(sos, fs: float = 100.0, worN: int = 2048):
    """(w, h) of an SOS cascade, the numbers behind gui.py:193 (``signal.sosfreqz(sos, worN=2048, fs=fs)``)."""
    return _sig.sosfreqz(np.asarray(sos, np.float64).reshape(-1, 6), worN=worN, fs=fs)


def fpga_effective_sos(quantized_two_sections) -> np.ndarray:
    """The linear filter the uploaded bytes select in the RTL, as a 6x6 scipy SOS.

    One biquad computes y[n] = (B2 x[n] + B1 x[n-1] + B0 x[n-2] - A0 y[n-2] - A1 y[n-1]) / 128 with
    every product floored separately (filter_iir_cust.vhd:96-117); the GUI's byte order
    [b0,b1,b2,a0,a1,a2] lands on ports B0,B1,B2,A0,A1,A2, and A2 is not connected.  Sections
    1,3,5 take bytes 0..5, sections 2,4,6 bytes 6..11 (filter_iir12_cust.vhd:68-240).  The
    truncations are ignored here: this is the small-signal response, for preview only."""
    secs = two_sections_for_fpga(quantized_two_sections)
    rows = []
    for s in secs:
        B0, B1, B2, A0, A1, _A2 = (float(c) for c in s)
        rows.append([B2 / 128.0, B1 / 128.0, B0 / 128.0, 1.0, A1 / 128.0, A0 / 128.0])
    return np.asarray([rows[0], rows[1]] * 3, np.float64)


def quantised_response(quantized_two_sections, fs: float = 100.0, worN: int = 2048):
    """(w, h) of the cascade the FPGA-exact path runs for an upload; compare with
    ``filter_response(design_iir_filter(...))`` to see what quantisation and the port mapping cost."""
    return filter_response(fpga_effective_sos(quantized_two_sections), fs=fs, worN=worN)


def generate_filter_response_plot(sos, fs: float = 100.0, quantized_two_sections=None):
    """Magnitude/phase plot as a ``data:image/png;base64,...`` URL, like gui.py:190-230; when
    ``quantized_two_sections`` is given the hardware's effective response is overlaid.
    Returns None when matplotlib is not importable (the reference prints and returns None on any error)."""
    try:
        import base64
        from io import BytesIO
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except ImportError:
        return None
    w, h = filter_response(sos, fs)
    fig, (ax_m, ax_p) = plt.subplots(2, 1, figsize=(10, 8))
    ax_m.plot(w, 20 * np.log10(np.maximum(np.abs(h), 1e-10)), label="design (float)")
    ax_p.plot(w, np.angle(h, deg=True))
    if quantized_two_sections is not None:
        wq, hq = quantised_response(quantized_two_sections, fs)
        ax_m.plot(wq, 20 * np.log10(np.maximum(np.abs(hq), 1e-10)), label="as uploaded (12 x int8, RTL mapping)")
        ax_p.plot(wq, np.angle(hq, deg=True))
        ax_m.legend()
    ax_m.set_title("Filter Frequency Response")
    ax_m.set_ylabel("Magnitude (dB)")
    ax_p.set_xlabel("Frequency (KHz)")
    ax_p.set_ylabel("Phase (degrees)")
    for ax in (ax_m, ax_p):
        ax.grid(True, alpha=0.3)
        ax.set_xlim(0, fs / 2)
    fig.tight_layout()
    buf = BytesIO()
    fig.savefig(buf, format="png", dpi=100, bbox_inches="tight")
    plt.close(fig)
    return "data:image/png;base64," + base64.b64encode(buf.getvalue()).decode()
