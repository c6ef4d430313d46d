#!/usr/bin/env python3
"""Generate tests/golden/*.npz.  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).

What comes from where:
  * reference Python, imported here as-is:  scripts/hann_coeff.py (run in a temp dir),
    and the pure helpers of scripts/fft_analyzer_gui.py (design_iir_filter :108,
    quantize_coefficients :159, decode_mag_16iq_le :250, decode_iq_components :262,
    MultiPacketAssembler :308) with PyQt5 / flask_socketio stubbed in sys.modules
    (both absent here -> ordinary ModuleNotFoundError; nothing was denied by the environment).
  * stock scipy.signal.sosfilt / numpy.fft (the float oracle BASELINE.json names).
  * this build's integer model (oracle/specan_oracle.c) for G4 -- the reference has no
    RTL simulator output; those vectors are pinned by the hand KATs of SURVEY.md 8(a) only.

The reference itself never travels: only inputs and expected outputs are stored.
"""
from __future__ import annotations

import hashlib
import os
import re
import runpy
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
N = 16384

sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def import_gui():
    """Import scripts/fft_analyzer_gui.py with the two absent GUI packages stubbed."""
    qt = types.ModuleType("PyQt5")
    qtcore = types.ModuleType("PyQt5.QtCore")

    class _QObject:  # minimal stand-ins: only class definitions touch them at import time
        def __init__(self, *a, **k):
            pass

    def _sig(*a, **k):
        return None

    def _slot(*a, **k):
        return lambda f: f

    qtcore.QObject = _QObject
    qtcore.pyqtSignal = _sig
    qtcore.pyqtSlot = _slot
    qtcore.QTimer = _QObject
    qtcore.QByteArray = bytes
    qtcore.QIODevice = _QObject
    qtcore.QThread = _QObject
    qtcore.QCoreApplication = _QObject
    qtcore.QMetaObject = _QObject
    qtcore.Qt = types.SimpleNamespace(QueuedConnection=0)
    qtcore.Q_ARG = lambda *a, **k: None
    qtnet = types.ModuleType("PyQt5.QtNetwork")
    qtnet.QUdpSocket = _QObject
    qtnet.QHostAddress = _QObject
    qtw = types.ModuleType("PyQt5.QtWidgets")
    qtw.QApplication = _QObject
    qt.QtCore, qt.QtNetwork, qt.QtWidgets = qtcore, qtnet, qtw
    sio = types.ModuleType("flask_socketio")

    class _SocketIO:
        def __init__(self, *a, **k):
            pass

        def on(self, *a, **k):
            return lambda f: f

        def emit(self, *a, **k):
            pass

        def run(self, *a, **k):
            pass

    sio.SocketIO = _SocketIO
    sio.emit = lambda *a, **k: None
    for name, mod in [("PyQt5", qt), ("PyQt5.QtCore", qtcore), ("PyQt5.QtNetwork", qtnet),
                      ("PyQt5.QtWidgets", qtw), ("flask_socketio", sio)]:
        sys.modules.setdefault(name, mod)
    sys.path.insert(0, os.path.join(REF, "scripts"))
    import fft_analyzer_gui as gui  # noqa
    return gui


def run_hann_coeff() -> np.ndarray:
    """Run scripts/hann_coeff.py in a temp dir and parse the package it writes."""
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            runpy.run_path(os.path.join(REF, "scripts", "hann_coeff.py"), run_name="__main__")
            txt = open("hann.vhd").read()
        finally:
            os.chdir(cwd)
    vals = [int(v) for v in re.findall(r"to_signed\((-?\d+),16\)", txt)]
    rom = np.array(vals, dtype=np.int16)
    committed = open(os.path.join(REF, "SDR_v2.srcs/sources_1/new/hann.vhd")).read()
    cvals = np.array([int(v) for v in re.findall(r"to_signed\((-?\d+),16\)", committed)], dtype=np.int16)
    assert rom.shape == (N,) and np.array_equal(rom, cvals), "regenerated ROM differs from new/hann.vhd"
    return rom


def main():
    os.makedirs(OUT, exist_ok=True)
    gui = import_gui()
    from scipy.signal import sosfilt

    # ---- G1: the Q15 Hann ROM (scripts/hann_coeff.py == new/hann.vhd)
    rom = run_hann_coeff()
    assert np.array_equal(rom, orc.hann_rom_q15()), "C restatement of hann_coeff.py disagrees"
    hann = 0.5 * (1 - np.cos(2 * np.pi * np.arange(N) / (N - 1)))       # hann_coeff.py:3-4
    assert np.array_equal(hann, orc.hann_f64()), "C hann_f64 disagrees bitwise"
    np.savez_compressed(os.path.join(OUT, "g1_hann_rom.npz"), rom=rom, sha256=np.array(sha(rom)))

    # ---- G2: config 1 -- one 16K sine frame through the designer's 12th-order Butterworth
    n = np.arange(N)
    x_i16 = np.round(2047 * np.sin(2 * np.pi * 2488 * n / N)).astype(np.int16)
    x_f32 = (x_i16.astype(np.float32) / np.float32(2048.0)).astype(np.float32)
    sos12 = gui.design_iir_filter("lowpass", 12, 10.0, None, 100.0, kind="butter")   # wn = 0.2
    assert sos12.shape == (6, 6)
    y = sosfilt(sos12, x_f32.astype(np.float64) * hann)
    X = np.fft.rfft(y)
    Xb = np.fft.rfft(x_f32.astype(np.float64) * hann)
    np.savez_compressed(os.path.join(OUT, "g2_config1.npz"), x_i16=x_i16, x_f32=x_f32, sos=sos12,
                        y=y, X=X, X_bypass=Xb)

    # ---- G3: 8 random fp32 frames (tone + noise), bypass and full chain -> magnitudes
    rng = np.random.default_rng(1234)
    fb = rng.uniform(0.01, 0.45, size=8)
    x3 = (0.8 * np.sin(2 * np.pi * fb[:, None] * n[None, :]) +
          0.05 * rng.standard_normal((8, N))).astype(np.float32)
    xw = x3.astype(np.float64) * hann
    mag_bypass = np.abs(np.fft.rfft(xw, axis=-1))
    y3 = sosfilt(sos12, xw, axis=-1)
    mag_full = np.abs(np.fft.rfft(y3, axis=-1))
    np.savez_compressed(os.path.join(OUT, "g3_fp32_frames.npz"), x=x3, sos=sos12,
                        y_full=y3.astype(np.float32), mag_bypass=mag_bypass.astype(np.float32),
                        mag_full=mag_full.astype(np.float32))

    # ---- G5: quantiser table at the GUI defaults (gui.py:76-78: lowpass, order 4, 10/100) per kind
    g5 = {}
    for kind in ("butter", "cheby1", "cheby2", "ellip", "bessel"):
        sos = gui.design_iir_filter("lowpass", 4, 10.0, 20.0, 100.0, kind=kind, ripple=1.0, attenuation=40)
        q = np.array(gui.quantize_coefficients(sos), dtype=np.int8)
        g5[f"sos_{kind}"] = sos
        g5[f"q_{kind}"] = q
    sosq12 = np.array(gui.quantize_coefficients(sos12), dtype=np.int8)
    g5["q_butter12"] = sosq12
    # other filter types at order 4 (band types use cutoff2 = 20)
    for ft in ("highpass", "bandpass", "bandstop"):
        sos = gui.design_iir_filter(ft, 4, 10.0, 20.0, 100.0, kind="butter")
        g5[f"sos_butter_{ft}"] = sos
        g5[f"q_butter_{ft}"] = np.array(gui.quantize_coefficients(sos), dtype=np.int8)
    np.savez_compressed(os.path.join(OUT, "g5_quantiser.npz"), **g5)
    gui_default = g5["q_butter"]                    # [[0,1,0,64,-67,19],[64,127,64,64,-85,40]]
    assert gui_default.tolist() == [[0, 1, 0, 64, -67, 19], [64, 127, 64, 64, -85, 40]], gui_default

    # ---- G4: integer path (build's integer model; pinned by SURVEY 8(a) KATs)
    rng = np.random.default_rng(4321)
    x4 = np.concatenate([rng.integers(-2048, 2048, size=(3, N)), rng.integers(-32768, 32768, size=(1, N))]
                        ).astype(np.int16)
    x4[2] = x_i16                                    # include the config-1 tone
    c_def = orc.default_coeffs_q7()
    c_gui = gui_default.reshape(12)
    g4 = {"x": x4, "c_default": c_def, "c_gui": c_gui}
    for name, cmd, c12 in (("bypass", 0xB1, None), ("default", 0x00, None), ("gui", 0xA1, c_gui)):
        iq, t = orc.chain_q15(x4, rom, 0, cmd, c12, None, want_time=True)
        g4[f"time_{name}_sha"] = np.array([sha(t[i]) for i in range(4)])
        g4[f"iq_{name}_sha"] = np.array([sha(iq[i]) for i in range(4)])
        g4[f"time_{name}_f2"] = t[2]
        if name == "default":
            g4["iq_default_f2"] = iq[2]
    sos_q14 = orc.quantize_sos_q14(sos12)
    iq, t = orc.chain_q15(x4, rom, 1, 0xA2, None, sos_q14, want_time=True)
    g4["sos_q14"] = sos_q14
    g4["time_wide_sha"] = np.array([sha(t[i]) for i in range(4)])
    g4["iq_wide_sha"] = np.array([sha(iq[i]) for i in range(4)])
    g4["time_wide_f2"] = t[2]
    np.savez_compressed(os.path.join(OUT, "g4_q15_frames.npz"), **g4)

    # ---- G6: one 65536-byte frame + the reference decoder's view of it
    iq2 = g4["iq_default_f2"]
    frame = iq2.astype("<i2").tobytes()
    assert len(frame) == 65536
    mag = gui.decode_mag_16iq_le(frame)
    re_, im_ = gui.decode_iq_components(frame)
    # MultiPacketAssembler semantics (gui.py:308-352): 64 packets x (index byte + 1024 B)
    asm = gui.MultiPacketAssembler(gui.PACKETS_PER_FRAME, gui.PACKET_DATA_SIZE)
    got = None
    order = list(range(64))
    np.random.default_rng(7).shuffle(order)
    for idx in order:
        got = asm.add(bytes([idx]) + frame[idx * 1024:(idx + 1) * 1024], 0)
    assert got == frame
    np.savez_compressed(os.path.join(OUT, "g6_frame.npz"), frame=np.frombuffer(frame, np.uint8),
                        mag=mag, re=re_, im=im_,
                        consts=np.array([gui.FRAME_SIZE_BYTES, gui.FFT_SIZE, gui.PACKETS_PER_FRAME,
                                         gui.PACKET_DATA_SIZE, gui.ETHERNET_PAYLOAD_SIZE]),
                        cmds=np.array([gui.UART_REQUEST_CMD, gui.FPGA_RESET_CMD, gui.ETHERNET_MODE_CMD,
                                       gui.UART_MODE_CMD, gui.START_COMMAND, gui.FILTER_UPDATE_CMD,
                                       gui.FILTER_DEFAULT_CMD, gui.FILTER_CUSTOM_CMD, gui.FILTER_NONE_CMD]))
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"golden fixtures written to {OUT}: {tot / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
