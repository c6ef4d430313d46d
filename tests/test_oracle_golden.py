"""CPU: the oracle against the golden fixtures and the hand KATs.

The reference holds no DSP vectors of its own; fixtures were generated in the build container by
oracle/gen_golden.py from the reference's Python (hann_coeff.py, gui.py helpers) + scipy/numpy.
"""
import hashlib

import numpy as np
import pytest

from conftest import N, load_golden, rel_maxnorm


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ---- G1 / WIN-1, WIN-2
def test_hann_rom_matches_reference_rom(oracle):
    g = load_golden("g1_hann_rom.npz")
    rom = oracle.hann_rom_q15()
    assert np.array_equal(rom, g["rom"])
    assert sha(rom) == str(g["sha256"])
    # quirk Q1: 58 entries equal -32768, 28 of them wrapped in the centre
    idx = np.where(rom == -32768)[0]
    assert idx.size == 58
    assert np.array_equal(idx[15:43], np.arange(8178, 8206))


def test_hann_float_is_np_hanning(oracle):
    assert np.allclose(oracle.hann_f64(), np.hanning(N), rtol=0, atol=1e-15)


# ---- WIN-3 KATs
@pytest.mark.parametrize("x,c,e", [(2047, 32767, 2047), (2047, -32768, -2047), (-2048, -32768, 2048),
                                   (1000, 12345, 377), (-1000, 12345, -377), (1, 16384, 1), (-1, 16384, 0),
                                   (3, 16384, 2), (-3, 16384, -1), (32767, 32767, 32766),
                                   (-32768, 32767, -32767), (-32768, -32768, 0)])
def test_window_kat(oracle, x, c, e):
    assert oracle.win_q15_1(x, c) == e


# ---- IIR-1 / IIR-2 / IIR-3 KATs
def test_biquad_kats(oracle):
    c = oracle.default_coeffs_q7()
    assert c.tolist() == [-14, 0, 14, 107, 21, 127, -15, 0, 15, 107, -21, 127]
    x = np.zeros(10, np.int16)
    x[0] = 1000
    assert oracle.biquad_q7(x, c[:6]).tolist() == [109, -17, -198, 48, 159, -66, -121, 76, 90, -77]
    assert oracle.biquad_q7(x, c[6:]).tolist() == [117, 20, -211, -50, 169, 70, -129, -79, 96, 83]


def test_cascade_kat(oracle):
    x = np.zeros(16, np.int16)
    x[0] = 32767
    y = oracle.iir12_q7(x, oracle.default_coeffs_q7())
    assert y.tolist() == [0, 0, -1, 0, 3, 0, -14, -1, 38, 1, -85, 0, 163, 0, -276, -2]


def test_q7_term_wraps_like_rtl_slice(oracle):
    # (22 downto 7) of (-32768 * -128) = 2^22 -> 0x8000 -> -32768
    y = oracle.biquad_q7(np.array([-32768], np.int16), np.array([0, 0, -128, 0, 0, 0], np.int8))
    assert y.tolist() == [-32768]


# ---- G4: integer chain digests
def test_q15_chain_golden(oracle):
    g = load_golden("g4_q15_frames.npz")
    rom = load_golden("g1_hann_rom.npz")["rom"]
    x = g["x"]
    for name, cmd, c12 in (("bypass", 0xB1, None), ("default", 0x00, None), ("gui", 0xA1, g["c_gui"])):
        iq, t = oracle.chain_q15(x, rom, 0, cmd, c12, None, want_time=True)
        assert [sha(t[i]) for i in range(4)] == list(g[f"time_{name}_sha"])
        assert [sha(iq[i]) for i in range(4)] == list(g[f"iq_{name}_sha"])
        assert np.array_equal(t[2], g[f"time_{name}_f2"])
    iq, t = oracle.chain_q15(x, rom, 1, 0xA2, None, g["sos_q14"], want_time=True)
    assert [sha(iq[i]) for i in range(4)] == list(g["iq_wide_sha"])
    assert np.array_equal(t[2], g["time_wide_f2"])


def test_bad_filter_cmd(oracle):
    with pytest.raises(ValueError):
        oracle.chain_q15(np.zeros((1, N), np.int16), filter_cmd=0x42)


# ---- SA-FXFFT-1 against fft(x)/N (tolerance: parity unpinned vs the Xilinx core)
@pytest.mark.parametrize("kind", ["adc12", "tone", "fullscale", "dc"])
def test_fxfft_close_to_float_fft(oracle, kind):
    rng = np.random.default_rng(5)
    if kind == "adc12":
        x = rng.integers(-2048, 2048, N)
    elif kind == "tone":
        x = np.round(2047 * np.sin(2 * np.pi * 2488 * np.arange(N) / N))
    elif kind == "fullscale":
        x = rng.integers(-32768, 32768, N)
    else:
        x = np.full(N, 32767)
    x = x.astype(np.int16)
    X = oracle.fxfft16k(x).astype(np.float64)
    ref = np.fft.fft(x.astype(np.float64)) / N
    err = np.abs(X[:, 0] + 1j * X[:, 1] - ref).max()
    assert err <= 6.0, err          # LSB; truncation bias of 7 stages, measured 4.6 worst case


def test_fxfft_linearity_in_shift(oracle):
    # a one-sample impulse has a flat spectrum of 1/N of its height: all bins are 0 or -1 after truncation
    x = np.zeros(N, np.int16)
    x[0] = 16384
    X = oracle.fxfft16k(x)
    assert np.array_equal(X[:, 0], np.ones(N, np.int16)) and not X[:, 1].any()


# ---- float path: C restatement vs scipy, golden G2/G3
def test_sosfilt_c_matches_scipy(oracle):
    from scipy.signal import sosfilt
    g = load_golden("g2_config1.npz")
    x = g["x_f32"].astype(np.float64) * oracle.hann_f64()
    y = oracle.sosfilt_f64_c(g["sos"], x)
    ref = sosfilt(g["sos"], x)
    assert np.abs(y - ref).max() <= 1e-12 * np.abs(ref).max()
    assert np.abs(y - g["y"]).max() <= 1e-12 * np.abs(ref).max()


def test_config1_plumbing(oracle):
    """BASELINE config 1: the oracle chain on the 16K sine frame (fp64 <= 1e-12, fp32 <= 1e-5)."""
    g = load_golden("g2_config1.npz")
    y, X, mag = oracle.chain_fp(g["x_f32"][None, :], g["sos"])
    assert rel_maxnorm(y, g["y"][None, :]) <= 1e-12
    assert np.abs(X[0] - g["X"]).max() <= 1e-12 * np.abs(g["X"]).max()
    assert mag.shape == (1, N) and np.array_equal(mag[0, 1:N // 2], mag[0, :N // 2:-1])
    # C double chain (radix-2 FFT) agrees with numpy's rfft
    m_c = oracle.chain_f64_c(g["x_f32"][None, :], g["sos"])
    assert rel_maxnorm(m_c, mag) <= 1e-6        # result is rounded to f32
    # a straight float32 evaluation of the same recurrence: the config-1 tone (bin 2488) sits 48 dB
    # down in the stop band, so max|y| is ~2e-3 while rounding happens at the scale of the input
    # (~1): one float32 ulp of the input is already 2.5e-5 of max|y|.  The spectrum -- the output of
    # the path -- gains sqrt(N) over that noise and meets 1e-5.
    y32 = oracle.sosfilt_f32_c(g["sos"], (g["x_f32"].astype(np.float64) * oracle.hann_f64()).astype(np.float32))
    assert rel_maxnorm(y32[None, :], g["y"][None, :]) <= 5e-5
    X32 = np.fft.rfft(y32.astype(np.float64))
    assert np.abs(X32 - g["X"]).max() <= 1e-5 * np.abs(g["X"]).max()


def test_g3_frames(oracle):
    g = load_golden("g3_fp32_frames.npz")
    _, _, mag = oracle.chain_fp(g["x"], g["sos"])
    assert rel_maxnorm(mag[:, :N // 2 + 1], g["mag_full"]) <= 1e-6
    _, _, magb = oracle.chain_fp(g["x"], None)
    assert rel_maxnorm(magb[:, :N // 2 + 1], g["mag_bypass"]) <= 1e-6


# ---- G6 frame layout
def test_frame_decode(oracle):
    g = load_golden("g6_frame.npz")
    frame = g["frame"].tobytes()
    assert np.array_equal(oracle.decode_mag(frame), g["mag"])
    assert g["consts"].tolist() == [65536, 16384, 64, 1024, 1025]
    assert g["cmds"].tolist() == [0xA5, 0xFF, 0xEF, 0xFE, 0x55, 0xF1, 0x00, 0xA1, 0xB1]
