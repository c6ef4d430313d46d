"""GPU: Q15 integer path, bit-exact against the integer model (oracle/specan_oracle.c) and the
golden digests, through the C ABI."""
import hashlib

import numpy as np
import pytest

from udp_collect import FrameCollector
from conftest import N, load_golden

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture()
def ch(chain_cls):
    c = chain_cls(0)
    yield c
    c.close()


def _dev(torch_mod, a):
    return torch_mod.from_numpy(np.ascontiguousarray(a)).cuda()


def test_power_on_state(ch, oracle):
    assert ch.filter_mode == 0xB1                                      # new/command_control.vhd:31
    assert not ch.coeffs_q7().any()                                    # new/filter_iir12_cust.vhd:51-52
    assert np.array_equal(ch.window_q15(), load_golden("g1_hann_rom.npz")["rom"])


def test_golden_digests_all_modes(ch, torch_mod):
    g = load_golden("g4_q15_frames.npz")
    x = _dev(torch_mod, g["x"])
    for name, cmd, c12 in (("bypass", 0xB1, None), ("default", 0x00, None), ("gui", 0xA1, g["c_gui"])):
        if c12 is not None:
            ch.load_coeffs_q7(c12)
        ch.set_filter_mode(cmd)
        t = ch.filter_q15(x).cpu().numpy()
        iq = ch.process_q15(x).cpu().numpy()
        assert np.array_equal(t[2], g[f"time_{name}_f2"])
        assert [sha(t[i]) for i in range(4)] == list(g[f"time_{name}_sha"])
        assert [sha(iq[i]) for i in range(4)] == list(g[f"iq_{name}_sha"])
    # wide Q2.14 mode with the unsigned-Hann window
    ch.load_sos_q14(g["sos_q14"])
    ch.set_window_mode_q15(1)
    ch.set_filter_mode(0xA2)
    t = ch.filter_q15(x).cpu().numpy()
    iq = ch.process_q15(x).cpu().numpy()
    assert np.array_equal(t[2], g["time_wide_f2"])
    assert [sha(iq[i]) for i in range(4)] == list(g["iq_wide_sha"])


@pytest.mark.parametrize("B,cmd", [(1, 0xB1), (7, 0x00), (9, 0xA1), (33, 0xB1), (64, 0x00), (1, 0xA2), (17, 0xA2), (65, 0xA2)])
def test_bit_exact_vs_integer_model(ch, torch_mod, oracle, B, cmd):
    rng = np.random.default_rng(B * 7 + cmd)
    x = rng.integers(-2048, 2048, size=(B, N)).astype(np.int16)
    x[-1] = rng.integers(-32768, 32768, size=N)                      # one full-scale frame
    c12 = np.array([0, 1, 0, 64, -67, 19, 64, 127, 64, 64, -85, 40], np.int8)
    sos14 = load_golden("g4_q15_frames.npz")["sos_q14"][:1 + B % 6]      # ragged batches (last wave, last workgroup partly
    ch.load_coeffs_q7(c12)                                               # filled) for the wide cascade too, 1..6 sections
    ch.load_sos_q14(sos14)
    ch.set_filter_mode(cmd)
    ref_iq, ref_t = oracle.chain_q15(x, None, 0, cmd, c12, sos14, want_time=True)
    xd = _dev(torch_mod, x)
    assert np.array_equal(ch.filter_q15(xd).cpu().numpy(), ref_t)
    assert np.array_equal(ch.process_q15(xd).cpu().numpy(), ref_iq)


def test_extreme_inputs(ch, torch_mod, oracle):
    x = np.zeros((4, N), np.int16)
    x[0] = 32767
    x[1] = -32768
    x[2, ::2] = 32767
    x[2, 1::2] = -32768
    x[3, 0] = -32768
    for cmd in (0xB1, 0x00):
        ch.set_filter_mode(cmd)
        ref = oracle.chain_q15(x, None, 0, cmd, None, None)
        assert np.array_equal(ch.process_q15(_dev(torch_mod, x)).cpu().numpy(), ref)
    e = torch_mod.empty((0, N), dtype=torch_mod.int16, device="cuda")
    assert ch.process_q15(e).shape == (0, N, 2)


def test_random_coefficient_uploads(ch, torch_mod, oracle):
    """60 random 12-byte uploads (every int8 value allowed, including -128 and unstable feedback taps whose
    outputs wrap around) on full-scale and 12-bit inputs: the window + cascade output and the IQ frame are
    bit-exact against the integer model."""
    rng = np.random.default_rng(2024)
    ch.set_filter_mode(0xA1)
    for case in range(60):
        c12 = rng.integers(-128, 128, 12).astype(np.int8)
        if case % 5 == 0:
            c12[rng.integers(0, 12)] = -128
        x = rng.integers(-32768, 32768, (3, N)).astype(np.int16)
        x[1] = rng.integers(-2048, 2048, N)
        ch.load_coeffs_q7(c12)
        iq_ref, t_ref = oracle.chain_q15(x, None, 0, 0xA1, c12, None, want_time=True)
        assert np.array_equal(ch.filter_q15(_dev(torch_mod, x)).cpu().numpy(), t_ref), (case, c12)
        assert np.array_equal(ch.process_q15(_dev(torch_mod, x)).cpu().numpy(), iq_ref), (case, c12)


def test_zero_tap_uploads_take_the_short_step_and_stay_exact(ch, torch_mod, oracle):
    """The cascade kernel drops the product and the add of port tap B1 when it is zero in BOTH coefficient sets (the
    fixed ALPHA / BETA cascade of mode 0x00, imp/filter_pkg.vhd:54-68: seven instructions per step instead of nine).
    t(0, v) = 0 exactly (new/filter_iir_cust.vhd:96-117 truncates every product separately), so results must not move:
    random uploads with B1 = 0 in both sets (short step), in one set only and in neither (long step), plus other zero
    taps, all bit-exact against the integer model; mode 0x00 itself equals the same bytes uploaded in mode 0xA1."""
    rng = np.random.default_rng(777)
    x = rng.integers(-32768, 32768, (3, N)).astype(np.int16)
    x[1] = rng.integers(-2048, 2048, N)
    xd = _dev(torch_mod, x)
    ch.set_filter_mode(0xA1)
    for case in range(36):
        c12 = rng.integers(-128, 128, 12).astype(np.int8)            # wire order b0,b1,b2,a0,a1,a2 per set
        kind = case % 4
        if kind == 0:
            c12[1] = c12[7] = 0                                      # short step
        elif kind == 1:
            c12[1] = 0                                               # one set only: long step
        elif kind == 2:
            c12[1] = c12[7] = 0
            c12[rng.integers(0, 12)] = 0                             # another zero tap on top
            c12[1] = c12[7] = 0
        ch.load_coeffs_q7(c12)
        iq_ref, t_ref = oracle.chain_q15(x, None, 0, 0xA1, c12, None, want_time=True)
        assert np.array_equal(ch.filter_q15(xd).cpu().numpy(), t_ref), (case, c12)
        assert np.array_equal(ch.process_q15(xd).cpu().numpy(), iq_ref), (case, c12)
    default = np.array([-14, 0, 14, 107, 21, 127, -15, 0, 15, 107, -21, 127], np.int8)      # imp/filter_pkg.vhd:54-68
    ch.load_coeffs_q7(default)
    up = ch.process_q15(xd).cpu().numpy()
    ch.set_filter_mode(0x00)
    assert np.array_equal(ch.process_q15(xd).cpu().numpy(), up)
    iq_ref, _ = oracle.chain_q15(x, None, 0, 0x00, None, None, want_time=True)
    assert np.array_equal(up, iq_ref)


def test_overlapped_launches_q15(ch, torch_mod, oracle):
    """sa_set_overlap on the integer path: the FFT of batch k-1 may run under the cascade of batch k (one workspace per
    launch slot, grown on demand without touching launches in flight).  Bit-exact against the integer model for growing
    batch sizes, a coefficient upload between overlapped calls, and back in the ordered mode."""
    torch = torch_mod
    rng = np.random.default_rng(99)
    sizes = [3, 17, 5, 64, 33, 2]
    xs = [rng.integers(-2048, 2048, (b, N)).astype(np.int16) for b in sizes]
    gui = np.array([0, 1, 0, 64, -67, 19, 64, 127, 64, 64, -85, 40], np.int8)
    wide = load_golden("g4_q15_frames.npz")["sos_q14"]
    ch.set_filter_mode(0x00)
    ch.set_overlap(2)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        xd = [_dev(torch, x) for x in xs]
        outs = []
        for k, x in enumerate(xd):
            if k == 3:                                       # control plane between overlapped calls: custom coefficients
                ch.load_coeffs_q7(gui)
                ch.set_filter_mode(0xA1)
            if k == 5:                                       # ... and the wide cascade (ordered behind the previous call)
                ch.load_sos_q14(wide)
                ch.set_filter_mode(0xA2)
            outs.append(ch.process_q15(x))
        ch.flush()
        got = [o.cpu().numpy() for o in outs]
    s.synchronize()
    for k, x in enumerate(xs):
        cmd = 0x00 if k < 3 else (0xA1 if k < 5 else 0xA2)
        ref = oracle.chain_q15(x, None, 0, cmd, gui if cmd == 0xA1 else None, wide if cmd == 0xA2 else None)
        assert np.array_equal(got[k], ref), k
    ch.set_filter_mode(0xA1)
    ch.set_overlap(1)
    assert np.array_equal(ch.process_q15(_dev(torch, xs[3])).cpu().numpy(), got[3])


def test_mode_changes_do_not_grow_the_workspace(chain_cls, torch_mod):
    """sa_set_overlap gives every launch slot the workspace the largest slot has.  With the geometric growth of the
    process calls applied there as well, two slots leap-frogged each other by a factor 1.5 per mode change until
    hipMalloc failed (a 10-minute soak found it).  300 mode changes with batches of changing size must leave the
    device's free memory where a handful of workspaces put it."""
    torch = torch_mod
    c = chain_cls(0)
    c.set_filter_mode(0x00)
    rng = np.random.default_rng(4)
    x = torch.zeros((257, N), dtype=torch.int16, device="cuda")
    out = torch.empty((257, N, 2), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for i in range(300):
        c.set_overlap(int(rng.choice([1, 2, 3])))
        B = int(rng.choice([1, 16, 100, 129, 200, 257]))
        c.process_q15(x[:B], out=out[:B])
        c.flush()
    torch.cuda.synchronize()
    used = free0 - torch.cuda.mem_get_info()[0]
    c.close()
    assert used < 256 * 1024 * 1024, f"{used / 2**20:.0f} MiB of workspace after 300 mode changes"


def test_handles_come_and_go_without_leaking(chain_cls, torch_mod):
    """120 handles created, put into overlap mode, used on both paths and destroyed: device memory returns to where it
    was (tables, workspaces, streams and events are the handle's own and sa_destroy releases them)."""
    torch = torch_mod
    x = torch.zeros((8, N), dtype=torch.int16, device="cuda")
    xf = torch.zeros((8, N), dtype=torch.float32, device="cuda")
    chain_cls(0).close()                                     # first use: the runtime's one-off allocations
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for i in range(120):
        c = chain_cls(0)
        c.set_overlap(1 + i % 3)
        c.set_filter_mode(0x00)
        c.process_q15(x)
        c.process_f32(xf)
        c.process_f32(x)
        c.flush()
        c.close()
    torch.cuda.synchronize()
    lost = free0 - torch.cuda.mem_get_info()[0]
    assert lost < 64 * 1024 * 1024, f"{lost / 2**20:.0f} MiB not returned after 120 handles"


def test_random_wide_cascades(ch, torch_mod, oracle):
    """Wide mode (0xA2): random Q2.14 cascades of 1..6 sections, any int16 tap (saturating accumulators make
    unstable ones well defined), both window modes, bit-exact against the integer model."""
    rng = np.random.default_rng(77)
    ch.set_filter_mode(0xA2)
    for case in range(30):
        nsec = int(rng.integers(1, 7))
        sos = rng.integers(-32768, 32768, (nsec, 6)).astype(np.int16)
        if case % 2:
            sos = (sos // 4).astype(np.int16)                           # mostly stable-ish magnitudes as well
        x = rng.integers(-32768, 32768, (2, N)).astype(np.int16)
        wm = case % 2
        ch.set_window_mode_q15(wm)
        ch.load_sos_q14(sos)
        ref = oracle.chain_q15(x, None, wm, 0xA2, None, sos)
        assert np.array_equal(ch.process_q15(_dev(torch_mod, x)).cpu().numpy(), ref), (case, sos)
    ch.set_window_mode_q15(0)


def test_custom_rom_and_window_modes(ch, torch_mod, oracle):
    rng = np.random.default_rng(99)
    x = rng.integers(-2048, 2048, size=(3, N)).astype(np.int16)
    rom = rng.integers(-32768, 32768, size=N).astype(np.int16)
    ch.set_window_q15(rom)
    for wm in (0, 1):
        ch.set_window_mode_q15(wm)
        ref = oracle.chain_q15(x, rom, wm, 0xB1, None, None)
        assert np.array_equal(ch.process_q15(_dev(torch_mod, x)).cpu().numpy(), ref)
    # the corners of the window arithmetic: x = c = -32768 (the 17-bit sum +32768 that resize16 maps to 0,
    # new/hann8192.vhd:36-39), full-scale products of either sign, in the FFT-side and the filter-side window
    xf = rng.integers(-32768, 32768, size=(3, N)).astype(np.int16)
    rom[::97] = -32768
    xf[:, ::97] = -32768
    rom[5::97] = 32767
    xf[0, 5::97] = 32767
    xf[1, 5::97] = -32768
    ch.set_window_q15(rom)
    for wm in (0, 1):
        ch.set_window_mode_q15(wm)
        for cmd in (0xB1, 0x00):
            ch.set_filter_mode(cmd)
            ref = oracle.chain_q15(xf, rom, wm, cmd, None, None)
            assert np.array_equal(ch.process_q15(_dev(torch_mod, xf)).cpu().numpy(), ref), (wm, cmd)
    # The same corner through the staging of the cascade kernels (the tests above reach it through the FFT-side window only):
    # a ROM whose -32768 entries sit in a few tiles, full-scale samples with -32768 at those very positions and at others,
    # all three cascade kernels, stream-ordered and with two launches in flight, time series and frames
    rom2 = rng.integers(-32767, 32768, size=N).astype(np.int16)
    hot = np.concatenate([np.arange(100, 130), np.arange(5000, 5003), np.arange(8191, 8194), np.arange(16380, 16384)])
    rom2[hot] = -32768
    xg = rng.integers(-32768, 32768, size=(5, N)).astype(np.int16)
    xg[:, hot] = -32768
    xg[:, 777::1001] = -32768
    ch.set_window_q15(rom2)
    ch.set_window_mode_q15(0)
    gui = np.array([0, 1, 0, 64, -67, 19, 64, 127, 64, 64, -85, 40], np.int8)
    sos14 = load_golden("g4_q15_frames.npz")["sos_q14"]
    ch.load_coeffs_q7(gui)
    ch.load_sos_q14(sos14)
    for depth in (1, 2):
        ch.set_overlap(depth)
        for cmd in (0x00, 0xA1, 0xA2):
            ch.set_filter_mode(cmd)
            ref_iq, ref_t = oracle.chain_q15(xg, rom2, 0, cmd, gui, sos14, want_time=True)
            got = ch.process_q15(_dev(torch_mod, xg))
            ch.flush()
            assert np.array_equal(got.cpu().numpy(), ref_iq), (depth, cmd)
            if depth == 1:
                assert np.array_equal(ch.filter_q15(_dev(torch_mod, xg)).cpu().numpy(), ref_t), cmd
    ch.set_overlap(1)
    ch.set_filter_mode(0xB1)
    ch.set_window_q15(None)
    ch.set_window_mode_q15(0)
    assert np.array_equal(ch.process_q15(_dev(torch_mod, x)).cpu().numpy(), oracle.chain_q15(x))


def test_command_byte_stream(ch, torch_mod, oracle):
    """UART bytes drive the path exactly like the RTL front door (rx_filter_coeff + command_control)."""
    from fpga_real_time_fft_analyzer_amd import designer
    q = designer.quantize_coefficients(designer.design_iir_filter("lowpass", 4, 10.0, 20.0, 100.0))
    wire = ch.send_filter_coefficients(designer.two_sections_for_fpga(q))
    assert wire == designer.coefficient_upload_bytes(q)
    assert ch.coeffs_q7().tolist() == [0, 1, 0, 64, -67, 19, 64, 127, 64, 64, -85, 40]
    assert ch.filter_mode == 0xB1                                     # upload does not select the filter
    assert ch.feed_command_bytes(bytes([0xA1])) == 0 and ch.filter_mode == 0xA1
    # coefficient bytes that look like commands are not decoded while busy
    ch.feed_command_bytes(bytes([0xF1] + [0x00, 0xB1, 0xFF, 0x55, 0xA5, 0xF1, 1, 2, 3, 4, 5, 6]))
    assert ch.filter_mode == 0xA1
    assert ch.coeffs_q7().tolist() == [0, -79, -1, 0x55, -91, -15, 1, 2, 3, 4, 5, 6]
    # split delivery, unknown bytes ignored; 0xA5 is the UART read request (imp/sequ2.vhd:216), 0x55 only starts
    # the acquisition (new/command_control.vhd:58-60); the transport bytes are tracked (imp/sequ2.vhd:82-96)
    assert ch.transport == 0xEF                                          # Ethernet after power-on
    ev = ch.feed_command_bytes_ex(bytes([0x42, 0x55, 0xA5, 0xEF, 0xFE, 0xF1, 9, 9]))
    assert (ev.n_start, ev.n_uart_request, ev.n_reset, ev.n_uploads, ev.control_changed, ev.transport) == (1, 1, 0, 0, 0, 0xFE)
    ev = ch.feed_command_bytes_ex(bytes([9] * 10 + [0xA5]))
    assert (ev.n_uploads, ev.control_changed, ev.n_uart_request) == (1, 1, 1)
    assert ch.coeffs_q7().tolist() == [9] * 12 and ch.transport == 0xFE
    assert ch.feed_command_bytes(bytes([0x55, 0xA5, 0xA5])) == 2          # legacy entry point: read requests only
    ch.feed_command_bytes(bytes([0x00]))
    assert ch.filter_mode == 0x00
    ev = ch.feed_command_bytes_ex(bytes([0xFF]))                       # reset
    assert ch.filter_mode == 0xB1 and not ch.coeffs_q7().any()
    assert ev.n_reset == 1 and ev.control_changed == 1 and ch.transport == 0xEF       # imp/sequ2.vhd:85-86
    # and the data path follows: custom mode with the GUI default upload
    ch.feed_command_bytes(wire + bytes([0xA1]))
    x = np.random.default_rng(5).integers(-2048, 2048, size=(2, N)).astype(np.int16)
    ref = oracle.chain_q15(x, None, 0, 0xA1, ch.coeffs_q7(), None)
    assert np.array_equal(ch.process_q15(_dev(torch_mod, x)).cpu().numpy(), ref)


def test_frames_feed_the_reference_decoder_contract(ch, torch_mod):
    from fpga_real_time_fft_analyzer_amd import frames
    g6 = load_golden("g6_frame.npz")
    g4 = load_golden("g4_q15_frames.npz")
    ch.set_filter_mode(0x00)
    iq = ch.process_q15(_dev(torch_mod, g4["x"][2:3]))
    fb = ch.frames_bytes(iq)
    assert len(fb) == 1 and fb[0] == g6["frame"].tobytes()
    assert np.array_equal(frames.decode_mag_16iq_le(fb[0]), g6["mag"])


def test_fxfft_close_to_float_fft_on_gpu(ch, torch_mod):
    rng = np.random.default_rng(1)
    x = rng.integers(-2048, 2048, size=(8, N)).astype(np.int16)
    rom = np.full(N, 32767, np.int16)                                  # ~unity window
    ch.set_window_q15(rom)
    iq = ch.process_q15(_dev(torch_mod, x)).cpu().numpy().astype(np.float64)
    ref = np.fft.fft(x.astype(np.float64), axis=-1) / N
    assert np.abs(iq[..., 0] + 1j * iq[..., 1] - ref).max() <= 7.0


GUI_UPLOAD = np.array([0, 1, 0, 64, -67, 19, 64, 127, 64, 64, -85, 40], np.int8)      # gui.py:159-179, 1186-1192 defaults


@pytest.mark.parametrize("full_scale", [False, True], ids=["12bit", "fullscale"])
@pytest.mark.parametrize("coeffs", ["default", "gui", "wide"])
def test_config4_whole_batch(ch, torch_mod, oracle, full_scale, coeffs):
    """BASELINE config 4 as SURVEY 8(d) defines it: B = 4096, int16 uniform in [-2048, 2047] (seed 2) and a
    second run uniform over the full int16 range, coefficient set (a) the fixed ALPHA/BETA cascade
    (imp/filter_pkg.vhd:54-68, mode 0x00) and (b) the GUI's default upload (gui.py:1186-1192, mode 0xA1);
    the window + cascade output AND the IQ frame of EVERY one of the 4096 frames equal the SA integer model
    (oracle/specan_oracle.c; parity unpinned vs the RTL / xfft_0).  Third set: mode 0xA2, all six
    sections of the 12th-order Butterworth of fixture G2 (gui.py:108-157 designs them, :1186-1192 keeps two) in Q2.14
    (fixture G4's `sos_q14`), unsigned-Hann window -- every frame against or_iir_sos_q14."""
    torch = torch_mod
    gen = torch.Generator(device="cuda").manual_seed(2 + int(full_scale))
    B = 4096
    lo, hi = (-32768, 32768) if full_scale else (-2048, 2048)
    x = torch.randint(lo, hi, (B, N), generator=gen, device="cuda", dtype=torch.int32).to(torch.int16)
    ch.reserve(B)
    sos, wm = None, 0
    if coeffs == "gui":
        ch.load_coeffs_q7(GUI_UPLOAD)
        cmd, c12 = 0xA1, GUI_UPLOAD
    elif coeffs == "wide":
        sos, wm = load_golden("g4_q15_frames.npz")["sos_q14"], 1
        assert sos.shape == (6, 6)
        ch.load_sos_q14(sos)
        ch.set_window_mode_q15(wm)
        cmd, c12 = 0xA2, None
    else:
        cmd, c12 = 0x00, None
    ch.set_filter_mode(cmd)
    iq = ch.process_q15(x).cpu().numpy()
    tm = ch.filter_q15(x).cpu().numpy()
    xh = x.cpu().numpy()
    for a in range(0, B, 512):                                         # the oracle in slices: bounded host memory
        ref_iq, ref_t = oracle.chain_q15(xh[a:a + 512], None, wm, cmd, c12, sos, want_time=True)
        bad_t = np.nonzero((tm[a:a + 512] != ref_t).any(axis=1))[0]
        bad = np.nonzero((iq[a:a + 512] != ref_iq).reshape(ref_iq.shape[0], -1).any(axis=1))[0]
        assert bad_t.size == 0, f"time series differs in frames {a + bad_t[:8]}"
        assert bad.size == 0, f"IQ differs in frames {a + bad[:8]}"
    # frame independence on the device: a permuted batch gives the permuted output
    perm = torch.randperm(B, generator=gen, device="cuda")
    iq2 = ch.process_q15(x[perm].contiguous())
    assert np.array_equal(iq2.cpu().numpy(), iq[perm.cpu().numpy()])


def test_batch_beyond_two_gib(ch, torch_mod):
    """36 000 frames: the IQ output is 2.36 GB (offsets pass 2^31), the filter workspace 1.18 GB."""
    torch = torch_mod
    gen = torch.Generator(device="cuda").manual_seed(4)
    B = 36000
    x = torch.randint(-2048, 2048, (B, N), generator=gen, device="cuda", dtype=torch.int32).to(torch.int16)
    idx = torch.tensor([0, 1, 16383, 16384, 32767, 32768, B - 1], device="cuda")
    ch.load_sos_q14(load_golden("g4_q15_frames.npz")["sos_q14"])
    for cmd in (0x00, 0xB1, 0xA2):
        ch.set_filter_mode(cmd)
        iq = ch.process_q15(x)
        assert torch.equal(iq[idx], ch.process_q15(x[idx].contiguous()))
        del iq


def test_virtual_fpga_drops_frames_computed_ahead_on_any_control_change(chain_cls, oracle):
    """Frames are computed `batch` acquisitions at a time.  A control change made directly on the exposed chain (window
    ROM, window mode, coefficients, filter mode) -- not only through command bytes -- must not let frames of the old
    settings out: the next frame served already uses the new ones (loadable window ROM: new/hann.vhd:5-6)."""
    from fpga_real_time_fft_analyzer_amd.virtual_fpga import VirtualFpga
    rng = np.random.default_rng(18)
    src = rng.integers(-2048, 2048, size=(8, N)).astype(np.int16)
    taken = []

    def source(n):
        i = len(taken)
        taken.extend(range(i, i + n))
        return src[[j % 8 for j in range(i, i + n)]]

    fpga = VirtualFpga(source, device=0, batch=4)
    fpga.write(bytes([0xB1, 0x55]))                                     # bypass, Ethernet streaming
    first = fpga.read_datagrams(1)
    assert len(first) == 64 and len(taken) == 4                        # three more frames wait in the queue
    rom = rng.integers(-32768, 32768, N).astype(np.int16)
    fpga.chain.set_window_q15(rom)                                      # directly on the chain
    dg = fpga.read_datagrams(1)
    got = b"".join(d[1:] for d in dg)
    assert len(taken) == 8                                              # the queue was dropped, a new batch acquired
    want = oracle.chain_q15(src[4 % 8][None, :], rom, 0, 0xB1, None, None)[0].astype("<i2").tobytes()
    assert got == want
    fpga.chain.set_window_mode_q15(1)                                   # the unsigned-Hann reading of the same ROM
    got = b"".join(d[1:] for d in fpga.read_datagrams(1))
    want = oracle.chain_q15(src[8 % 8][None, :], rom, 1, 0xB1, None, None)[0].astype("<i2").tobytes()
    assert got == want and len(taken) == 12
    fpga.chain.close()


def test_virtual_fpga_uart_and_udp(chain_cls, torch_mod, oracle):
    """N1/N2: the board as gui.py sees it -- command bytes in, frames / datagrams out, sequenced like
    imp/sequ2.vhd: Ethernet after reset, 0x55 arms the UART and the first 0xA5 starts a CONTINUOUS byte stream
    (what gui.py:529-549 sends and :616-689 slices), frames computed ahead are dropped when the control state
    changes, coefficient bytes are never commands."""
    import socket
    import threading
    from fpga_real_time_fft_analyzer_amd import designer, frames
    from fpga_real_time_fft_analyzer_amd.virtual_fpga import VirtualFpga, VirtualSerial
    rng = np.random.default_rng(8)
    src_frames = rng.integers(-2048, 2048, size=(6, N)).astype(np.int16)
    served = []

    def source(n):
        i = len(served)
        served.extend(range(i, i + n))
        return src_frames[[j % 6 for j in range(i, i + n)]]

    def ref_frame(idx, cmd, c12=None):
        return oracle.chain_q15(src_frames[idx % 6][None, :], None, 0, cmd, c12, None)[0].astype("<i2").tobytes()

    fpga = VirtualFpga(source, device=0, batch=3)
    ser = VirtualSerial.factory(fpga)("COM5", 230400, timeout=0.001, writeTimeout=0.5, exclusive=True)   # gui.py:468-477
    assert ser.is_open and fpga.transport == "ETHERNET" and not fpga.started                             # sequ2.vhd:85-86
    ser.reset_input_buffer()
    ser.reset_output_buffer()
    q = designer.two_sections_for_fpga(designer.quantize_coefficients(designer.design_iir_filter("lowpass", 4, 10.0, 20.0, 100.0)))
    c12 = np.array(q, np.int8).reshape(12)
    # UART session exactly as the GUI runs it: mode byte, upload, select custom, 0x55, (100 ms), 0xA5
    ser.write(bytes([0xFE]))
    ser.write(designer.coefficient_upload_bytes(q) + bytes([0xA1]))
    ser.write(bytes([0x55]))
    ser.flush()
    assert fpga.uart_state == "IDLE2" and ser.in_waiting == 0 and ser.read(4096) == b""                 # armed, silent
    ser.write(bytes([0xA5]))
    assert fpga.uart_state == "STREAM"
    got = bytearray()
    while len(got) < 2 * 65536:                                                                        # gui.py:616-627
        n = min(ser.in_waiting, 4096)
        assert n > 0
        got += ser.read(n)
    assert bytes(got[:65536]) == ref_frame(0, 0xA1, c12) and bytes(got[65536:2 * 65536]) == ref_frame(1, 0xA1, c12)
    # a control change drops what was computed ahead (batch = 3: frame 2 is pending with the old filter):
    # the next frame on the wire is a NEW acquisition filtered with the new setting
    ser.reset_input_buffer()
    ser.write(bytes([0xB1]))
    nxt = ser.read(65536)
    assert len(nxt) == 65536 and nxt == ref_frame(3, 0xB1)
    # coefficient bytes equal to command values (0xA5, 0xEF, 0xFF, 0x55) are data, not commands
    ser.reset_input_buffer()
    ser.write(bytes([0xF1] + [0xA5, 0xEF, 0xFF, 0x55] * 3))
    assert fpga.transport == "UART" and fpga.uart_state == "STREAM"
    assert list(fpga.chain.coeffs_q7().view(np.uint8)) == [0xA5, 0xEF, 0xFF, 0x55] * 3
    # Ethernet: switching idles both machines until a new 0x55; then 64 datagrams per frame reassemble
    ser.write(bytes([0xEF]))
    assert fpga.transport == "ETHERNET" and not fpga.started and fpga.read_datagrams() == [] and ser.in_waiting == 0
    ser.write(bytes([0xB1, 0x55]))
    dg = fpga.read_datagrams(2)
    assert len(dg) == 128 and all(len(d) == 1025 for d in dg) and [d[0] for d in dg[:64]] == list(range(64))
    asm = FrameCollector()
    out = [fr for fr in (asm.add(d, 0) for d in dg) if fr is not None]
    assert len(out) == 2 and all(len(fr) == 65536 for fr in out)
    # UDP serve loop against a plain socket on the GUI's port layout (any free port here), capped frame rate
    rx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    rx.bind(("127.0.0.1", 0))
    rx.settimeout(5.0)
    th = threading.Thread(target=fpga.serve_udp, kwargs=dict(addr=rx.getsockname(), n_frames=3, fps_limit=200.0))
    t0 = __import__("time").monotonic()
    th.start()
    asm, got_frames = FrameCollector(), []
    while len(got_frames) < 3:
        fr = asm.add(rx.recv(2048), 0)
        if fr is not None:
            got_frames.append(fr)
    th.join(timeout=10)
    rx.close()
    assert __import__("time").monotonic() - t0 >= 2 / 200.0 and all(frames.decode_mag_16iq_le(fr).shape == (N,) for fr in got_frames)
    # reset: Ethernet, idle, bypass, coefficients cleared
    ser.write(bytes([0xFE, 0x55, 0xFF]))
    assert fpga.transport == "ETHERNET" and not fpga.started and fpga.chain.filter_mode == 0xB1
    assert not fpga.chain.coeffs_q7().any()
    ser.close()
    assert not ser.is_open
    with pytest.raises(OSError):
        ser.write(b"\x55")
    fpga.close()


def test_ingest_double_buffer_feeds_the_chain(ch, torch_mod, oracle):
    """N3: framed host samples -> pinned double buffer -> device -> path, bit-exact per batch."""
    from fpga_real_time_fft_analyzer_amd.ingest import DeviceFeeder, FrameCutter
    rng = np.random.default_rng(12)
    stream = rng.integers(-2048, 2048, size=5 * N + 100).astype(np.int16)
    fc = FrameCutter()
    batches = [fc.push(stream[:2 * N + 7]), fc.push(stream[2 * N + 7:4 * N]), fc.push(stream[4 * N:])]
    batches = [b for b in batches if b.shape[0]]
    assert sum(b.shape[0] for b in batches) == 5
    feeder = DeviceFeeder(0, max_batch=8)
    outs = [ch.process_q15(xd).cpu().numpy() for xd in feeder.feed(batches)]
    ref = oracle.chain_q15(stream[:5 * N].reshape(5, N))
    assert np.array_equal(np.concatenate(outs), ref)
    # the same int16 batches into the FLOAT chain (sa_process_f32_i16: no conversion pass on the device)
    from scipy import signal
    sos = signal.butter(6, 0.25, output="sos")
    ch.load_sos(sos)
    ch.set_filter_mode(0xA1)
    fouts = [ch.process_f32(xd).cpu().numpy() for xd in DeviceFeeder(0, max_batch=8).feed(batches)]
    xf = (stream[:5 * N].reshape(5, N).astype(np.float32) / np.float32(2048.0)).astype(np.float32)
    _, _, mag = oracle.chain_fp(xf, sos)
    got = np.concatenate(fouts)
    assert (np.abs(got - mag).max(axis=1) / np.abs(mag).max(axis=1)).max() <= 1e-5
    ch.set_filter_mode(0xB1)
