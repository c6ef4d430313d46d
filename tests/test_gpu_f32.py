"""GPU: float path parity, through the C ABI, against the oracle and the golden fixtures.

Gate: max-norm relative error per frame <= 1e-5 on the spectrum (the output of the
path) against scipy.signal.sosfilt + numpy.fft.rfft evaluated in float64 on the same float32 inputs.
"""
import os

import numpy as np
import pytest

from conftest import N, load_golden, rel_maxnorm

pytestmark = pytest.mark.gpu
TOL = 1e-5


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


def synth(B, seed):
    rng = np.random.default_rng(seed)
    n = np.arange(N)
    fb = rng.uniform(0.01, 0.45, size=B)
    return (0.8 * np.sin(2 * np.pi * fb[:, None] * n[None, :]) + 0.05 * rng.standard_normal((B, N))).astype(np.float32)


def test_native_library_is_loaded(ch):
    """The HIP path is the one that runs: the in-tree .so is mapped into this process."""
    maps = open("/proc/self/maps").read()
    assert "libspecan_hip.so" in maps


def test_build_then_smoke_in_one_process():
    """__graft_entry__.build() maps the library before anything imported torch; smoke() must still find the
    GPU (torch's wheel bundles its own HIP runtime: two runtimes in one process see no device)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import __graft_entry__ as g; from fpga_real_time_fft_analyzer_amd import abi; "
            "assert abi.lib().sa_abi_version() == 4; g.smoke()")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "smoke ok" in r.stdout


def test_config1_tone_full_chain(ch, torch_mod, oracle):
    g = load_golden("g2_config1.npz")
    x = _dev(torch_mod, g["x_f32"][None, :])
    ch.load_sos(g["sos"])
    ch.set_filter_mode(0xA1)
    spec = ch.process_f32(x, out_kind="spec_half").cpu().numpy()
    assert np.abs(spec[0] - g["X"]).max() <= TOL * np.abs(g["X"]).max()
    mag = ch.process_f32(x, out_kind="mag_full").cpu().numpy()
    ref = np.abs(g["X"])
    assert rel_maxnorm(mag[:, :N // 2 + 1], ref[None, :]) <= TOL
    assert np.array_equal(mag[0, N // 2 + 1:], mag[0, 1:N // 2][::-1])       # mirrored upper half
    # time series: the tone is 48 dB into the stop band, float32 resolves max|y| only to ~2.5e-5
    # (see tests/test_oracle_golden.py::test_config1_plumbing); bound = 2x a sequential f32 evaluation
    y = ch.process_f32(x, out_kind="time").cpu().numpy()
    xw32 = (g["x_f32"].astype(np.float64) * oracle.hann_f64()).astype(np.float32)
    seq = rel_maxnorm(oracle.sosfilt_f32_c(g["sos"], xw32)[None, :], g["y"][None, :])
    assert rel_maxnorm(y, g["y"][None, :]) <= max(TOL, 2 * seq)
    # bypass
    ch.set_filter_mode(0xB1)
    spec = ch.process_f32(x, out_kind="spec_half").cpu().numpy()
    assert np.abs(spec[0] - g["X_bypass"]).max() <= TOL * np.abs(g["X_bypass"]).max()


def test_g3_golden_frames(ch, torch_mod):
    g = load_golden("g3_fp32_frames.npz")
    x = _dev(torch_mod, g["x"])
    mag = ch.process_f32(x, out_kind="mag_half").cpu().numpy()               # power-on mode = bypass
    assert rel_maxnorm(mag, g["mag_bypass"]) <= TOL
    ch.load_sos(g["sos"])
    ch.set_filter_mode(0xA1)
    mag = ch.process_f32(x, out_kind="mag_half").cpu().numpy()
    assert rel_maxnorm(mag, g["mag_full"]) <= TOL
    y = ch.process_f32(x, out_kind="time").cpu().numpy()
    assert rel_maxnorm(y, g["y_full"]) <= TOL


@pytest.mark.parametrize("B", [1, 3, 64, 256, 512, 513])
def test_bypass_vs_oracle(ch, torch_mod, oracle, B):
    """BASELINE config 2 (B=256) and ragged batches, IIR bypassed.  Batches up to 512 frames take the one-round
    stage-in (a whole frame in a 64 KiB LDS image, two workgroups per CU), larger ones the two half-frame rounds:
    512 and 513 sit on either side of the switch, and the two forms must agree bit for bit on the same frames."""
    x = synth(B, seed=B)
    _, X, mag = oracle.chain_fp(x, None)
    xd = _dev(torch_mod, x)
    got = ch.process_f32(xd, out_kind="mag_full").cpu().numpy()
    assert rel_maxnorm(got, mag) <= TOL
    spec = ch.process_f32(xd, out_kind="spec_half").cpu().numpy()
    assert rel_maxnorm(np.abs(spec - X), np.abs(X)) <= 1.0 and np.abs(spec - X).max(axis=1).max() <= TOL * np.abs(X).max()
    y = ch.process_f32(xd, out_kind="time").cpu().numpy()
    assert rel_maxnorm(y, x.astype(np.float64) * oracle.hann_f64()) <= 1e-6
    if B == 513:                                    # the first 512 frames again, now through the small-batch form
        for kind in ("mag_full", "mag_half", "spec_half"):
            big = ch.process_f32(xd, out_kind=kind)
            small = ch.process_f32(xd[:512].contiguous(), out_kind=kind)
            assert torch_mod.equal(big[:512], small), kind


@pytest.mark.parametrize("B", [1, 5, 128])
def test_full_chain_vs_oracle(ch, torch_mod, oracle, B):
    g = load_golden("g2_config1.npz")
    x = synth(B, seed=100 + B)
    y_ref, X, mag = oracle.chain_fp(x, g["sos"])
    ch.load_sos(g["sos"])
    ch.set_filter_mode(0xA1)
    xd = _dev(torch_mod, x)
    got = ch.process_f32(xd).cpu().numpy()
    assert rel_maxnorm(got, mag) <= TOL
    y = ch.process_f32(xd, out_kind="time").cpu().numpy()
    assert rel_maxnorm(y, y_ref) <= TOL


@pytest.mark.parametrize("kind,ft,order", [("cheby1", "highpass", 12), ("ellip", "bandpass", 6),
                                           ("bessel", "lowpass", 8), ("butter", "lowpass", 2), ("cheby2", "bandstop", 4)])
def test_other_filter_families(ch, torch_mod, oracle, kind, ft, order):
    from fpga_real_time_fft_analyzer_amd import designer
    sos = designer.design_iir_filter(ft, order, 10.0, 20.0, 100.0, kind=kind)[:6]
    x = synth(4, seed=7)
    _, X, mag = oracle.chain_fp(x, sos)
    ch.load_sos(sos)
    ch.set_filter_mode(0xA1)
    got = ch.process_f32(_dev(torch_mod, x)).cpu().numpy()
    assert rel_maxnorm(got, mag) <= TOL


def test_long_memory_filter_uses_every_scan_level(ch, torch_mod, oracle):
    """A narrow low-pass (poles at radius ~0.996) keeps a state alive for thousands of samples: no scan level
    may be skipped and the second-level scan over the 16 rows runs.  Checked against the float64 oracle; the
    bound is the error a *sequential* float32 sosfilt makes on the same input (such poles amplify rounding)."""
    from scipy import signal
    from test_host_logic import _parse_plan
    from fpga_real_time_fft_analyzer_amd.chain import iir_plan_from_sos
    sos = signal.butter(4, 0.002, output="sos")
    _, secs, _, _ = _parse_plan(iir_plan_from_sos(sos))
    assert all(int(np.asarray(secs[i][0][5:6]).view(np.int32)[0]) == 0 for i in range(2))     # nothing skipped
    rng = np.random.default_rng(5)
    n = np.arange(N)
    x = (0.5 * np.sin(2 * np.pi * 0.0004 * n)[None, :] + 0.1 * rng.standard_normal((3, N))).astype(np.float32)
    y64, X, mag = oracle.chain_fp(x, sos)
    ch.load_sos(sos)
    ch.set_filter_mode(0xA1)
    got_t = ch.process_f32(_dev(torch_mod, x), out_kind="time").cpu().numpy()
    got_m = ch.process_f32(_dev(torch_mod, x)).cpu().numpy()
    # sequential float32 reference of the same recurrence
    hann = oracle.hann_f64().astype(np.float32)
    seq = np.stack([oracle.sosfilt_f32_c(sos, row) for row in (x * hann).astype(np.float32)])
    seq_err = np.abs(seq - y64).max() / np.abs(y64).max()
    err_t = np.abs(got_t - y64).max() / np.abs(y64).max()
    assert err_t <= max(1e-5, 4 * seq_err), (err_t, seq_err)
    assert rel_maxnorm(got_m, mag) <= max(TOL, 4 * seq_err)


def test_random_designs(ch, torch_mod):
    """160 random cascades (five scipy families, all four band types, orders 1..12, hand-made sections with
    zero / unit / negative numerators) x random tones + noise.  The accuracy statement of include/specan.h, in one
    norm (spectrum error relative to the spectrum peak, for the GPU and for the sequential float32 evaluation alike):
    within 1e-5 of the float64 oracle, or -- when the filter removes the dominant input or has poles so close to the
    unit circle that float32 itself runs out -- within 4x of what a *sequential* float32 sosfilt achieves on the same
    data.  Over 4500 designs (seeds 7 / 11 / 23, profiles/r4_fuzz.txt) two exceed that factor; they and a third outlier
    are pinned here by name with the bounds they meet, so that a change of the cascade's algebra that moves them shows."""
    from fuzz_parity import sweep
    for err, att, seq_err, label in sweep(ch, seed=7, ncases=160):
        assert err <= max(TOL, 4 * seq_err), (label, err, att, seq_err)
    # named outliers: (seed, case) -> bound on the spectrum error; measured 8.2e-5 (4.0x sequential), 4.0e-5 (3.4x), 2.4e-4 (15.6x)
    pinned = {7: {752: 1.2e-4, 1203: 6e-5}, 11: {929: 3.5e-4}}
    for seed, want in pinned.items():
        got = sweep(ch, seed=seed, ncases=max(want) + 1, only=set(want))
        assert len(got) == len(want)
        for (err, att, seq_err, label), case in zip(got, sorted(want)):
            assert f"case {case}]" in label
            assert TOL < err <= want[case], (label, err, seq_err)      # still an outlier, still within its bound


def test_default_mode_is_the_rtl_taps_as_reals(ch, torch_mod, oracle):
    """Filter 0x00 on the float path = ALPHA/BETA taps /128 (imp/filter_pkg.vhd:54-68), 3x each."""
    a = [14 / 128, 0, -14 / 128, 1, 21 / 128, 107 / 128]
    b = [15 / 128, 0, -15 / 128, 1, -21 / 128, 107 / 128]
    sos = np.array([a, b, a, b, a, b])
    x = synth(3, seed=11)
    _, _, mag = oracle.chain_fp(x, sos)
    ch.set_filter_mode(0x00)
    got = ch.process_f32(_dev(torch_mod, x)).cpu().numpy()
    assert rel_maxnorm(got, mag) <= TOL
    # the same through the q7 upload in custom mode
    ch.load_coeffs_q7([-14, 0, 14, 107, 21, 127, -15, 0, 15, 107, -21, 127])
    ch.set_filter_mode(0xA1)
    got2 = ch.process_f32(_dev(torch_mod, x)).cpu().numpy()
    assert np.array_equal(got, got2)


def test_custom_window_and_restore(ch, torch_mod, oracle):
    x = synth(2, seed=21)
    w = np.blackman(N).astype(np.float32)
    ch.set_window_f32(w)
    _, _, mag = oracle.chain_fp(x, None, hann=w.astype(np.float64))
    got = ch.process_f32(_dev(torch_mod, x)).cpu().numpy()
    assert rel_maxnorm(got, mag) <= TOL
    ch.set_window_f32(None)
    _, _, mag = oracle.chain_fp(x, None)
    got = ch.process_f32(_dev(torch_mod, x)).cpu().numpy()
    assert rel_maxnorm(got, mag) <= TOL


@pytest.mark.parametrize("wname", ["blackman", "hamming"])
def test_table_and_fitted_windows_through_the_iir(ch, torch_mod, oracle, wname):
    """The window is a loadable ROM (new/hann.vhd:5-6, new/hann8192.vhd:34-41).  Blackman does not fit
    a0 - a1 cos: the IIR kernels read the gain-folded table (WINGEN = false x NSEC 6 / 4 / 2); Hamming fits with
    a0 != a1: the in-place generator with other constants than Hann's.  12th-order Butterworth in mode 0xA1, all
    output kinds; then shorter cascades; then the default window again."""
    torch = torch_mod
    g = load_golden("g2_config1.npz")
    w = (np.blackman(N) if wname == "blackman" else np.hamming(N)).astype(np.float32)
    x = synth(5, seed=77)
    xd = _dev(torch, x)
    ch.set_window_f32(w)
    ch.load_sos(g["sos"])
    ch.set_filter_mode(0xA1)
    y, X, mag = oracle.chain_fp(x, g["sos"], hann=w.astype(np.float64))
    H = N // 2 + 1
    assert rel_maxnorm(ch.process_f32(xd, out_kind="mag_full").cpu().numpy(), mag) <= TOL
    assert rel_maxnorm(ch.process_f32(xd, out_kind="mag_half").cpu().numpy(), mag[:, :H]) <= TOL
    spec = ch.process_f32(xd, out_kind="spec_half").cpu().numpy()
    assert (np.abs(spec - X).max(axis=1) <= TOL * np.abs(X).max(axis=1)).all()
    seq = max(rel_maxnorm(oracle.sosfilt_f32_c(g["sos"], (x[i].astype(np.float64) * w).astype(np.float32))[None, :], y[i][None, :])
              for i in range(x.shape[0]))
    assert rel_maxnorm(ch.process_f32(xd, out_kind="time").cpu().numpy(), y) <= max(TOL, 2 * seq)
    # 4- and 2-section cascades (the other compiled section counts) with the same window
    from scipy import signal
    for order, kind in ((8, "lowpass"), (4, "highpass"), (3, "lowpass")):
        sos = signal.butter(order, 0.3, kind, output="sos")
        ch.load_sos(sos)
        _, _, m = oracle.chain_fp(x, sos, hann=w.astype(np.float64))
        assert rel_maxnorm(ch.process_f32(xd, out_kind="mag_half").cpu().numpy(), m[:, :H]) <= TOL, (order, kind)
    # a non-unit-numerator cascade (band-pass: b = [1, 0, -1]) on the same window
    sos = signal.butter(4, [0.1, 0.3], "bandpass", output="sos")
    ch.load_sos(sos)
    _, _, m = oracle.chain_fp(x, sos, hann=w.astype(np.float64))
    assert rel_maxnorm(ch.process_f32(xd, out_kind="mag_half").cpu().numpy(), m[:, :H]) <= TOL
    # default window again (Hann: the generator path with a0 = a1 = 0.5)
    ch.set_window_f32(None)
    ch.load_sos(g["sos"])
    _, _, mag = oracle.chain_fp(x, g["sos"])
    assert rel_maxnorm(ch.process_f32(xd, out_kind="mag_half").cpu().numpy(), mag[:, :H]) <= TOL


@pytest.mark.parametrize("depth", [2, 3])
def test_overlapped_launches_contract(ch, torch_mod, oracle, depth):
    """sa_set_overlap: consecutive calls may run beside each other (frames are independent: every frame
    starts from a zero filter state, SURVEY quirk Q5 / new/filter_iir_cust.vhd:142-146).  Outputs are bit-identical to the ordered mode,
    the results of call k are visible on the caller's stream after call k+depth-1 or after flush(), control-plane
    calls stay ordered between calls, and capture is refused."""
    torch = torch_mod
    g = load_golden("g2_config1.npz")
    ch.load_sos(g["sos"])
    ch.set_filter_mode(0xA1)
    xs = [_dev(torch, synth(64, seed=100 + i)) for i in range(6)]
    ref = [ch.process_f32(x).clone() for x in xs]                 # ordered mode
    ch.set_filter_mode(0xB1)
    ref_bypass = ch.process_f32(xs[5]).clone()
    ch.set_filter_mode(0xA1)
    torch.cuda.synchronize()
    assert ch.overlap == 1
    ch.set_overlap(depth)
    assert ch.overlap == depth
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        outs = [torch.zeros_like(ref[0]) for _ in xs]
        seen = []
        for k in range(5):
            ch.process_f32(xs[k], out=outs[k])
            if k >= depth - 1:                                   # call k-depth+1 is joined now: read it on the stream
                seen.append((k - depth + 1, outs[k - depth + 1].clone()))
        ch.set_filter_mode(0xB1)                                 # control plane between overlapped calls
        ch.process_f32(xs[5], out=outs[5])
        ch.flush()
        tail = [o.clone() for o in outs]
    s.synchronize()
    for k, got in seen:
        assert torch.equal(got, ref[k]), k
    for k in range(5):
        assert torch.equal(tail[k], ref[k]), k
    assert torch.equal(tail[5], ref_bypass)
    # not capturable in this mode; ordered mode still is
    from fpga_real_time_fft_analyzer_amd.abi import SpecanError
    graph = torch.cuda.CUDAGraph()
    with pytest.raises(SpecanError):
        with torch.cuda.graph(graph):
            ch.process_f32(xs[0], out=outs[0])
    torch.cuda.synchronize()
    ch.set_overlap(1)
    ch.set_filter_mode(0xA1)
    assert torch.equal(ch.process_f32(xs[0]), ref[0])


def test_overlap_streams_are_picked_to_run_side_by_side(chain_cls, torch_mod):
    """The runtime maps streams onto four hardware queues and two streams on one queue run in order (tools/ubench/
    stream_pairs.hip): a handle whose internal streams shared a queue lost instead of gained in overlap mode.
    The library probes for streams that do overlap -- among themselves in sa_set_overlap, against the caller's stream
    in the first overlapped call; here ten handles are created with foreign streams in between (which moves the
    mapping along) and every one of them must end up with streams that run beside each other and the caller's."""
    keep, foreign = [], []
    x = torch_mod.zeros((4, N), dtype=torch_mod.float32, device="cuda")
    for i in range(10):
        foreign += [torch_mod.cuda.Stream() for _ in range(i % 3)]
        c = chain_cls(0)
        c.set_overlap(2 if i % 4 else 3)
        c.process_f32(x)                      # the first overlapped call fits the streams to the caller's stream
        c.flush()
        keep.append(c)
    torch_mod.cuda.synchronize()

    def side_by_side(c):
        """The probe is a wall-clock measurement (include/specan.h: best effort on a busy GPU): one repeat on a quiet
        device before it counts as a failure."""
        if c.overlap_streams_side_by_side():
            return True
        torch_mod.cuda.synchronize()
        return c.overlap_streams_side_by_side()
    assert all(side_by_side(c) for c in keep)
    # ... and to another caller stream when the calls move there
    side = torch_mod.cuda.Stream()
    with torch_mod.cuda.stream(side):
        for c in keep[:4]:
            c.process_f32(x)
            c.flush()
        side.synchronize()
        assert all(side_by_side(c) for c in keep[:4])
    for c in keep:
        c.close()


def test_overlap_mode_keeps_lent_tensors_alive(ch, torch_mod):
    """Overlap mode runs a call on a stream torch's allocator knows nothing about; the wrapper must hold the call's
    input and output until the caller's stream has joined it (depth-1 further calls, flush, or a change of depth).
    Temporaries as inputs and dropped results, with allocations of the same size in between (which the allocator
    would serve from a freed block at once), must still give the ordered mode's results."""
    torch = torch_mod
    g = load_golden("g2_config1.npz")
    ch.load_sos(g["sos"])
    ch.set_filter_mode(0xA1)
    gen = torch.Generator(device="cuda").manual_seed(5)
    xs = [torch.randn(96, N, generator=gen, device="cuda") for _ in range(6)]
    ref = [ch.process_f32(x).clone() for x in xs]
    torch.cuda.synchronize()
    for depth in (2, 3):
        ch.set_overlap(depth)
        assert len(ch._lent) == 0
        outs = []
        for k, x in enumerate(xs):
            o = ch.process_f32(x.clone())                    # the input is a temporary
            assert len(ch._lent) == min(k + 1, depth - 1)
            scribble = torch.full_like(x, float("nan"))      # same size: would land in a block freed too early
            del scribble
            if k % 2:
                outs.append((k, o))
            del o                                            # every other result is dropped at once
        ch.flush()
        assert len(ch._lent) == 0
        for k, o in outs:
            assert torch.equal(o, ref[k]), (depth, k)
    ch.set_overlap(1)
    assert len(ch._lent) == 0


def test_launch_timing_ring(ch, torch_mod):
    """sa_set_profiling: the device time of each stream-ordered call from the launch's own start / stop events.  The
    times are positive, no longer than the wall time of the train they were part of, the ring keeps the last n calls,
    results do not change, overlap mode and timing exclude each other, and the integer chain reports cascade + FFT."""
    import time
    from fpga_real_time_fft_analyzer_amd.abi import SpecanError
    torch = torch_mod
    g = load_golden("g2_config1.npz")
    ch.load_sos(g["sos"])
    ch.set_filter_mode(0xA1)
    gen = torch.Generator(device="cuda").manual_seed(6)
    x = torch.randn(512, N, generator=gen, device="cuda")
    out = torch.empty_like(x)
    ref = ch.process_f32(x).clone()
    ch.set_profiling(8)
    assert ch.profile_read(8) == []
    for _ in range(3):
        ch.process_f32(x, out=out)
    assert len(ch.profile_read(8)) == 3
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(12):
        ch.process_f32(x, out=out)
    ms = ch.profile_read(100)
    wall_ms = (time.perf_counter() - t0) * 1e3
    assert len(ms) == 8 and all(0.0 < v < wall_ms for v in ms)
    assert sum(ms) <= wall_ms * 1.02                         # a kernel cannot take longer than its step
    assert len(ch.profile_read(2)) == 2
    assert torch.equal(out, ref)
    with pytest.raises(SpecanError):
        ch.set_overlap(2)
    # the integer chain: one timed call covers the cascade and the FFT; the cascade alone is shorter
    xq = torch.randint(-2048, 2048, (256, N), generator=gen, device="cuda", dtype=torch.int32).to(torch.int16)
    ch.set_filter_mode(0x00)
    ch.process_q15(xq)
    ch.process_q15(xq)
    both = ch.profile_read(1)[0]
    ch.filter_q15(xq)
    ch.filter_q15(xq)
    casc = ch.profile_read(1)[0]
    assert 0.0 < casc < both
    # captured calls are not timed and leave the ring alone
    ch.set_filter_mode(0xA1)
    before = ch.profile_read(8)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            ch.process_f32(x, out=out)
        graph.replay()
    s.synchronize()
    assert ch.profile_read(8) == before
    ch.set_profiling(0)
    with pytest.raises(SpecanError):
        ch.profile_read(1)
    ch.set_overlap(2)
    with pytest.raises(SpecanError):
        ch.set_profiling(4)
    ch.set_overlap(1)
    assert torch.equal(ch.process_f32(x), ref)


@pytest.mark.parametrize("scale", [1.0 / 2048.0, 3.1e-4])
def test_int16_samples_take_the_float_path_bit_for_bit(ch, torch_mod, oracle, scale):
    """sa_process_f32_i16: the ADC's int16 samples (imp/dsp_system_top.vhd:435) converted and scaled in the stage-in.
    x = float(sample) * scale is rounded once, so every output must EQUAL what sa_process_f32 gives on the frames
    converted by the caller -- every filter mode and padded section count, cosine and table window, every output
    kind -- and, through it, the float64 oracle within the float tolerance."""
    from scipy import signal
    torch = torch_mod
    rng = np.random.default_rng(17)
    xi = rng.integers(-2048, 2048, size=(5, N)).astype(np.int16)
    xi[1] = rng.integers(-32768, 32768, size=N).astype(np.int16)          # full-scale samples
    xi[2, :4] = [-32768, 32767, -1, 0]
    d_i = _dev(torch, xi)
    d_f = (d_i.to(torch.float32) * np.float32(scale)).contiguous()            # one rounding, as the kernel does it
    cascades = [None, signal.butter(12, 0.2, output="sos"), signal.cheby1(7, 1.0, 0.3, output="sos"),
                signal.ellip(4, 0.5, 40.0, [0.1, 0.3], btype="bandpass", output="sos")[:3], signal.butter(3, 0.4, output="sos")]
    for win in (None, np.blackman(N).astype(np.float32)):
        if win is not None:
            ch.set_window_f32(win)
        for sos in cascades:
            if sos is None:
                ch.set_filter_mode(0xB1)
            else:
                ch.load_sos(sos)
                ch.set_filter_mode(0xA1)
            for kind in ("mag_full", "mag_half", "spec_half", "time"):
                a = ch.process_f32(d_i, out_kind=kind, scale=scale)
                b = ch.process_f32(d_f, out_kind=kind)
                assert torch.equal(torch.view_as_real(a) if a.is_complex() else a,
                                   torch.view_as_real(b) if b.is_complex() else b), (win is not None, kind, None if sos is None else len(sos))
        ch.set_filter_mode(0x00)                                             # the RTL's default taps as reals
        assert torch.equal(ch.process_f32(d_i, scale=scale), ch.process_f32(d_f))
    ch.set_window_f32(None)
    # and against the float64 oracle (Hann, the headline cascade)
    ch2_sos = cascades[1]
    ch.set_window_f32(None)
    ch.load_sos(ch2_sos)
    ch.set_filter_mode(0xA1)
    xf = (xi.astype(np.float32) * np.float32(scale)).astype(np.float32)
    _, _, mag = oracle.chain_fp(xf, ch2_sos)
    got = ch.process_f32(d_i, scale=scale).cpu().numpy()
    assert rel_maxnorm(got, mag) <= 1e-5
    with pytest.raises(Exception):
        ch.process_f32(d_i, scale=float("nan"))


def test_edge_inputs(ch, torch_mod):
    g = load_golden("g2_config1.npz")
    ch.load_sos(g["sos"])
    ch.set_filter_mode(0xA1)
    z = torch_mod.zeros((2, N), dtype=torch_mod.float32, device="cuda")
    assert not ch.process_f32(z).any()
    e = torch_mod.empty((0, N), dtype=torch_mod.float32, device="cuda")
    assert ch.process_f32(e).shape == (0, N)
    # impulse at n = 8191 with bypass: flat spectrum of height hann[8191]
    ch.set_filter_mode(0xB1)
    imp = torch_mod.zeros((1, N), dtype=torch_mod.float32, device="cuda")
    imp[0, 8191] = 1.0
    m = ch.process_f32(imp).cpu().numpy()
    h = 0.5 * (1 - np.cos(2 * np.pi * 8191 / (N - 1)))
    assert np.abs(m - h).max() <= 1e-5 * h


def test_argument_errors(ch, torch_mod):
    from fpga_real_time_fft_analyzer_amd.abi import SpecanError
    with pytest.raises(SpecanError):
        ch.process_f32(torch_mod.zeros((2, 100), dtype=torch_mod.float32, device="cuda"))
    with pytest.raises(SpecanError):
        ch.process_f32(torch_mod.zeros((2, N), dtype=torch_mod.float64, device="cuda"))
    with pytest.raises(SpecanError):
        ch.process_f32(torch_mod.zeros((2, N), dtype=torch_mod.float32))            # host tensor
    with pytest.raises(SpecanError):
        ch.set_filter_mode(0x42)
    with pytest.raises(SpecanError):
        ch.load_sos(np.zeros((7, 6)))
    with pytest.raises(SpecanError):
        ch.process_f32(torch_mod.zeros((1, N), dtype=torch_mod.float32, device="cuda"), out_kind="bogus")
    ch.set_filter_mode(0xA2)
    with pytest.raises(SpecanError):
        ch.process_f32(torch_mod.zeros((1, N), dtype=torch_mod.float32, device="cuda"))


def test_full_size_properties(ch, torch_mod):
    """BASELINE config 3 size (B=4096): linearity and frame independence, no oracle needed."""
    torch = torch_mod
    g = load_golden("g2_config1.npz")
    ch.load_sos(g["sos"])
    ch.set_filter_mode(0xA1)
    gen = torch.Generator(device="cuda").manual_seed(1)
    B = 4096
    n = torch.arange(N, device="cuda", dtype=torch.float32)
    fb = torch.rand(B, 1, generator=gen, device="cuda") * 0.44 + 0.01
    x = 0.8 * torch.sin(2 * np.pi * fb * n) + 0.05 * torch.randn(B, N, generator=gen, device="cuda")
    s = ch.process_f32(x, out_kind="spec_half")
    # (1) frame independence: any permutation of the batch permutes the output
    perm = torch.randperm(B, generator=gen, device="cuda")
    s2 = ch.process_f32(x[perm].contiguous(), out_kind="spec_half")
    assert torch.equal(s2, s[perm])
    # (2) linearity: chain(a + b) = chain(a) + chain(b) within float32 noise
    a, b = x[:2048], x[2048:]
    sab = ch.process_f32((a + b).contiguous(), out_kind="spec_half")
    err = (sab - (s[:2048] + s[2048:])).abs().amax(dim=1) / sab.abs().amax(dim=1)
    assert float(err.max()) <= 2e-5
    # (3) magnitude output = |spectrum|, mirrored
    m = ch.process_f32(x, out_kind="mag_full")
    assert torch.allclose(m[:, :N // 2 + 1], s.abs(), rtol=2e-6, atol=1e-6)
    assert torch.equal(m[:, N // 2 + 1:], m[:, 1:N // 2].flip(1))


def test_config3_whole_batch(ch, torch_mod, oracle):
    """BASELINE config 3, every frame: B = 4096 frames of the bench distribution (seed 1), custom mode with the
    12th-order Butterworth of G2; ALL 4096 magnitude spectra (the bench's output kind, 16384 bins each) and
    all 4096 half spectra are within 1e-5 (max-norm relative, per frame: SURVEY 8(d)) of
    scipy.signal.sosfilt + numpy.fft.rfft in float64 (oracle.chain_fp)."""
    torch = torch_mod
    g = load_golden("g2_config1.npz")
    ch.load_sos(g["sos"])
    ch.set_filter_mode(0xA1)
    gen = torch.Generator(device="cuda").manual_seed(1)
    B = 4096
    n = torch.arange(N, device="cuda", dtype=torch.float32)
    fb = torch.rand(B, 1, generator=gen, device="cuda") * 0.44 + 0.01
    x = (0.8 * torch.sin(2 * np.pi * fb * n) + 0.05 * torch.randn(B, N, generator=gen, device="cuda")).contiguous()
    mag = ch.process_f32(x).cpu().numpy()
    spec = ch.process_f32(x, out_kind="spec_half").cpu().numpy()
    xh = x.cpu().numpy()
    worst_m = worst_s = 0.0
    for a in range(0, B, 256):                                          # oracle in slices: bounded host memory
        _, X, M = oracle.chain_fp(xh[a:a + 256], g["sos"])
        em = np.abs(mag[a:a + 256] - M).max(axis=1) / np.abs(M).max(axis=1)
        es = np.abs(spec[a:a + 256] - X).max(axis=1) / np.abs(X).max(axis=1)
        assert em.max() <= TOL, f"magnitude of frame {a + int(em.argmax())} off by {em.max():.3e}"
        assert es.max() <= TOL, f"spectrum of frame {a + int(es.argmax())} off by {es.max():.3e}"
        worst_m, worst_s = max(worst_m, float(em.max())), max(worst_s, float(es.max()))
    print(f"config 3 whole batch: worst magnitude error {worst_m:.2e}, worst spectrum error {worst_s:.2e}")


def test_batch_beyond_two_gib(ch, torch_mod):
    """40 000 frames = 2.6 GB in and out: byte offsets pass 2^31 (BASELINE config 5 is 32 768 frames, sharded).
    The last frames of the big batch must equal the same frames processed alone."""
    torch = torch_mod
    g = load_golden("g2_config1.npz")
    ch.load_sos(g["sos"])
    ch.set_filter_mode(0xA1)
    B = 40000
    gen = torch.Generator(device="cuda").manual_seed(3)
    x = torch.empty((B, N), dtype=torch.float32, device="cuda")
    x.normal_(generator=gen)
    out = ch.process_f32(x)
    idx = torch.tensor([0, 1, 32767, 32768, B - 2, B - 1], device="cuda")
    small = ch.process_f32(x[idx].contiguous())
    assert torch.equal(out[idx], small)
    assert torch.isfinite(out[::997]).all()
    del out
    m = ch.process_f32(x, out_kind="mag_half")
    assert torch.equal(m[idx], ch.process_f32(x[idx].contiguous(), out_kind="mag_half"))


def test_side_stream_and_graph_replay(ch, torch_mod, oracle):
    """Calls are asynchronous on the caller's stream and capturable into a HIP graph (no allocation,
    no synchronisation inside sa_process_f32 once the kernels have been used once)."""
    torch = torch_mod
    g = load_golden("g2_config1.npz")
    ch.load_sos(g["sos"])
    ch.set_filter_mode(0xA1)
    x = _dev(torch, synth(8, seed=31))
    ref = ch.process_f32(x).clone()                       # warm-up on the default stream
    out = torch.empty_like(ref)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ch.process_f32(x, out=out)
    s.synchronize()
    assert torch.equal(out, ref)
    graph = torch.cuda.CUDAGraph()
    out.zero_()
    with torch.cuda.graph(graph):
        ch.process_f32(x, out=out)
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    x.copy_(_dev(torch, synth(8, seed=32)))               # new data, same graph
    graph.replay()
    torch.cuda.synchronize()
    _, _, mag = oracle.chain_fp(x.cpu().numpy(), g["sos"])
    assert rel_maxnorm(out.cpu().numpy(), mag) <= TOL


def test_control_plane_refused_during_capture_changes_nothing(ch, torch_mod, oracle):
    """A control-plane call while the capture of the handle's process call is open is refused (SA_ESTATE) BEFORE it
    changes anything: host plan, taps in the kernel arguments and device tables stay those of the old filter and window
    (a half-applied change would pair new taps with old lane matrices: wrong magnitudes, no error)."""
    from fpga_real_time_fft_analyzer_amd.abi import SpecanError
    from scipy import signal
    torch = torch_mod
    g = load_golden("g2_config1.npz")
    ch.load_sos(g["sos"])
    ch.set_filter_mode(0xA1)
    x = _dev(torch, synth(4, seed=41))
    ref = ch.process_f32(x).clone()
    out = torch.empty_like(ref)
    other = signal.butter(6, 0.4, "highpass", output="sos")
    graph = torch.cuda.CUDAGraph()
    refused = []
    with torch.cuda.graph(graph):
        ch.process_f32(x, out=out)
        for call in (lambda: ch.load_sos(other), lambda: ch.set_window_f32(np.blackman(N).astype(np.float32)),
                     lambda: ch.set_filter_mode(0xB1), lambda: ch.load_coeffs_q7([1] * 12), lambda: ch.set_overlap(2)):
            try:
                call()
                refused.append(False)
            except SpecanError as e:
                refused.append(e.code == -4)
    assert refused == [True] * 5
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert ch.filter_mode == 0xA1 and ch.overlap == 1
    # the record of an open capture is sticky: an uncaptured call made on ANOTHER stream meanwhile does not re-open the
    # control plane (the handle used to look at its most recent call only)
    out2 = torch.empty_like(ref)
    side = torch.cuda.Stream()
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph2, capture_error_mode="relaxed"):
        ch.process_f32(x, out=out)
        with torch.cuda.stream(side):
            ch.process_f32(x, out=out2)
        with pytest.raises(SpecanError):
            ch.set_filter_mode(0xB1)
    side.synchronize()
    assert torch.equal(out2, ref) and ch.filter_mode == 0xA1
    assert torch.equal(ch.process_f32(x), ref)                       # outside the capture: still the old filter and window
    ch.load_sos(other)                                               # ... and control-plane calls work again
    _, _, mag = oracle.chain_fp(x.cpu().numpy(), other)
    assert rel_maxnorm(ch.process_f32(x).cpu().numpy(), mag) <= TOL
