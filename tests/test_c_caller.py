"""The C ABI driven from C: tests/c/abi_caller.c (the snippet of INTEGRATION.md section 3 as a real program) is
compiled with gcc against libspecan_hip.so + libamdhip64 and run as a fresh child process.

CPU box: it builds and links (every symbol it uses resolves), and without a GPU sa_create() fails loudly
(exit code 3, "no usable HIP device"), also from eight host threads at once with per-thread error strings.
GPU box: its frames equal the golden digests of G4 (integer model) bit for bit."""
import ctypes
import hashlib
import os
import shutil
import subprocess
import threading

import numpy as np
import pytest

from conftest import N, ROOT, load_golden

PKG = os.path.join(ROOT, "fpga_real_time_fft_analyzer_amd")


@pytest.fixture(scope="module")
def caller(tmp_path_factory, hip_lib_built):
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    exe = str(tmp_path_factory.mktemp("c_caller") / "abi_caller")
    cmd = [gcc, "-O2", "-Wall", "-Werror", "-std=c11", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_caller.c"),
           "-L" + PKG, "-lspecan_hip", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_caller_builds_and_fails_loudly_without_gpu(caller):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by test_c_caller_frames_match_golden")
    r = subprocess.run([caller], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr)
    assert "no usable HIP device" in r.stderr and "no CPU fallback" in r.stderr


def test_create_from_eight_threads_without_gpu(hip_lib_built):
    """sa_create / sa_last_error(NULL) from eight host threads at once (the one-thread-per-GPU model of SURVEY
    8(e)): every thread gets its own clean failure and its own message on a box without a GPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by test_two_threads_two_handles")
    L = hip_lib_built
    out = [None] * 8

    def work(i):
        res = []
        for k in range(50):
            h = ctypes.c_void_p()
            rc = L.sa_create(i % 3 - 1 if k % 2 else 0, ctypes.byref(h))       # also bad device indices
            res.append((rc, bool(h.value), L.sa_last_error(None)))
        out[i] = res
    th = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for res in out:
        assert res is not None
        for rc, got, msg in res:
            assert rc == -3 and not got and b"no usable HIP device" in msg


@pytest.mark.gpu
def test_c_caller_frames_match_golden(caller, tmp_path):
    g = load_golden("g4_q15_frames.npz")
    fin, fco = tmp_path / "in.bin", tmp_path / "coef.bin"
    g["x"].astype("<i2").tofile(fin)
    g["c_gui"].astype(np.int8).tofile(fco)
    for name, mode, coef in (("bypass", "0xB1", None), ("default", "0x00", None), ("gui", "0xA1", str(fco))):
        fout = tmp_path / f"out_{name}.bin"
        cmd = [caller, mode, str(fin), str(fout), "4"] + ([coef] if coef else [])
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        raw = np.fromfile(fout, np.uint8)
        assert raw.size == 4 * 65536
        frames_ = raw.reshape(4, 65536)
        digests = [hashlib.sha256(np.frombuffer(fr.tobytes(), "<i2").reshape(N, 2).tobytes()).hexdigest() for fr in frames_]
        assert digests == list(g[f"iq_{name}_sha"]), name
        if name == "default":
            assert np.array_equal(np.frombuffer(frames_[2].tobytes(), "<i2").reshape(N, 2), g["iq_default_f2"])


@pytest.mark.gpu
def test_two_threads_two_handles(chain_cls):
    """Two host threads, one handle and one stream each, one device: concurrent create / upload / process /
    destroy; each thread's results equal what the same calls give single-threaded."""
    import torch
    g = load_golden("g4_q15_frames.npz")
    x = torch.from_numpy(g["x"]).cuda()
    with chain_cls(0) as ch:
        ch.set_filter_mode(0x00)
        ref_def = ch.process_q15(x).cpu().numpy()
        ch.load_coeffs_q7(g["c_gui"])
        ch.set_filter_mode(0xA1)
        ref_gui = ch.process_q15(x).cpu().numpy()
    errs = []

    def work(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for k in range(6):
                    with chain_cls(0) as c:
                        if (i + k) % 2:
                            c.load_coeffs_q7(g["c_gui"])
                            c.set_filter_mode(0xA1)
                            want = ref_gui
                        else:
                            c.set_filter_mode(0x00)
                            want = ref_def
                        got = c.process_q15(x)
                        st.synchronize()
                        if not np.array_equal(got.cpu().numpy(), want):
                            errs.append((i, k, "mismatch"))
        except Exception as e:              # noqa: BLE001
            errs.append((i, repr(e)))
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
