#!/usr/bin/env python3
"""Stage-by-stage check of the 512-thread kernel on the GPU (diagnostic builds: make ab NAME=dbg1 EXTRA=-DSA_W8_DEBUG=1,
NAME=dbg2 EXTRA=-DSA_W8_DEBUG=2): windowed samples, cascade output, then the spectrum with identity sections and with the
headline filter, each against the float64 oracle."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402  (checker only)

PKG = os.path.join(ROOT, "fpga_real_time_fft_analyzer_amd")
N = 16384
g = np.load(os.path.join(ROOT, "tests", "golden", "g2_config1.npz"))
rng = np.random.default_rng(5)
B = 3
n = np.arange(N)
x = (0.8 * np.sin(2 * np.pi * rng.uniform(0.01, 0.45, (B, 1)) * n) + 0.05 * rng.standard_normal((B, N))).astype(np.float32)
xd = torch.from_numpy(x).cuda()


def run(lib, sos, kind=0):
    L = C.CDLL(os.path.join(PKG, lib))
    h = C.c_void_p()
    assert L.sa_create(0, C.byref(h)) == 0
    L.sa_load_sos_f64.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
    L.sa_process_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.sa_set_filter_mode.argtypes = [C.c_void_p, C.c_uint8]
    s = np.ascontiguousarray(sos, np.float64)
    assert L.sa_load_sos_f64(h, s.ctypes.data_as(C.POINTER(C.c_double)), s.shape[0]) == 0
    L.sa_set_filter_mode(h, 0xA1)
    out = torch.zeros((B, N), dtype=torch.float32, device="cuda")
    rc = L.sa_process_f32(h, xd.data_ptr(), out.data_ptr(), B, kind, None)
    torch.cuda.synchronize()
    assert rc == 0, rc
    return out.cpu().numpy()


def rel(a, b):
    return float((np.abs(a - b).max(axis=1) / np.abs(b).max(axis=1)).max())


ident = np.array([[1.0, 0, 0, 1, 0, 0]] * 2)
hann = orc.hann_f64()
if os.path.exists(os.path.join(PKG, "libspecan_ab_dbg1.so")):
    got = run("libspecan_ab_dbg1.so", g["sos"])
    print("stage-in x window*gain:", rel(got / got.std(), (x * hann) / (x * hann).std()))
if os.path.exists(os.path.join(PKG, "libspecan_ab_dbg2.so")):
    for name, sos in (("identity", ident), ("butter12", g["sos"]), ("butter2 (1 section + identity)", g["sos"][:1])):
        got = run("libspecan_ab_dbg2.so", sos)
        y, _, _ = orc.chain_fp(x, sos)
        print(f"cascade output [{name}]:", rel(got, y), " first bad sample:", int(np.argmax(np.abs(got - y)[0] > 1e-4 * np.abs(y[0]).max())))
for name, sos in (("identity", ident), ("butter12", g["sos"])):
    got = run("libspecan_hip.so", sos)
    _, X, mag = orc.chain_fp(x, sos)
    print(f"spectrum [{name}]:", rel(got, mag))
    if rel(got, mag) > 1e-4:
        bad = np.nonzero(np.abs(got[0] - mag[0]) > 1e-4 * mag[0].max())[0]
        print("   bad bins:", len(bad), bad[:24], "...", bad[-8:])
