#!/usr/bin/env python3
"""A/B timing of alternative builds of the library in one process, interleaved rounds.
usage: [SA_B=frames] ab_libs.py libA.so libB.so[@ovD] ...   (paths relative to the package dir; "@ov2" = overlap depth 2
on that handle, sa_set_overlap: timed with a flush before the final synchronisation; SA_B = batch size, default 4096:
the rotating buffers are slices of one 4096-frame pool, so small batches still stream from HBM)"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "fpga_real_time_fft_analyzer_amd")
sos = np.ascontiguousarray(np.load(os.path.join(ROOT, "tests", "golden", "g2_config1.npz"))["sos"], np.float64)
B = int(os.environ.get("SA_B", "4096"))
POOL = max(B, 4096)
gen = torch.Generator(device="cuda").manual_seed(1)
n = torch.arange(16384, device="cuda", dtype=torch.float32)
fb = torch.rand(POOL, 1, generator=gen, device="cuda") * 0.44 + 0.01
pool = (0.8 * torch.sin(2 * np.pi * fb * n) + 0.05 * torch.randn(POOL, 16384, generator=gen, device="cuda")).contiguous()
x = pool[:B]
out = torch.empty_like(x)
# ROT buffer pairs used round-robin in the timed loops: one pair (256 MiB in) would sit in the 256 MB Infinity
# Cache from launch to launch and the comparison would be about the cache, not HBM
ROT = int(os.environ.get("SA_ROT", "4"))
if B < 4096:                                   # 4 x 4096 frames of input and output, cut into B-frame slices
    pools = [pool] + [pool.clone() for _ in range(3)]
    opools = [torch.empty_like(pool) for _ in range(4)]
    xs = [p[j:j + B] for p in pools for j in range(0, POOL - B + 1, B)]
    outs = [p[j:j + B] for p in opools for j in range(0, POOL - B + 1, B)]
    ROT = len(xs)
else:
    xs = [x] + [x.clone() for _ in range(ROT - 1)]
    outs = [out] + [torch.empty_like(x) for _ in range(ROT - 1)]
libs = []
for name in sys.argv[1:]:
    # "lib.so:VAR=value" sets an environment variable while that library's handle is created (plan-time switches)
    envset = None
    depth = 1
    if "@ov" in name:
        name, d = name.split("@ov", 1)
        depth = int(d)
    if ":" in name:
        name, envset = name.split(":", 1)
        k, v = envset.split("=", 1)
        os.environ[k] = v
    L = C.CDLL(os.path.join(PKG, name))
    h = C.c_void_p()
    assert L.sa_create(0, C.byref(h)) == 0
    L.sa_load_sos_f64.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
    L.sa_process_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.sa_set_filter_mode.argtypes = [C.c_void_p, C.c_uint8]
    assert L.sa_load_sos_f64(h, sos.ctypes.data_as(C.POINTER(C.c_double)), 6) == 0
    if envset:
        del os.environ[envset.split("=", 1)[0]]
        name = name + ":" + envset
    if depth > 1:
        L.sa_set_overlap.argtypes = [C.c_void_p, C.c_int]
        L.sa_flush.argtypes = [C.c_void_p, C.c_void_p]
        assert L.sa_set_overlap(h, depth) == 0
        name = f"{name}@ov{depth}"
    if any(nm == name for nm, _, _ in libs):
        name = f"{name}#{len(libs)}"            # the same build on a further handle
    libs.append((name, L, h))
st = torch.cuda.current_stream().cuda_stream
ROUNDS, REPS = 12, (40 if B >= 4096 else 200)
for mode in (0xA1, 0xB1):
    res, ref = {}, None
    # sustained pre-warm so that every build is measured at the settled clock
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        for name, L, h in libs:
            L.sa_set_filter_mode(h, mode)
            L.sa_process_f32(h, x.data_ptr(), out.data_ptr(), B, 0, st)
        torch.cuda.synchronize()
    for rnd in range(ROUNDS):
        order = libs[rnd % len(libs):] + libs[:rnd % len(libs)]          # rotate: no build always runs first
        for name, L, h in order:
            L.sa_set_filter_mode(h, mode)
            for _ in range(3):
                L.sa_process_f32(h, x.data_ptr(), out.data_ptr(), B, 0, st)
            torch.cuda.synchronize()
            if rnd == 0:
                if ref is None:
                    ref = out.clone()
                else:
                    d = (out - ref).abs().max().item() / ref.abs().max().item()
                    print(f"   {name} vs {order[0][0]} mode 0x{mode:02X}: max rel diff {d:.2e}")
            t0 = time.perf_counter()
            for i in range(REPS):
                L.sa_process_f32(h, xs[i % ROT].data_ptr(), outs[i % ROT].data_ptr(), B, 0, st)
            if "@ov" in name:
                L.sa_flush(h, st)
            torch.cuda.synchronize()
            res.setdefault(name, []).append((time.perf_counter() - t0) / REPS)
    for name, _, _ in libs:
        v = sorted(res[name])
        print(f"mode 0x{mode:02X} {name:32s} median {v[len(v)//2]*1e6:8.1f} us  min {v[0]*1e6:8.1f} us  max {v[-1]*1e6:8.1f} us  -> {B/v[len(v)//2]/1e6:6.2f} M frames/s")
