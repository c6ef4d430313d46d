#!/usr/bin/env python3
"""A/B timing of sa_filter_q15 / sa_process_q15 for alternative builds (diagnostic)."""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fpga_real_time_fft_analyzer_amd")
B = 4096
gen = torch.Generator(device="cuda").manual_seed(2)
x = torch.randint(-2048, 2048, (B, 16384), generator=gen, device="cuda", dtype=torch.int32).to(torch.int16)
ot = torch.empty((B, 16384), dtype=torch.int16, device="cuda")
oq = torch.empty((B, 16384, 2), dtype=torch.int16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for name in sys.argv[1:]:
    # "lib.so:VAR=value": the variable is set while this library runs its first calls (launch-time switches are read once)
    envkey = None
    if ":" in name:
        name, kv = name.split(":", 1)
        envkey, val = kv.split("=", 1)
        os.environ[envkey] = val
    L = C.CDLL(os.path.join(PKG, name))
    h = C.c_void_p()
    assert L.sa_create(0, C.byref(h)) == 0
    for fn in (L.sa_filter_q15, L.sa_process_q15):
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.sa_set_filter_mode.argtypes = [C.c_void_p, C.c_uint8]
    L.sa_reserve.argtypes = [C.c_void_p, C.c_int]
    L.sa_reserve(h, B)
    for mode in (0x00, 0xB1):
        L.sa_set_filter_mode(h, mode)
        for label, fn, o in (("filter", L.sa_filter_q15, ot), ("process", L.sa_process_q15, oq)):
            for _ in range(2):
                fn(h, x.data_ptr(), o.data_ptr(), B, st)
            torch.cuda.synchronize()
            dt = 1e9
            for _ in range(3):                      # best of three rounds of eight calls
                t0 = time.perf_counter()
                for _ in range(8):
                    fn(h, x.data_ptr(), o.data_ptr(), B, st)
                torch.cuda.synchronize()
                dt = min(dt, (time.perf_counter() - t0) / 8)
            print(f"{name:28s} mode 0x{mode:02X} {label:8s} {dt*1e6:9.1f} us  {B/dt/1e6:6.2f} M frames/s")
    if envkey:
        del os.environ[envkey]

# diagnostic: filter kernel with the caches thrashed / after a heavy kernel in between
big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda")
L.sa_set_filter_mode(h, 0x00)
for label, between in (("filter after 1 GiB memset", lambda: big.zero_()),
                       ("filter after sin() on 64 MiB", lambda: torch.sin_(big[:16 * 1024 * 1024]))):
    evs = []
    for _ in range(6):
        between()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.sa_filter_q15(h, x.data_ptr(), ot.data_ptr(), B, st)
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    print(f"{label:32s} median {ts[len(ts)//2]:9.1f} us  min {ts[0]:9.1f}")
