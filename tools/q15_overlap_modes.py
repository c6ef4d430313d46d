#!/usr/bin/env python3
"""The integer chain per filter mode (0xA2 wide, 0x00 default, 0xA1 GUI upload) against the number of launches in flight.
usage: q15_overlap_modes.py [library file in the package directory]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fpga_real_time_fft_analyzer_amd import abi  # noqa: E402
NAME = sys.argv[1] if len(sys.argv) > 1 else "product"
if len(sys.argv) > 1:
    abi.LIB_PATH = os.path.join(os.path.dirname(abi.LIB_PATH), sys.argv[1])
from fpga_real_time_fft_analyzer_amd.chain import SpectrumChain  # noqa: E402

B, N, R = 4096, 16384, 6
ch = SpectrumChain(0)
g4 = np.load(os.path.join(ROOT, "tests", "golden", "g4_q15_frames.npz"))
ch.load_sos_q14(g4["sos_q14"])
ch.load_coeffs_q7(g4["c_gui"])
xs = [torch.randint(-2048, 2048, (B, N), device="cuda", dtype=torch.int32).to(torch.int16) for _ in range(R)]
outs = [torch.empty((B, N, 2), dtype=torch.int16, device="cuda") for _ in range(R)]
for mode in (0xA2, 0x00, 0xA1):
    ch.set_filter_mode(mode)
    for d in (1, 2, 3):
        ch.set_overlap(d)
        ch.reserve(B)
        k = [0]

        def step():
            ch.process_q15(xs[k[0] % R], out=outs[k[0] % R])
            k[0] += 1
        for _ in range(2 * R):
            step()
        ch.flush()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(24):
                step()
            ch.flush()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 24)
        ts.sort()
        print(f"{NAME:26s} mode 0x{mode:02X} depth {d}: {ts[2] * 1e6:7.1f} us per batch = {B / ts[2] / 1e6:5.2f} M frames/s")
    ch.set_overlap(1)
