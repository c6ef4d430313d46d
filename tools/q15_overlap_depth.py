#!/usr/bin/env python3
"""Config 4 (Q15 chain, default IIR, B = 4096) against the number of launches in flight (sa_set_overlap 1..4).  The integer
cascade runs one wave per SIMD and is latency-bound, so two cascades side by side cost little more than one.
usage: q15_overlap_depth.py [batch] [library file in the package directory]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fpga_real_time_fft_analyzer_amd import abi  # noqa: E402
if len(sys.argv) > 2:                                    # an A/B build (make ab NAME=...) instead of the product library
    abi.LIB_PATH = os.path.join(os.path.dirname(abi.LIB_PATH), sys.argv[2])
from fpga_real_time_fft_analyzer_amd.chain import SpectrumChain  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N, R = 16384, 8
ch = SpectrumChain(0)
xs = [torch.randint(-2048, 2048, (B, N), device="cuda", dtype=torch.int32).to(torch.int16) for _ in range(R)]
outs = [torch.empty((B, N, 2), dtype=torch.int16, device="cuda") for _ in range(R)]
ref = None
for mode, name in ((0x00, "default IIR"), (0xB1, "IIR bypassed")):
    ch.set_filter_mode(mode)
    for d in (1, 2, 3, 4):
        ch.set_overlap(d)
        ch.reserve(B)
        k = 0

        def step():
            global k
            ch.process_q15(xs[k % R], out=outs[k % R])
            k += 1
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
        chk = outs[0].clone()
        if d == 1:
            ref = chk
        same = bool(torch.equal(chk, ref))
        print(f"Q15 {name:13s} B={B} depth {d}: {ts[2] * 1e6:7.1f} us per batch = {B / ts[2] / 1e6:5.2f} M frames/s   (output equals depth 1: {same})")
    ch.set_overlap(1)
