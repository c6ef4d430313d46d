#!/usr/bin/env python3
"""Diagnostic: aggregate frames/s of the fused float chain when launches alternate over S streams
(one handle per stream), against the single-stream loop.  Inputs rotate over 4 buffer pairs per stream."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fpga_real_time_fft_analyzer_amd.chain import SpectrumChain  # noqa: E402

B, N, R = 4096, 16384, 4
sos = np.load(os.path.join(ROOT, "tests", "golden", "g2_config1.npz"))["sos"]
for S in (1, 2, 3):
    chains, streams, xs, outs = [], [], [], []
    for s in range(S):
        ch = SpectrumChain(0)
        ch.load_sos(sos)
        ch.set_filter_mode(0xA1)
        chains.append(ch)
        streams.append(torch.cuda.Stream())
        xs.append([torch.randn(B, N, device="cuda") for _ in range(R)])
        outs.append([torch.empty(B, N, device="cuda") for _ in range(R)])

    def run(steps):
        for i in range(steps):
            s = i % S
            with torch.cuda.stream(streams[s]):
                chains[s].process_f32(xs[s][(i // S) % R], out=outs[s][(i // S) % R])
    run(200)
    torch.cuda.synchronize()
    res = []
    for _ in range(5):
        t0 = time.perf_counter()
        run(60)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 60)
    dt = sorted(res)[2]
    print(f"{S} stream(s): {dt*1e6:7.1f} us per 4096-frame launch  {B/dt/1e6:6.2f} M frames/s", flush=True)
    for ch in chains:
        ch.close()
    del xs, outs
    torch.cuda.empty_cache()
