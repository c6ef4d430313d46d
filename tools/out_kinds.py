#!/usr/bin/env python3
"""Diagnostic: time of the fused float chain for each output kind at B = 4096 (4 rotating buffer pairs)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fpga_real_time_fft_analyzer_amd.chain import SpectrumChain  # noqa: E402

B, N, R = 4096, 16384, 4
ch = SpectrumChain(0)
ch.load_sos(np.load(os.path.join(ROOT, "tests", "golden", "g2_config1.npz"))["sos"])
xs = [torch.randn(B, N, device="cuda") for _ in range(R)]
for mode in (0xA1, 0xB1):
    ch.set_filter_mode(mode)
    for kind in ("mag_full", "mag_half", "spec_half", "time"):
        outs = [ch.process_f32(xs[r], out_kind=kind) for r in range(R)]
        for i in range(50):
            ch.process_f32(xs[i % R], out=outs[i % R], out_kind=kind)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(60):
            ch.process_f32(xs[i % R], out=outs[i % R], out_kind=kind)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 60
        nbytes = B * N * 4 + outs[0].numel() * outs[0].element_size()
        print(f"mode 0x{mode:02X} {kind:10s} {dt*1e6:7.1f} us  {B/dt/1e6:6.2f} M frames/s  {nbytes/dt/1e12:5.2f} TB/s", flush=True)
        del outs
