#!/usr/bin/env python3
"""Diagnostic: frames/s of the fused float kernel vs batch size (convoy / fill-drain effects)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fpga_real_time_fft_analyzer_amd.chain import SpectrumChain  # noqa: E402

ch = SpectrumChain(0)
sos = np.load(os.path.join(ROOT, "tests", "golden", "g2_config1.npz"))["sos"]
ch.load_sos(sos)
for mode in (0xA1, 0xB1):
    ch.set_filter_mode(mode)
    for B in (256, 512, 1024, 2048, 4096, 8192, 16384):
        x = torch.randn(B, 16384, device="cuda")
        out = torch.empty_like(x)
        for _ in range(3):
            ch.process_f32(x, out=out)
        torch.cuda.synchronize()
        n = max(5, 40960 // B)
        t0 = time.perf_counter()
        for _ in range(n):
            ch.process_f32(x, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"mode 0x{mode:02X} B={B:6d}: {dt*1e6:9.1f} us/launch  {B/dt/1e6:7.2f} M frames/s  {B*131072/dt/1e9:7.1f} GB/s", flush=True)
