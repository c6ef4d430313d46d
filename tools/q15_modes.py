#!/usr/bin/env python3
"""Time the integer cascade kernels mode by mode at B frames, stream-ordered: the cascade alone (sa_filter_q15) and the
whole integer chain (sa_process_q15), per call from the launches' own start / stop events (sa_set_profiling), after a warm-up.  usage: q15_modes.py [B] [LIB]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fpga_real_time_fft_analyzer_amd import abi  # noqa: E402
from fpga_real_time_fft_analyzer_amd.chain import SpectrumChain  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
if len(sys.argv) > 2:                                  # an A/B build of the library (make ab NAME=...)
    abi.LIB_PATH = os.path.join(ROOT, "fpga_real_time_fft_analyzer_amd", sys.argv[2])
    print("library:", sys.argv[2])
N = 16384
g4 = np.load(os.path.join(ROOT, "tests", "golden", "g4_q15_frames.npz"))
gen = torch.Generator(device="cuda").manual_seed(2)
xs = [torch.randint(-2048, 2048, (B, N), generator=gen, device="cuda", dtype=torch.int32).to(torch.int16) for _ in range(3)]
ot = torch.empty((B, N), dtype=torch.int16, device="cuda")
oq = torch.empty((B, N, 2), dtype=torch.int16, device="cuda")
ch = SpectrumChain(0)
ch.reserve(B)
ch.load_coeffs_q7(g4["c_gui"])
ch.load_sos_q14(g4["sos_q14"])
for _ in range(40):                                    # leave the idle clocks
    ch.process_q15(xs[0], out=oq)
torch.cuda.synchronize()
for name, cmd in (("0x00 default (7-instruction step)", 0x00), ("0xA1 GUI upload (9-instruction step)", 0xA1),
                  ("0xA2 wide Q2.14, 6 sections", 0xA2), ("0xB1 bypass", 0xB1)):
    ch.set_filter_mode(cmd)
    for label, fn, o in (("cascade", ch.filter_q15, ot), ("chain", ch.process_q15, oq)):
        for k in range(3):
            fn(xs[k % 3], out=o)
        ch.set_profiling(12)                              # device time of each call from the launch's own events
        for k in range(12):
            fn(xs[k % 3], out=o)
        ts = sorted(v * 1e3 for v in ch.profile_read(12))
        ch.set_profiling(0)
        if label == "chain":
            import hashlib
            digest = hashlib.sha256(o[:64].cpu().numpy().tobytes()).hexdigest()[:12]
        else:
            digest = ""
        print(f"{name:40s} {label:8s} {digest:12s} median {ts[len(ts)//2]:8.1f} us  min {ts[0]:8.1f}  -> {B / ts[len(ts)//2]:6.2f} M frames/s")
