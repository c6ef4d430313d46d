#!/usr/bin/env python3
"""Diagnostic: where does a workgroup of the fused float kernel spend its cycles?

Loads the SA_STAMPS build (make -C fpga_real_time_fft_analyzer_amd/csrc stamps), runs one batch and
prints per-phase shader-clock deltas (median over workgroups).  The stamped build serialises what
the product kernel overlaps: read the SHARES, never its run time (cdna_hip_programming.md section 7).
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fpga_real_time_fft_analyzer_amd import abi  # noqa: E402

abi.LIB_PATH = os.path.join(ROOT, "fpga_real_time_fft_analyzer_amd", "libspecan_hip_stamps.so")
from fpga_real_time_fft_analyzer_amd.chain import SpectrumChain  # noqa: E402

NAMES = {"0->1": "stage-in (DMA + chunk read x window)", "1->2": "IIR cascade (6 sections)",
         "2->3": "exchange to pass A", "0->3": "stage-in (DMA + read x window)", "3->4": "FFT32 + twiddle A",
         "4->5": "exchange A->B (2 rounds)", "5->6": "FFT16x2 + twiddle B", "6->7": "exchange B->C (in-row)",
         "7->8": "FFT16x2", "8->9": "natural image round 0", "9->10": "split+store round 0, image round 1",
         "10->11": "split+store round 1", "11->12": "store drain"}


# build with `make stamps EXTRA=-DSA_STAMP_IIR`: stamps 3..8 sit inside cascade section 2 (usage: ... 4096 0xA1 iir)
NAMES_IIR = {"3->4": "sec 2: thread total + in-row scan + row total out", "4->5": "sec 2: LDS barrier",
             "5->6": "sec 2: row start, start states, DF2T states", "6->7": "sec 2: recursion (32 steps)",
             "7->8": "sec 2: predictor of section 3 (32 taps)"}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    mode = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0xA1
    ch = SpectrumChain(0)
    L = abi.lib()
    L.sa_debug_set_stamps.argtypes = [C.c_void_p]
    stamps = torch.zeros((B, 16), dtype=torch.int64, device="cuda")
    assert L.sa_debug_set_stamps(stamps.data_ptr()) == 0
    sos = np.load(os.path.join(ROOT, "tests", "golden", "g2_config1.npz"))["sos"]
    ch.load_sos(sos)
    ch.set_filter_mode(mode)
    x = torch.randn(B, 16384, device="cuda")
    for _ in range(3):
        ch.process_f32(x)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(np.float64)
    iir = mode != 0xB1
    idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12] if iir else [0, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]
    if len(sys.argv) > 3 and sys.argv[3] == "iir":
        idx = [3, 4, 5, 6, 7, 8]
        NAMES.update(NAMES_IIR)
    d = np.diff(s[:, idx], axis=1)
    tot = s[:, 12] - s[:, 0]
    if idx[0] == 3:
        tot = s[:, 8] - s[:, 3]
    print(f"B={B} mode=0x{mode:02X}: workgroup lifetime median {np.median(tot):.0f} clk (s_memtime ticks = 100 MHz? see note)")
    labels = [f"{idx[i]}->{idx[i+1]}" for i in range(len(idx) - 1)]
    for i, lab in enumerate(labels):
        print(f"  {lab:7s} {NAMES.get(lab, ''):42s} median {np.median(d[:, i]):9.0f}  p10 {np.percentile(d[:, i], 10):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f}"
              f"  share {np.median(d[:, i]) / np.median(tot) * 100:5.1f}%")
    # (stamps 0..12 are s_memtime readings of each workgroup's own XCD: spans across workgroups are meaningful only in the
    #  100 MHz s_memrealtime stamps 13 / 14, which tools/wg_timeline.py evaluates)


if __name__ == "__main__":
    main()
