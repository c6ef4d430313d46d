#!/usr/bin/env python3
"""Diagnostic: workgroup residency over the launch (SA_STAMPS build).

Every workgroup records s_memrealtime (100 MHz, one counter for the whole chip) at its first and last
instruction and its placement (HW_ID, XCC_ID).  From that: how many workgroups a CU holds over time, how
long a freed slot stays empty, how the launch ramps up and drains.  usage: wg_timeline.py [B] [mode]"""
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


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    mode = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0xA1
    ch = SpectrumChain(0)
    L = abi.lib()
    L.sa_debug_set_stamps.argtypes = [C.c_void_p]
    stamps = torch.zeros((B, 16), dtype=torch.int64, device="cuda")
    assert L.sa_debug_set_stamps(stamps.data_ptr()) == 0
    ch.load_sos(np.load(os.path.join(ROOT, "tests", "golden", "g2_config1.npz"))["sos"])
    ch.set_filter_mode(mode)
    xs = [torch.randn(B, 16384, device="cuda") for _ in range(4)]
    outs = [torch.empty(B, 16384, device="cuda") for _ in range(4)]
    for i in range(12):                                   # the last launch is the one whose stamps survive
        ch.process_f32(xs[i % 4], out=outs[i % 4])
    torch.cuda.synchronize()
    s = stamps.cpu().numpy()
    t0, t1 = s[:, 13].astype(np.float64) * 10.0, s[:, 14].astype(np.float64) * 10.0     # ns
    hw = s[:, 15]
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 0x7
    xcc = (hw >> 32) & 0xF
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    base = t0.min()
    t0 -= base
    t1 -= base
    span = t1.max()
    life = t1 - t0
    ncu = len(np.unique(cuid))
    print(f"B={B} mode=0x{mode:02X}: launch span {span / 1e3:.1f} us, {ncu} distinct CUs, workgroup lifetime mean "
          f"{life.mean() / 1e3:.2f} us median {np.median(life) / 1e3:.2f} us p90 {np.percentile(life, 90) / 1e3:.2f} us")
    print(f"  mean resident workgroups per CU over the span: {life.sum() / span / ncu:.2f}")
    # shader clock the launch ran at: s_memtime ticks (stamps 0 and 12, one per shader cycle) over the 100 MHz counter
    cyc = (s[:, 12] - s[:, 0]).astype(np.float64)
    ok = (life > 0) & (cyc > 0)
    ghz = cyc[ok] / life[ok]
    print(f"  shader clock over a workgroup's life (s_memtime / s_memrealtime): median {np.median(ghz):.3f} GHz, "
          f"p10 {np.percentile(ghz, 10):.3f}, p90 {np.percentile(ghz, 90):.3f}")
    # residency histogram over time
    nb = 20
    edges = np.linspace(0, span, nb + 1)
    res = [(np.minimum(t1, edges[i + 1]) - np.maximum(t0, edges[i])).clip(min=0).sum() / (edges[i + 1] - edges[i]) / ncu
           for i in range(nb)]
    print("  resident WGs per CU in 20 time slices: " + " ".join(f"{r:.1f}" for r in res))
    # slot turnover: per CU, gap between the k-th end and the (k+4)-th start (4 slots per CU)
    gaps = []
    per_cu_counts = []
    for c in np.unique(cuid):
        m = cuid == c
        st, en = np.sort(t0[m]), np.sort(t1[m])
        per_cu_counts.append(m.sum())
        if len(st) > 4:
            gaps.extend((st[4:] - en[:len(st) - 4]).tolist())
    gaps = np.array(gaps)
    print(f"  workgroups per CU: min {min(per_cu_counts)} max {max(per_cu_counts)}")
    if gaps.size:
        print(f"  slot turnover (start of the next WG - end of the WG that freed the slot): median {np.median(gaps):.0f} ns, "
              f"p10 {np.percentile(gaps, 10):.0f} ns, p90 {np.percentile(gaps, 90):.0f} ns")
    first_round = np.sort(t0)[:min(B, 4 * ncu)]
    print(f"  first {len(first_round)} workgroups start within {first_round.max() / 1e3:.2f} us; last workgroup starts at "
          f"{t0.max() / 1e3:.1f} us, first one ends at {t1.min() / 1e3:.1f} us")
    # lifetime by start time quartile
    order = np.argsort(t0)
    for q in range(4):
        sel = order[q * B // 4:(q + 1) * B // 4]
        print(f"  start-order quartile {q}: lifetime mean {life[sel].mean() / 1e3:.2f} us")


if __name__ == "__main__":
    main()
