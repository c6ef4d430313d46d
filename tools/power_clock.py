#!/usr/bin/env python3
"""Is the headline launch power-limited?  Runs the float chain back to back for a few seconds per mode and samples
rocm-smi (socket power, shader clock) from a side thread meanwhile.  usage: power_clock.py [seconds_per_mode]
Output kept in profiles/r3_power_clock.txt."""
import os
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fpga_real_time_fft_analyzer_amd.chain import SpectrumChain  # noqa: E402

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
B, N = 4096, 16384


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--json"], capture_output=True,
                             text=True, timeout=20).stdout
        return out.strip()
    except Exception as e:                                   # noqa: BLE001
        return f"rocm-smi failed: {e}"


def sample_while(fn, secs):
    """Run fn() in a loop for `secs` seconds; a side thread takes rocm-smi snapshots."""
    stop, shots = threading.Event(), []

    def watcher():
        while not stop.is_set():
            shots.append(smi())
            time.sleep(0.3)
    th = threading.Thread(target=watcher)
    th.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < secs:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    return n, dt, shots


def main():
    print("idle:", smi())
    ch = SpectrumChain(0)
    ch.load_sos(np.load(os.path.join(ROOT, "tests", "golden", "g2_config1.npz"))["sos"])
    xs = [torch.randn(B, N, device="cuda") for _ in range(4)]
    outs = [torch.empty(B, N, device="cuda") for _ in range(4)]
    k = [0]

    def step():
        i = k[0] % 4
        k[0] += 1
        ch.process_f32(xs[i], out=outs[i])
    for mode, name in ((0xA1, "window + 6-section IIR + FFT + magnitude"), (0xB1, "IIR bypassed")):
        ch.set_filter_mode(mode)
        n, dt, shots = sample_while(step, SECS)
        print(f"\n== {name}: {n} launches of {B} frames in {dt:.2f} s = {dt / n * 1e6:.1f} us per launch (stream-ordered)")
        for s in shots[1:6]:
            print("  ", s)
    # the Q15 chain (config 4): integer cascade (one wave per SIMD, latency-bound) + fixed-point FFT
    xq = [torch.randint(-2048, 2048, (B, N), device="cuda", dtype=torch.int32).to(torch.int16) for _ in range(4)]
    oq = [torch.empty((B, N, 2), dtype=torch.int16, device="cuda") for _ in range(4)]
    ch.reserve(B)
    ch.set_filter_mode(0x00)

    def qstep():
        i = k[0] % 4
        k[0] += 1
        ch.process_q15(xq[i], out=oq[i])
    n, dt, shots = sample_while(qstep, SECS)
    print(f"\n== Q15 chain, default IIR: {n} launches of {B} frames in {dt:.2f} s = {dt / n * 1e6:.1f} us per launch")
    for s in shots[1:6]:
        print("  ", s)
    # a copy kernel for comparison: HBM traffic only
    def copy():
        i = k[0] % 4
        k[0] += 1
        outs[i].copy_(xs[i])
    n, dt, shots = sample_while(copy, SECS)
    print(f"\n== torch copy of the same buffers: {n} copies in {dt:.2f} s = {dt / n * 1e6:.1f} us per copy")
    for s in shots[1:6]:
        print("  ", s)


if __name__ == "__main__":
    main()
