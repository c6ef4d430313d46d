#!/usr/bin/env python3
"""Diagnostic (SA_STAMPS build): where do the waves of the Q15 filter kernel land, and how long does each run?

The integer cascade is serial in time: one wave per SIMD is all the parallelism a 4096-frame batch offers
(1024 waves, 1024 SIMDs).  A wave that shares its SIMD with another one runs at about 0.7 of the speed, and the
launch takes as long as its slowest wave.  This prints, per scenario, the histogram of waves per SIMD, the wave
run time by co-residency and the launch duration, for the first form (1024 one-wave workgroups, SA_Q7_OLD=1)
and the second form (256 four-wave workgroups).  usage: q15_placement.py"""
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

B = 4096
ch = SpectrumChain(0)
L = abi.lib()
L.sa_debug_set_q15_stamps.argtypes = [C.c_void_p]
stamps = torch.zeros((1024, 3), dtype=torch.int64, device="cuda")
assert L.sa_debug_set_q15_stamps(stamps.data_ptr()) == 0
gen = torch.Generator(device="cuda").manual_seed(2)
x = torch.randint(-2048, 2048, (B, 16384), generator=gen, device="cuda", dtype=torch.int32).to(torch.int16)
ot = torch.empty((B, 16384), dtype=torch.int16, device="cuda")
big = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device="cuda")
ch.set_filter_mode(0x00)


def report(label):
    s = stamps.cpu().numpy()
    t0, t1, hw = s[:, 0].astype(np.float64) * 10e-3, s[:, 1].astype(np.float64) * 10e-3, s[:, 2]     # us
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    xcc = (hw >> 32) & 0xF
    sid = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    dur = t1 - t0
    uniq, cnt = np.unique(sid, return_counts=True)
    per_wave = cnt[np.searchsorted(uniq, sid)]
    hist = {int(k): int((cnt == k).sum()) for k in np.unique(cnt)}
    print(f"{label}: launch span {t1.max() - t0.min():8.1f} us; SIMDs in use {len(uniq)} of 1024; SIMDs by number of waves {hist}")
    for k in sorted(set(per_wave.tolist())):
        d = dur[per_wave == k]
        print(f"      waves sharing their SIMD {k}-fold: {d.size:5d} waves, run time median {np.median(d):8.1f} us  max {d.max():8.1f} us")


if os.environ.get("SA_Q15_PLACEMENT_CHILD") is None:
    # the form is chosen by an environment switch read once per process: one child process per form
    import subprocess
    ch.close()
    del stamps, x, ot, big
    rc = 0
    for form, env in (("first form (1024 one-wave workgroups)", "1"), ("second form (256 four-wave workgroups)", None)):
        print(form, flush=True)
        e = dict(os.environ, SA_Q15_PLACEMENT_CHILD="1")
        if env:
            e["SA_Q7_OLD"] = env
        else:
            e.pop("SA_Q7_OLD", None)
        rc |= subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, check=False).returncode
    sys.exit(rc)

while True:
    for _ in range(3):
        ch.filter_q15(x, out=ot)
    torch.cuda.synchronize()
    ch.filter_q15(x, out=ot)
    torch.cuda.synchronize()
    report("   back to back          ")
    for _ in range(3):
        big.zero_()
        ch.filter_q15(x, out=ot)
        torch.cuda.synchronize()
    report("   after a 256 MiB memset")
    oq = torch.empty((B, 16384, 2), dtype=torch.int16, device="cuda")
    for _ in range(3):
        ch.process_q15(x, out=oq)            # filter + integer FFT, the real call sequence of config 4
        ch.filter_q15(x, out=ot)
        torch.cuda.synchronize()
    report("   after the integer FFT ")
    break
