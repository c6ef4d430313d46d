#!/usr/bin/env python3
"""Soak run on the GPU box (test infrastructure: uses the oracle): random batch sizes, filter modes, coefficient
uploads, window modes and input scales through the Q15 path, bit-exact against the integer model every time;
float chain on random batch sizes (float32 frames or int16 samples in) within tolerance.  Round 4: mode 0xA2 (random wide cascades) is in the mix.  Round 3: the launch mode changes at random (ordered, two or three
launches in flight: sa_set_overlap), in overlap mode up to three calls are issued back to back before the flush, with control-plane
calls between them.  usage: soak.py SECONDS [SEED] [big]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fpga_real_time_fft_analyzer_amd.chain import SpectrumChain  # noqa: E402
from oracle import oracle as orc  # noqa: E402

N = 16384
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
# "big" as third argument: batches of hundreds to thousands of frames (workspace growth in flight, ragged last workgroups
# of both integer cascades, many generations of workgroups per launch)
BATCHES = ([300, 511, 513, 1000, 1023, 1025, 2047, 2049, 3000, 4095, 4096, 4097, 5000] if len(sys.argv) > 3 and sys.argv[3] == "big"
           else [1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 33, 63, 64, 65, 100, 129, 255, 257])
ch = SpectrumChain(0)
sos = np.load(os.path.join(ROOT, "tests", "golden", "g2_config1.npz"))["sos"]
t0 = time.time()
n_q15 = n_f32 = n_i16 = 0
worst = 0.0
last = t0
n_mode = [0, 0, 0, 0]
while time.time() - t0 < budget:
    if rng.integers(0, 8) == 0:
        ch.set_overlap(int(rng.choice([1, 2, 3])))
    depth = ch.overlap
    n_mode[depth] += 1
    burst = 1 if depth == 1 else int(rng.integers(1, 4))
    pending = []
    for _ in range(burst):
        B = int(rng.choice(BATCHES))
        scale = int(rng.choice([16, 2048, 32768]))
        x = rng.integers(-scale, scale, size=(B, N)).astype(np.int16)
        cmd = int(rng.choice([0x00, 0xA1, 0xB1, 0xA2]))
        wm = int(rng.integers(0, 2))
        sos14 = None
        if cmd == 0xA2:                                     # round 4: random wide cascades, 1..6 sections, any int16 tap
            sos14 = rng.integers(-32768, 32768, (int(rng.integers(1, 7)), 6)).astype(np.int16)
            if rng.integers(0, 2):
                sos14 = (sos14 // 4).astype(np.int16)
            ch.load_sos_q14(sos14)
        c12 = rng.integers(-128, 128, size=12).astype(np.int8)
        if rng.integers(0, 3) == 0:
            c12[1] = c12[7] = 0                             # the short integer step
        ch.set_window_mode_q15(wm)
        if cmd == 0xA1:
            ch.load_coeffs_q7(c12)
        ch.set_filter_mode(cmd)
        ref = orc.chain_q15(x, None, wm, cmd, c12 if cmd == 0xA1 else None, sos14)
        xd = torch.from_numpy(x).cuda()                     # kept until the flush: in overlap mode the tensors of a call
        pending.append((ch.process_q15(xd), ref, B, cmd, wm, scale, xd))     # belong to the library until it is joined
    if depth > 1:
        ch.flush()
    for out, ref, B, cmd, wm, scale, _ in pending:
        got = out.cpu().numpy()
        if not np.array_equal(got, ref):
            bad = np.argwhere(got != ref)
            print(f"MISMATCH q15 depth={depth} B={B} cmd=0x{cmd:02X} wm={wm} scale={scale}: {len(bad)} values differ, first {bad[0]}")
            sys.exit(1)
        n_q15 += 1
    if n_q15 % 4 < burst:
        Bf = int(rng.choice([130, 257, 600, 1025] if len(BATCHES) == 13 else [1, 3, 8, 17, 64, 130]))
        xf = (rng.uniform(0.1, 1.0) * np.sin(2 * np.pi * rng.uniform(0.001, 0.2, (Bf, 1)) * np.arange(N))
              + 0.05 * rng.standard_normal((Bf, N))).astype(np.float32)
        ch.load_sos(sos)
        ch.set_filter_mode(0xA1 if rng.integers(0, 2) else 0xB1)       # (bypassed batches <= 512 take the one-round stage-in)
        _, _, mag = orc.chain_fp(xf, sos if ch.filter_mode == 0xA1 else None)
        if rng.integers(0, 2):                              # the same chain from int16 samples (sa_process_f32_i16)
            xi = np.clip(np.round(xf * 2048.0), -32768, 32767).astype(np.int16)
            xf = (xi.astype(np.float32) * np.float32(1.0 / 2048.0)).astype(np.float32)
            _, _, mag = orc.chain_fp(xf, sos if ch.filter_mode == 0xA1 else None)
            xfd = torch.from_numpy(xi).cuda()
            n_i16 += 1
        else:
            xfd = torch.from_numpy(xf).cuda()
        outf = ch.process_f32(xfd)
        if depth > 1:
            ch.flush()
        gotf = outf.cpu().numpy()
        err = float((np.abs(gotf - mag).max(axis=1) / np.abs(mag).max(axis=1)).max())
        worst = max(worst, err)
        if not err < 1e-5:
            print(f"MISMATCH f32 B={Bf} mode=0x{ch.filter_mode:02X}: rel err {err:.2e}")
            sys.exit(1)
        n_f32 += 1
    if time.time() - last > 30:
        last = time.time()
        print(f"  {n_q15} Q15 cases bit-exact, {n_f32} float cases (worst {worst:.2e}) after {last - t0:.0f} s", flush=True)
print(f"soak ok: {n_q15} Q15 cases bit-exact, {n_f32} float cases ({n_i16} of them from int16 samples) within 1e-5 (worst {worst:.2e}) in {time.time() - t0:.0f} s; "
      f"bursts in ordered / depth-2 / depth-3 mode: {n_mode[1]} / {n_mode[2]} / {n_mode[3]}")
