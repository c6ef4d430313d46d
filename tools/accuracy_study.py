#!/usr/bin/env python3
"""CPU study (no GPU): would a float64 / double-float predictor + scan improve the float chain's accuracy?

VERDICT r1 item 4 proposed it for sections the host flags (poles near the unit circle, outputs deep in a stop
band).  This script answers the question on the CPU model of the kernel's algebra before any kernel is written:

  A  the kernel's algorithm as it is (tests/test_host_logic.py::emulate_chunked_iir: float32 predictor taps,
     float32 Kogge-Stone scans in pole coordinates, float32 DF2T recursion per 32-sample chunk);
  B  the same recursion started from EXACT chunk start states (predictor and scan in float64, rounded to
     float32 once) -- the best any higher-precision predictor could possibly do;
  S  a plain sequential float32 sosfilt (oracle.sosfilt_f32_c).

All three are compared with the float64 oracle on the designs of tests/fuzz_parity.py (seeds 7 / 11 / 23 x 1500),
in the norm the parity gate uses (max-norm of the magnitude spectrum relative to its peak).  Output kept in
profiles/r2_accuracy_study.txt.  usage: accuracy_study.py [cases_per_seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as orc  # noqa: E402
from fpga_real_time_fft_analyzer_amd.chain import iir_plan_from_sos  # noqa: E402
import test_host_logic as thl  # noqa: E402

N = 16384
f = np.float32


def random_sos(rng):
    """The generator of tests/fuzz_parity.py (kept in step with it; that module imports torch + the GPU chain)."""
    from scipy import signal
    kind = rng.choice(["butter", "cheby1", "cheby2", "ellip", "bessel", "hand"])
    if kind == "hand":
        rows = []
        for _ in range(int(rng.integers(1, 7))):
            r, th = rng.uniform(0.1, 0.97), rng.uniform(0.05, 3.0)
            a1, a2 = -2 * r * np.cos(th), r * r
            b = rng.choice([0, 1, 2, 3])
            num = [[rng.normal(), rng.normal(), rng.normal()], [0.0, 1.0, 0.0], [0.0, 0.0, 0.0], [-0.3, 0.0, -0.3]][b]
            rows.append(num + [1.0, a1, a2])
        return np.array(rows), "hand"
    ft = rng.choice(["lowpass", "highpass", "bandpass", "bandstop"])
    order = int(rng.integers(1, 13))
    if ft in ("bandpass", "bandstop"):
        order = max(1, order // 2)
        lo = rng.uniform(0.02, 0.6)
        wn = [lo, min(0.95, lo + rng.uniform(0.05, 0.3))]
    else:
        wn = rng.uniform(0.01, 0.9)
    if kind == "butter":
        sos = signal.butter(order, wn, btype=ft, output="sos")
    elif kind == "cheby1":
        sos = signal.cheby1(order, rng.uniform(0.1, 3), wn, btype=ft, output="sos")
    elif kind == "cheby2":
        sos = signal.cheby2(order, rng.uniform(20, 80), wn, btype=ft, output="sos")
    elif kind == "ellip":
        sos = signal.ellip(order, rng.uniform(0.1, 3), rng.uniform(20, 80), wn, btype=ft, output="sos")
    else:
        sos = signal.bessel(order, wn, btype=ft, output="sos", norm="phase")
    return sos[:6], f"{kind}/{ft}/{order}"


def algo_a(sos, xw):
    plan = iir_plan_from_sos(sos)
    return thl.emulate_chunked_iir(plan, (xw * f(0.5)).astype(f)).astype(np.float64) * 2.0


def algo_b(sos, xw):
    sos = sos / sos[:, 3:4]
    v = xw.astype(f).reshape(512, 32).copy()
    for b0, b1, b2, _, a1, a2 in sos:
        A = np.array([[-a1, 1.0], [-a2, 0.0]])
        vv = np.array([b1 - a1 * b0, b2 - a2 * b0])
        taps = np.zeros((32, 2))
        for j in range(31, -1, -1):
            taps[j] = vv
            vv = A @ vv
        P = np.linalg.matrix_power(A, 32)
        z = v.astype(np.float64) @ taps
        s = np.zeros((512, 2))
        cur = np.zeros(2)
        for c in range(512):
            s[c] = cur
            cur = P @ cur + z[c]
        s1, s2 = s[:, 0].astype(f), s[:, 1].astype(f)
        b0f, b1f, b2f, a1f, a2f = f(b0), f(b1), f(b2), f(a1), f(a2)
        for j in range(32):
            x = v[:, j]
            y = (b0f * x + s1).astype(f)
            s1 = (b1f * x + s2 - a1f * y).astype(f)
            s2 = (b2f * x - a2f * y).astype(f)
            v[:, j] = y
    return v.reshape(-1).astype(np.float64)


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    hann64 = orc.hann_f64()
    hann = hann64.astype(f)
    tot = {"A": [0, 0, 0], "B": [0, 0, 0]}
    worst = []
    for seed in (7, 11, 23):
        rng = np.random.default_rng(seed)
        for case in range(ncases):
            sos, label = random_sos(rng)
            n = np.arange(N)
            x = (rng.uniform(0.1, 1.0) * np.sin(2 * np.pi * rng.uniform(0.001, 0.49, (3, 1)) * n)
                 + rng.uniform(0.0, 0.2) * rng.standard_normal((3, N))).astype(f)
            _, _, mag = orc.chain_fp(x, sos)
            den = np.abs(mag).max(axis=1)
            xw = (x * hann).astype(f)
            seq = np.stack([orc.sosfilt_f32_c(sos / sos[:, 3:4], r) for r in xw]).astype(np.float64)
            e = {}
            for name, y in (("S", seq), ("A", np.stack([algo_a(sos, r) for r in xw])), ("B", np.stack([algo_b(sos, r) for r in xw]))):
                e[name] = float((np.abs(np.abs(np.fft.fft(y, axis=1)) - mag).max(axis=1) / np.where(den > 0, den, 1.0)).max())
            for name in ("A", "B"):
                tot[name][0] += e[name] > 1e-5
                tot[name][1] += e[name] > max(1e-5, 1.5 * e["S"])
                tot[name][2] += e[name] > max(1e-5, 4.0 * e["S"])
            if e["A"] > max(1e-5, 1.5 * e["S"]):
                worst.append((e["A"] / e["S"], e["A"], e["B"], e["S"], seed, case, label))
    n = 3 * ncases
    print(f"{n} designs (seeds 7 / 11 / 23 x {ncases}), error = max-norm of the magnitude spectrum relative to its peak, vs float64")
    for name, what in (("A", "kernel algebra (float32 predictor + scan)"), ("B", "exact start states (float64 predictor + scan)")):
        t = tot[name]
        print(f"  {name}: {what:46s} above 1e-5: {t[0]:3d}   above max(1e-5, 1.5 x sequential f32): {t[1]:3d}   above max(1e-5, 4 x): {t[2]:3d}")
    print("  cases where A exceeds 1.5 x the sequential float32 result (ratio, A, B, sequential):")
    for r in sorted(worst, reverse=True)[:12]:
        print(f"     x{r[0]:5.1f}   A {r[1]:.2e}   B {r[2]:.2e}   S {r[3]:.2e}   seed {r[4]} case {r[5]} {r[6]}")


if __name__ == "__main__":
    main()
