#!/usr/bin/env python3
"""Index model of chain_f32_w8.hip (512 threads per frame): every exchange, every twiddle and the output stage
written with the kernel's own index formulas on numpy arrays, checked against numpy.fft.rfft.  CPU only; it exists
so that an indexing mistake costs a second here instead of a GPU call.  (tests/test_host_logic.py runs it.)

Frame x[16384] real -> z[m] = x[2m] + i x[2m+1] (8192 complex) -> E = FFT4096(z[2j]), O = FFT4096(z[2j+1]), each
16 x 16 x 16 -> output stage: Z[k] = E[k] + W_8192^k O[k] merged into the split step of the packed real FFT.
"""
import numpy as np

N, M = 16384, 8192


def brev4(v):
    return ((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3)


def fft16_dit(a):
    """a[brev4(n)] = x[n] in, natural order out (what safft::fft_dit<16> does)."""
    x = np.empty(16, complex)
    for n in range(16):
        x[n] = a[brev4(n)]
    return np.fft.fft(x)


def model(x):
    T = 512
    z = x[0::2] + 1j * x[1::2]
    img = np.zeros(4352, complex)
    # ---- exchange IIR layout -> pass A (two rounds by writer half)
    a_in = np.zeros((T, 16), complex)                 # a_in[t][brev4(m1)]
    for h in range(2):
        img[:] = np.nan
        for t in range(256 * h, 256 * h + 256):
            tp = t & 255
            for i in range(16):
                img[17 * tp + i] = z[16 * t + i]
        for t in range(T):
            for m1p in range(8):
                a_in[t][brev4(8 * h + m1p)] = img[544 * m1p + 17 * (t >> 4) + (t & 15)]
    for t in range(T):
        for m1 in range(16):
            assert a_in[t][brev4(m1)] == z[512 * m1 + t]
    # ---- pass A: FFT16 over m1, twiddle W_4096^(k1 u), u = t >> 1
    A = np.zeros((T, 16), complex)
    for t in range(T):
        u = t >> 1
        A[t] = fft16_dit(a_in[t]) * np.exp(-2j * np.pi * np.arange(16) * u / 4096)
    # ---- exchange A -> B, two rounds; every thread hands over eight values and takes eight in per round (registers
    #      and LDS hold the data exactly once): round q pairs writer half (t >> 8) and reader half (tb >> 8) that
    #      differ by q.  Registers modelled in place: a value is gone once written, bq fills up over the two rounds.
    b_in = np.full((T, 16), np.nan + 0j)
    Areg = A.copy()
    for q in range(2):
        img[:] = np.nan
        for t in range(T):
            hi = t >> 8
            for r in range(8):
                k1 = 8 * (hi ^ q) + r
                img[512 * r + t] = Areg[t][k1]
                Areg[t][k1] = np.nan
        for tb in range(T):
            k1, p, b, hi = tb >> 5, (tb >> 4) & 1, tb & 15, tb >> 8
            for aa in range(8):
                a = 8 * (hi ^ q) + aa
                b_in[tb][brev4(a)] = img[512 * (k1 & 7) + 32 * a + 2 * b + p]
    assert np.isnan(Areg).all() and not np.isnan(b_in).any()
    # ---- pass B: FFT16 over a, twiddle W_256^(b k2)
    Bv = np.zeros((T, 16), complex)
    for tb in range(T):
        b = tb & 15
        Bv[tb] = fft16_dit(b_in[tb]) * np.exp(-2j * np.pi * np.arange(16) * b / 256)
    # ---- B -> C: 16x16 transpose inside each 16-lane group (plane-wise through 272 floats per group)
    c_in = np.zeros((T, 16), complex)
    fimg = np.zeros(32 * 272)
    for plane in range(2):
        fimg[:] = np.nan
        for tb in range(T):
            g16, b = tb >> 4, tb & 15
            for k2 in range(16):
                v = Bv[tb][k2]
                fimg[g16 * 272 + k2 * 17 + b] = v.real if plane == 0 else v.imag
        for tc in range(T):
            g16, k2 = tc >> 4, tc & 15
            for bb in range(16):
                v = fimg[g16 * 272 + k2 * 17 + bb]
                if plane == 0:
                    c_in[tc][brev4(bb)] = v
                else:
                    c_in[tc][brev4(bb)] += 1j * v
    # ---- pass C: FFT16 over b -> k3; thread tc = 32 k1 + 16 p + k2 holds F_p[k1 + 16 k2 + 256 k3]
    C = np.zeros((T, 16), complex)
    for tc in range(T):
        C[tc] = fft16_dit(c_in[tc])
    E = np.fft.fft(z[0::2])
    O = np.fft.fft(z[1::2])
    for tc in range(T):
        k1, p, k2 = tc >> 5, (tc >> 4) & 1, tc & 15
        ref = (E, O)[p]
        for k3 in range(16):
            assert abs(C[tc][k3] - ref[k1 + 16 * k2 + 256 * k3]) < 1e-6 * np.abs(ref).max(), (tc, k3)
    # ---- natural image (E, O interleaved) + output stage, two rounds
    out = np.full(N, np.nan)
    spec = np.full(M + 1, np.nan + 0j)
    count = np.zeros(N, int)
    X = np.fft.fft(x)

    def slot(kk, p):                                   # kk = compacted bin 0..2047, p = 0 (E) / 1 (O)
        s = 2 * kk + p
        return s + 2 * (s >> 5)

    def compact(k, r):                                 # position of bin k (0..4095) in round r's image
        k3 = k >> 8
        dd = (k3 if k3 < 4 else k3 - 8) if r == 0 else k3 - 4
        assert 0 <= dd < 8, (k, r)
        return (k & 255) + 256 * dd

    for r in range(2):
        img[:] = np.nan
        side = {}
        for tc in range(T):
            k1, p, k2 = tc >> 5, (tc >> 4) & 1, tc & 15
            for dd in range(8):
                k3 = (dd if dd < 4 else dd + 8) if r == 0 else dd + 4
                kk = k1 + 16 * k2 + 256 * dd
                img[slot(kk, p)] = C[tc][k3]
            if r == 0 and k1 == 0 and k2 == 0:
                side[(1024, p)] = C[tc][4]             # bin 1024 (k3 = 4) for round 0's last group
            if r == 1 and k1 == 0 and k2 == 0:
                side[(3072, p)] = C[tc][12]            # bin 3072 (k3 = 12) for round 1's first group
        for t in range(T):
            g, half = t >> 1, t & 1
            k0 = 4 * g + 1024 * r
            kap0 = k0 + 4096 * half
            P = np.zeros(5, complex)
            Q = np.zeros(5, complex)
            for e in range(5):
                k = k0 + e
                kp = (4096 - k) & 4095

                def get(bin_, p):
                    if r == 0 and bin_ == 1024:
                        return side[(1024, p)]
                    if r == 1 and bin_ == 3072:
                        return side[(3072, p)]
                    return img[slot(compact(bin_, r), p)]
                Ek, Ok, Ep, Op = get(k, 0), get(k, 1), get(kp, 0), get(kp, 1)
                w = np.exp(-2j * np.pi * (kap0 + e) / N)           # W_N^(kappa), kappa = k + 4096 half
                u = w * w
                zk = Ek + u * Ok                                      # Z[kappa]
                zm = Ep + np.conj(u) * Op                             # Z[8192 - kappa]
                kap = kap0 + e
                assert abs(zk - np.fft.fft(z)[kap % M]) < 1e-6 * N and abs(zm - np.fft.fft(z)[(M - kap) % M]) < 1e-6 * N
                # split step (chain_f32_dev.hpp split_eval): P = X[kappa], Q = conj X[M - kappa]; the 1/2 is in the window
                s, d = zk + zm, zk - zm
                Tw = s.imag * w + d.real * (w.imag - 1j * w.real)
                P[e] = complex(s.real + Tw.real, d.imag + Tw.imag) / 2
                Q[e] = complex(s.real - Tw.real, d.imag - Tw.imag) / 2
                assert abs(P[e] - X[kap]) < 1e-6 * N and abs(np.conj(Q[e]) - X[(M - kap) % N]) < 1e-6 * N, (r, t, e)
            # split_store(k0' = kap0), MAG_FULL: four aligned groups of four
            for e in range(4):
                for idx, v in ((kap0 + e, P[e]), (N - kap0 - 4 + e, P[4 - e]), (M + kap0 + e, Q[e]), (M - kap0 - 4 + e, Q[4 - e])):
                    out[idx] = abs(v)
                    count[idx] += 1
            for e in range(4):
                spec[kap0 + e] = P[e]
            for e in range(1, 5):
                spec[M - kap0 - e] = np.conj(Q[e])
            if kap0 == 0:
                spec[M] = np.conj(Q[0])
    assert (count == 1).all(), "every bin of the frame is written exactly once"
    return out, spec


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    x = rng.standard_normal(N)
    mag, spec = model(x)
    X = np.fft.fft(x)
    err = np.abs(mag - np.abs(X)).max() / np.abs(X).max()
    err2 = np.abs(spec - X[:M + 1]).max() / np.abs(X).max()
    print(f"w8 index model vs numpy.fft: magnitudes {err:.2e}, half spectrum {err2:.2e}")
    assert err < 1e-10 and err2 < 1e-10
