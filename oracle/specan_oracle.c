/*
 * specan_oracle.c -- CPU restatement of the reference signal path. TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product path (fpga_real_time_fft_analyzer_amd/) never links or calls it.
 *
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference; new/ = SDR_v2.srcs/sources_1/new, imp/ = SDR_v2.srcs/sources_1/imports/new).
 *
 * Parity status:
 *   window / biquad / cascade: follows the RTL arithmetic; pinned by the hand KATs of
 *     SURVEY.md section 8(a) and by the ROM new/hann.vhd (byte-identical regeneration).
 *     The reference holds no DSP test vectors of its own.
 *   fixed-point FFT: the reference FFT is an encrypted Xilinx IP (ip/xfft_0/xfft_0.xci:5-6),
 *     no model in the tree => PARITY UNPINNED at that boundary.  or_fxfft16k() is this
 *     build's own published spec ("SA-FXFFT-1"), checked to tolerance against fft(x)/N.
 *   float chain: scipy.signal.sosfilt + numpy.fft (BASELINE.json north_star) are the oracle;
 *     or_sosfilt_f64 restates scipy's DF2T loop and is checked against scipy in tests.
 */
#include <stdint.h>
#include <stddef.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define OR_N 16384

/* ---------------------------------------------------------------- window */

/* scripts/hann_coeff.py:3-4 -- hann[n] = 0.5*(1-cos(2*pi*n/(N-1))) (symmetric Hann). */
void or_hann_f64(double *w, int n)
{
    for (int i = 0; i < n; ++i)
        w[i] = 0.5 * (1.0 - cos(2.0 * M_PI * (double)i / (double)(n - 1)));
}

/* scripts/hann_coeff.py:5 -- q15 = round((hann-0.5)*2**16).astype(int16).
 * numpy round = half-to-even (rint); astype(int16) wraps +32768 -> -32768 (quirk Q1,
 * visible in new/hann.vhd entries 8178..8205). */
void or_hann_rom_q15(int16_t *rom, int n)
{
    for (int i = 0; i < n; ++i) {
        double h = 0.5 * (1.0 - cos(2.0 * M_PI * (double)i / (double)(n - 1)));
        double r = rint((h - 0.5) * 65536.0);
        int32_t v = (int32_t)r;
        rom[i] = (int16_t)(uint16_t)(v & 0xFFFF);
    }
}

/* new/hann8192.vhd:36-39 -- product = sample*coef (32 bit);
 * out = resize(product(31 downto 15) + product(14), 16).
 * numeric_std resize(signed) keeps the sign bit and the low 15 bits. */
static inline int16_t win_q15_1(int16_t x, int16_t c)
{
    int32_t p = (int32_t)x * (int32_t)c;
    int32_t hi = p >> 15;                 /* 17-bit signed slice */
    int32_t r = hi + ((p >> 14) & 1);     /* + rounding bit, 17-bit */
    uint32_t r17 = (uint32_t)r & 0x1FFFFu;
    uint32_t out = (r17 & 0x7FFFu) | ((r17 >> 16) & 1u) << 15;
    return (int16_t)(uint16_t)out;
}

int16_t or_win_q15_1(int16_t x, int16_t c) { return win_q15_1(x, c); }

/* Frame-aligned window (SURVEY quirks Q3/Q4 not reproduced): y[n] = rnd(x[n]*rom[n]). */
void or_window_q15(const int16_t *x, const int16_t *rom, int16_t *y, int n)
{
    for (int i = 0; i < n; ++i) y[i] = win_q15_1(x[i], rom[i]);
}

/* Second integer window mode (SURVEY quirk Q2 "evident intent"): unsigned Q16 Hann
 * w_u16 = rom + 32768; y = floor((x*w_u16 + 2^15) / 2^16), saturating is never needed
 * because w_u16 < 65536. */
void or_window_u16(const int16_t *x, const int16_t *rom, int16_t *y, int n)
{
    for (int i = 0; i < n; ++i) {
        int32_t w = (int32_t)rom[i] + 32768;          /* 0..65535 */
        int64_t p = (int64_t)x[i] * w + 32768;
        y[i] = (int16_t)(p >> 16);
    }
}

/* ---------------------------------------------------------------- integer IIR */

/* new/filter_iir_cust.vhd:96-100 -- each 24-bit product is sliced (22 downto 7):
 * floor(v*c/128) mod 2^16. */
static inline uint16_t q7_term(int16_t v, int8_t c)
{
    int32_t p = (int32_t)v * (int32_t)c;
    return (uint16_t)((p >> 7) & 0xFFFF);
}

typedef struct { int16_t x1, x2, y1, y2; } biq_state;

/* One biquad step.  Tap naming follows the RTL ports (new/filter_iir_cust.vhd:104-108):
 * B2*x[n] + B1*x[n-1] + B0*x[n-2] - A0*y[n-2] - A1*y[n-1]; the sum wraps at 16 bit.
 * coefficient order in c[]: B0,B1,B2,A0,A1,A2 (A2 unused, quirk Q7). */
static inline int16_t biq_step(biq_state *s, const int8_t *c, int16_t x)
{
    uint16_t acc = (uint16_t)(q7_term(x, c[2]) + q7_term(s->x1, c[1]) + q7_term(s->x2, c[0])
                              - q7_term(s->y2, c[3]) - q7_term(s->y1, c[4]));
    int16_t y = (int16_t)acc;
    s->x2 = s->x1; s->x1 = x;
    s->y2 = s->y1; s->y1 = y;
    return y;
}

/* Single biquad over a stream, zero state at start (KAT helper). */
void or_biquad_q7(const int16_t *x, int16_t *y, int n, const int8_t *c6)
{
    biq_state s = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) y[i] = biq_step(&s, c6, x[i]);
}

/* new/filter_iir12_cust.vhd:68-240 (and imp/filter_iir12.vhd:38-137): six biquads in
 * series, stages 1,3,5 = coefficient set 0 (c[0..5]), stages 2,4,6 = set 1 (c[6..11]).
 * State is zero at frame start (quirk Q5 handling) and carried across the frame. */
void or_iir12_q7(const int16_t *x, int16_t *y, int n, const int8_t *c12)
{
    biq_state s[6];
    memset(s, 0, sizeof s);
    for (int i = 0; i < n; ++i) {
        int16_t v = x[i];
        for (int k = 0; k < 6; ++k) v = biq_step(&s[k], c12 + ((k & 1) ? 6 : 0), v);
        y[i] = v;
    }
}

/* imp/filter_pkg.vhd:54-68 -- default coefficients, order B0,B1,B2,A0,A1,A2 per set. */
static const int8_t k_default_q7[12] = { -14, 0, 14, 107, 21, 127,  -15, 0, 15, 107, -21, 127 };
void or_default_coeffs_q7(int8_t *c12) { memcpy(c12, k_default_q7, 12); }

/* Wide integer mode (north-star, not in the RTL; spec defined here):
 * up to 6 independent sections, int16 Q2.14 coefficients in scipy row order
 * [b0,b1,b2,a0,a1,a2] (a0 ignored, assumed 1.0 = 16384), direct form I,
 * 64-bit accumulator, one round-half-up shift by 14, saturation to int16. */
void or_iir_sos_q14(const int16_t *x, int16_t *y, int n, const int16_t *sos, int nsec)
{
    biq_state s[6];
    memset(s, 0, sizeof s);
    for (int i = 0; i < n; ++i) {
        int16_t v = x[i];
        for (int k = 0; k < nsec; ++k) {
            const int16_t *c = sos + 6 * k;
            int64_t acc = (int64_t)c[0] * v + (int64_t)c[1] * s[k].x1 + (int64_t)c[2] * s[k].x2
                        - (int64_t)c[4] * s[k].y1 - (int64_t)c[5] * s[k].y2;
            acc = (acc + 8192) >> 14;
            if (acc > 32767) acc = 32767;
            if (acc < -32768) acc = -32768;
            s[k].x2 = s[k].x1; s[k].x1 = v;
            s[k].y2 = s[k].y1; s[k].y1 = (int16_t)acc;
            v = (int16_t)acc;
        }
        y[i] = v;
    }
}

/* ---------------------------------------------------------------- fixed-point FFT (SA-FXFFT-1) */
/*
 * Stands where ip/xfft_0 stands (16384 points xfft_0.xci:12, 16-bit data :17-18, 16-bit
 * twiddles :19, scaled :20, truncation :21, natural order :27; forward, imag input = 0
 * new/command_control.vhd:123).  Spec:
 *   radix-4 decimation-in-frequency, 7 stages, span L = N/4^s, q = L/4.
 *   butterfly in 32-bit: t0=a0+a1+a2+a3, t1=a0-i*a1-a2+i*a3, t2=a0-a1+a2-a3, t3=a0+i*a1-a2-i*a3
 *   scale: u_i = t_i >> 2 (arithmetic, truncation) -- 1/4 per stage, 1/N overall
 *   twiddle: exponent e = i*j*(N/L); e == 0 -> pass-through; else
 *            y.re = (u.re*wr - u.im*wi) >> 15, y.im = (u.re*wi + u.im*wr) >> 15 (truncation),
 *            saturated to int16, with (wr,wi) = clamp16(rint(32768*cos), rint(-32768*sin)).
 *   result is stored digit-reversed by the in-place DIF; final pass restores natural order.
 */
static int16_t *g_twr = NULL, *g_twi = NULL;

static inline int16_t clamp16(int32_t v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : (int16_t)v); }

void or_fxfft_twiddles(int16_t *wr, int16_t *wi)
{
    for (int m = 0; m < OR_N; ++m) {
        double a = 2.0 * M_PI * (double)m / (double)OR_N;
        wr[m] = clamp16((int32_t)rint(32768.0 * cos(a)));
        wi[m] = clamp16((int32_t)rint(-32768.0 * sin(a)));
    }
}

static void fx_init(void)
{
    if (g_twr) return;
    g_twr = (int16_t *)malloc(sizeof(int16_t) * OR_N);
    g_twi = (int16_t *)malloc(sizeof(int16_t) * OR_N);
    or_fxfft_twiddles(g_twr, g_twi);
}

static inline void fx_twiddle(int32_t ur, int32_t ui, int e, int16_t *yr, int16_t *yi)
{
    if (e == 0) { *yr = clamp16(ur); *yi = clamp16(ui); return; }
    int32_t wr = g_twr[e], wi = g_twi[e];
    int32_t pr = (ur * wr - ui * wi) >> 15;
    int32_t pi = (ur * wi + ui * wr) >> 15;
    *yr = clamp16(pr);
    *yi = clamp16(pi);
}

/* in: N real int16 samples; out: N x {re,im} int16, natural bin order (frame layout of
 * imp/sequ2.vhd:153 / gui.py:250-260: re lo,hi then im lo,hi, little endian). */
void or_fxfft16k(const int16_t *in, int16_t *out_iq)
{
    fx_init();
    int16_t *re = (int16_t *)malloc(sizeof(int16_t) * OR_N);
    int16_t *im = (int16_t *)malloc(sizeof(int16_t) * OR_N);
    for (int i = 0; i < OR_N; ++i) { re[i] = in[i]; im[i] = 0; }
    for (int L = OR_N; L >= 4; L >>= 2) {
        int q = L >> 2, tw = OR_N / L;
        for (int base = 0; base < OR_N; base += L) {
            for (int j = 0; j < q; ++j) {
                int i0 = base + j, i1 = i0 + q, i2 = i1 + q, i3 = i2 + q;
                int32_t ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
                int32_t cr = re[i2], ci = im[i2], dr = re[i3], di = im[i3];
                int32_t t0r = ar + br + cr + dr, t0i = ai + bi + ci + di;
                int32_t t1r = ar + bi - cr - di, t1i = ai - br - ci + dr;   /* a - i b - c + i d */
                int32_t t2r = ar - br + cr - dr, t2i = ai - bi + ci - di;
                int32_t t3r = ar - bi - cr + di, t3i = ai + br - ci - dr;   /* a + i b - c - i d */
                fx_twiddle(t0r >> 2, t0i >> 2, 0,          &re[i0], &im[i0]);
                fx_twiddle(t1r >> 2, t1i >> 2, 1 * j * tw, &re[i1], &im[i1]);
                fx_twiddle(t2r >> 2, t2i >> 2, 2 * j * tw, &re[i2], &im[i2]);
                fx_twiddle(t3r >> 2, t3i >> 2, 3 * j * tw, &re[i3], &im[i3]);
            }
        }
    }
    /* base-4 digit reversal (7 digits) back to natural order */
    for (int k = 0; k < OR_N; ++k) {
        int r = 0, t = k;
        for (int d = 0; d < 7; ++d) { r = (r << 2) | (t & 3); t >>= 2; }
        out_iq[2 * r] = re[k];
        out_iq[2 * r + 1] = im[k];
    }
    free(re); free(im);
}

/* Whole Q15 chain for one batch: window (mode 0 = RTL signed ROM, 1 = unsigned Q16 Hann),
 * filter select byte as new/command_control.vhd:53-58 (0x00 default, 0xA1 custom q7,
 * 0xB1 bypass; 0xA2 = wide q14 mode, build extension), FFT.
 * time_out (optional) receives the FFT input (post window/IIR) for debugging. */
int or_chain_q15(const int16_t *in, int16_t *out_iq, int16_t *time_out, int B,
                 const int16_t *rom, int win_mode, int filter_cmd,
                 const int8_t *c12, const int16_t *sos_q14, int nsec)
{
    int16_t *a = (int16_t *)malloc(sizeof(int16_t) * OR_N);
    int16_t *b = (int16_t *)malloc(sizeof(int16_t) * OR_N);
    for (int f = 0; f < B; ++f) {
        const int16_t *x = in + (size_t)f * OR_N;
        if (win_mode == 0) or_window_q15(x, rom, a, OR_N); else or_window_u16(x, rom, a, OR_N);
        const int16_t *t = a;
        if (filter_cmd == 0x00) { or_iir12_q7(a, b, OR_N, k_default_q7); t = b; }
        else if (filter_cmd == 0xA1) { or_iir12_q7(a, b, OR_N, c12); t = b; }
        else if (filter_cmd == 0xA2) { or_iir_sos_q14(a, b, OR_N, sos_q14, nsec); t = b; }
        else if (filter_cmd != 0xB1) { free(a); free(b); return -1; }
        if (time_out) memcpy(time_out + (size_t)f * OR_N, t, sizeof(int16_t) * OR_N);
        or_fxfft16k(t, out_iq + (size_t)f * OR_N * 2);
    }
    free(a); free(b);
    return 0;
}

/* ---------------------------------------------------------------- float IIR (scipy restatement) */

/* scipy.signal.sosfilt (scipy 1.15.3, _sosfilt.pyx inner loop), transposed direct form II,
 * float64, zero initial state, rows [b0,b1,b2,a0,a1,a2] with a0 == 1 (scipy normalises on
 * design; the north-star oracle per BASELINE.json).  Call site in the reference's host code:
 * gui.py:108-157 designs the SOS that this consumes. */
void or_sosfilt_f64(const double *sos, int nsec, const double *x, double *y, int n)
{
    double z[12][2];
    memset(z, 0, sizeof z);
    for (int i = 0; i < n; ++i) {
        double v = x[i];
        for (int s = 0; s < nsec; ++s) {
            const double *c = sos + 6 * s;
            double xn = v;
            v = c[0] * xn + z[s][0];
            z[s][0] = c[1] * xn - c[4] * v + z[s][1];
            z[s][1] = c[2] * xn - c[5] * v;
        }
        y[i] = v;
    }
}

/* Same recurrence evaluated in float32 (what a straight GPU port would compute); used only to
 * size the tolerance of the chunked GPU form against a sequential fp32 evaluation. */
void or_sosfilt_f32(const float *sos, int nsec, const float *x, float *y, int n)
{
    float z[12][2];
    memset(z, 0, sizeof z);
    for (int i = 0; i < n; ++i) {
        float v = x[i];
        for (int s = 0; s < nsec; ++s) {
            const float *c = sos + 6 * s;
            float xn = v;
            v = c[0] * xn + z[s][0];
            z[s][0] = c[1] * xn - c[4] * v + z[s][1];
            z[s][1] = c[2] * xn - c[5] * v;
        }
        y[i] = v;
    }
}

/* ---------------------------------------------------------------- float FFT (double, radix-2) */

/* Forward DFT, natural order, double precision, N = power of two.  Stands for numpy.fft.fft in
 * places where the C side must be self-contained (cpu_baseline "port" leg of bench.py). */
void or_fft_f64(double *re, double *im, int n)
{
    /* bit reversal */
    for (int i = 1, j = 0; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
    }
    for (int len = 2; len <= n; len <<= 1) {
        double ang = -2.0 * M_PI / (double)len;
        for (int i = 0; i < n; i += len) {
            for (int k = 0; k < len / 2; ++k) {
                double wr = cos(ang * k), wi = sin(ang * k);
                int a = i + k, b = a + len / 2;
                double tr = re[b] * wr - im[b] * wi, ti = re[b] * wi + im[b] * wr;
                re[b] = re[a] - tr; im[b] = im[a] - ti;
                re[a] += tr; im[a] += ti;
            }
        }
    }
}

/* Whole float chain for a batch, double precision internally:
 * mag[f][k] = |FFT(sosfilt(x*hann))[k]|, k = 0..N-1 (all N bins, gui.py:294-305 axis). */
void or_chain_f64(const float *in, float *out_mag, int B, const double *sos, int nsec)
{
    static double *hann = NULL;
    if (!hann) { hann = (double *)malloc(sizeof(double) * OR_N); or_hann_f64(hann, OR_N); }
    double *a = (double *)malloc(sizeof(double) * OR_N);
    double *b = (double *)malloc(sizeof(double) * OR_N);
    double *c = (double *)malloc(sizeof(double) * OR_N);
    for (int f = 0; f < B; ++f) {
        const float *x = in + (size_t)f * OR_N;
        for (int i = 0; i < OR_N; ++i) a[i] = (double)x[i] * hann[i];
        if (nsec > 0) or_sosfilt_f64(sos, nsec, a, b, OR_N); else memcpy(b, a, sizeof(double) * OR_N);
        memset(c, 0, sizeof(double) * OR_N);
        or_fft_f64(b, c, OR_N);
        float *o = out_mag + (size_t)f * OR_N;
        for (int i = 0; i < OR_N; ++i) o[i] = (float)sqrt(b[i] * b[i] + c[i] * c[i]);
    }
    free(a); free(b); free(c);
}
