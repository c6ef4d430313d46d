// sa_common.hpp -- types shared by the host side of the C ABI and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SA_NPTS 16384          // real samples per frame
#define SA_MC 8192             // complex points of the packed transform
#define SA_CHUNK 64            // IIR samples per thread
#define SA_NCHUNK 256          // chunks per frame = threads per workgroup
#define SA_MAXSEC 6

// Float IIR plan: the cascade in "predict / scan / recurse" form.
// Section s is the transposed-direct-form-II biquad of scipy.signal.sosfilt
//   y = b0 x + s1;  s1' = b1 x - a1 y + s2;  s2' = b2 x - a2 y
// with state transition A = [[-a1,1],[-a2,0]], input vector Bv = [b1 - a1 b0, b2 - a2 b0].
// P = A^64 is the per-chunk transition.
struct SaIirSecPlan {
    float c[8];            // b0,b1,b2,a1,a2,0,0,0
    float plev[6][4];      // P^(2^i), i = 0..5, row-major p00,p01,p10,p11  (in-wave scan)
    float p64[4];          // P^64                                          (cross-wave carry)
    float m[2][SA_CHUNK];  // predictor: state after a chunk from zero state = sum_j m[.][j] x[j]
    float ppow[64][4];     // P^l, l = 0..63                                (carry injection per lane)
};

struct SaIirPlan {
    int nsec;
    int pad[3];
    SaIirSecPlan sec[SA_MAXSEC];
};

// Integer-path parameters passed by value (kernarg => stream-ordered for free).
struct SaQ15Params {
    int win_mode;          // SA_WIN_*
    int filter;            // SA_FILTER_* byte
    int nsec_wide;
    int pad;
    int8_t c12[12];        // active q7 coefficients (default or custom), wire order
    int8_t pad2[4];
    int16_t sos_q14[SA_MAXSEC * 6];
};

// launchers (defined in chain_f32.hip / chain_q15.hip)
struct SaF32Tables {
    const float *win_half;     // [16384] 0.5 * window
    const float2 *twA;         // [32][256]  W_8192^(k1*m2)
    const float2 *twB;         // [16][16]   W_256^(c*b)
    const float2 *twP;         // [4097]     W_16384^k
    const SaIirPlan *plan;     // device copy (may be null when no IIR)
};

hipError_t sa_launch_chain_f32(const float *in, void *out, int batch, int out_kind, bool iir,
                               const SaF32Tables &t, hipStream_t stream);

struct SaQ15Tables {
    const int16_t *rom;        // [16384] window ROM
    const uint32_t *tw;        // [16384] packed Q15 twiddles (wr | wi << 16)
};

hipError_t sa_launch_filter_q15(const int16_t *in, int16_t *out_time, int batch, const SaQ15Params &p,
                                const SaQ15Tables &t, hipStream_t stream);
hipError_t sa_launch_fft_q15(const int16_t *in_time, int16_t *out_iq, int batch, bool apply_window,
                             const SaQ15Params &p, const SaQ15Tables &t, hipStream_t stream);
