// sa_common.hpp -- types shared by the host side of the C ABI and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#define SA_NPTS 16384          // real samples per frame
#define SA_MC 8192             // complex points of the packed transform
#define SA_CHUNK 32            // IIR samples per chunk; a thread owns two consecutive chunks
#define SA_NTHREADS 256        // threads per workgroup = chunk pairs per frame
#define SA_MAXSEC 6

// Float IIR plan: the cascade in "predict / scan / recurse" form.
// Section s is the transposed-direct-form-II biquad of scipy.signal.sosfilt
//   y = b0 x + s1;  s1' = b1 x - a1 y + s2;  s2' = b2 x - a2 y
// with state transition A = [[-a1,1],[-a2,0]] and input vector Bv = [b1 - a1 b0, b2 - a2 b0].
// Pc = A^32 (one chunk), P2 = A^64 (one thread = two chunks), Prow = A^1024 (one 16-lane row).
// Everything here, the predictor taps included, is wave-uniform and travels by value in the kernel-argument
// segment (scalar loads; stream-ordered for free; 2.9 KiB of the 4 KiB a launch may carry).
// flags: bit i (i = 0..3) set = in-row scan level 2^i is skipped because P2^(2^i) is below float
// resolution (|entries| < 1e-10: its contribution is < 1e-10 of the state); bit 4 set = the same holds
// for Prow, so the start state of a row is just the previous row's total (no scan over the rows).
#define SA_IIR_SKIP_ROWSCAN 16
// State coordinates.  Everything between the predictor taps and the start states (taps m, Pc, plev, prow,
// the per-lane table) lives in the section's *pole coordinates* z' = T z: T A T^-1 is a scaled rotation for a
// complex pole pair and diagonal for two distinct real poles, so its powers never exceed |pole|^k.  In the
// DF2T coordinates the powers of A = [[-a1,1],[-a2,0]] grow to ~1/angle for poles close to the real axis and
// the float32 scan then loses up to 30x against a sequential evaluation (measured).
// mback = T^-1 (row-major) takes the two start states back to DF2T right before the recursion.  It is the
// identity for first-order, repeated-pole and padding sections.
#define SA_PRED_TAPS 16    // predictor taps per half chunk (block Horner, below)
struct SaIirSecK {
    // Predictor of the NEXT section (zeros for the last one): its chunk end state from zero state,
    //     z = sum_{j<32} A^(31-j) Bv y[j]  =  A^16 (sum_{j<16} m[j] y[j]) + sum_{j<16} m[j] y[16+j],   m[j] = A^(15-j) Bv,
    // accumulated while this section's outputs appear.  Sixteen tap pairs and one matrix (36 scalar registers) stay
    // resident across the recursion; all 32 tap pairs (64 registers) did not fit beside the section's other constants
    // and the compiler re-fetched half of them piecemeal, each fetch an exposed scalar-load round trip.
    // They sit in front of this section's own constants so that everything a section needs is one contiguous run
    // of scalar loads from the kernel-argument segment.
    float mnext[SA_PRED_TAPS][2];
    float p16next[4];      // A^16 of the next section, column-major
    float c[5];            // b0,b1,b2,a1,a2
    int flags;
    float pad[2];
    // every 2x2 matrix below and in SaIirLaneTab::p is stored COLUMN-major (m00, m10, m01, m11): a column is an aligned
    // register pair and a matrix-vector product is two packed FMAs (chain_f32.hip, mv_s)
    float pc[4];           // Pc
    float mback[4];        // T^-1
    float plev[4][4];      // P2^(1,2,4,8)      in-row scan (DPP row_shr 1,2,4,8)
    float prow[4][4];      // Prow^(1,2,4,8)    scan over the 16 rows of a frame
    float pad2[12];        // 96 floats: a whole number of 64-byte scalar-cache lines per section
};

struct SaIirK {
    int nsec;              // padded section count the kernel is compiled for: 0, 2, 4 or 6
    int unit;              // 1: every section is in unit-numerator form b = [1, r1, 1] (cascade gain folded
                           //    into the window table of this plan); the recursion then needs 4 ops, not 5
    float gain;            // the folded cascade gain (1 when unit == 0); informational for tests
    int wingen;            // 1: the window is a0 - a1 cos(2 pi n / (N-1)) and the IIR kernels evaluate it in place
                           //    (SaIirLaneTab::wgen / wcs / wg0); 0: they read the table win_t
    float m0[SA_PRED_TAPS][2]; // predictor of section 0 (SaIirSecK::mnext): taps A^(15-j) Bv ...
    float p16_0[4];            // ... and A^16, column-major
    SaIirSecK sec[SA_MAXSEC];
};

// device-memory part of the plan:
//   p[s][i]   = P2^i, i = lane index inside its 16-lane row (start-state injection per lane)
//   win_t     = 0.5 * window * G, transposed for the chunk layout (win_t[g][t][e] = w[64t + 4g + e]);
//               G = product of the sections' b0 when the plan is in unit-numerator form, else 1
//   wgen, wcs, wg0 = the in-place window generator of the IIR kernels (chain_f32.hip, stage_in_direct):
//               W[64t + 32h + j] = wg0 + wgen[t][2h] * wcs[j][0] + wgen[t][2h+1] * wcs[j][1], all including the
//               factor 0.5 * G; valid when the plan's `wingen` flag is set
struct SaIirLaneTab {
    float p[SA_MAXSEC][16][4];
    float wgen[SA_NTHREADS][4];
    float wcs[SA_CHUNK][2];
    float wg0;
    float pad[3];
    float win_t[SA_NPTS];
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

// Raise a kernel's dynamic-LDS limit once per (kernel, device); thread-safe (specan_abi.cpp).  Not a stream
// operation: doing it on every launch costs host time and cannot be captured into a hipGraph.
hipError_t sa_set_dyn_lds_once(const void *kernel, int bytes);

// launchers (defined in chain_f32.hip / chain_q15.hip)
struct SaF32Tables {
    const float4 *win_b;       // [16][256] 0.5 * window in the pass-A layout of the no-IIR kernel:
                               //   win_b[p][t] = w[512(2p)+2t], w[..+1], w[512(2p+1)+2t], w[..+1]
    const float4 *win_t;       // [16][256] the same, transposed: win_t[g][t] = win_half[64t + 4g .. +3]
    const float4 *twT;         // [6][256]   per-thread twiddle anchors, row i holds two complex values for thread t:
                               //   rows 0..3: W^(1t) W^(2t) | W^(3t) W^(4t) | W^(5t) W^(6t) | W^(7t) W^(8t)   (W = W_8192)
                               //   row 4:     W^(16t) W^(24t)          row 5: W_16384^(4t), W_16384^(4((t+1) & 255))
    const float4 *twB;         // [8][16]    (W_256^(2p*b),   W_256^((2p+1)*b))
    const float2 *twC;         // [25]       W_16384^(1024 blk + e), index blk * 5 + e, blk = 2r + jj (+1): the wave-uniform
                               //            factor of the split-step twiddles
    const SaIirLaneTab *lanetab;   // device
    const SaIirK *iir;             // HOST pointer, copied into the kernel arguments (null = no IIR)
};

// Events of a launch (either may be null), attached to the dispatch packet itself by hipExtLaunchKernel: no marker
// packet between two launches, unlike hipEventRecord around the launch.  `stop` is bound to the completion of the
// call's LAST kernel (ordering of uploads, joins of overlapped launches); `start` to the begin of its FIRST kernel
// (only while sa_set_profiling is on: device time of a call = stop - start).
struct SaLaunchEv {
    hipEvent_t start, stop;
};
hipError_t sa_launch_chain_f32(const float *in, void *out, int batch, int out_kind, const SaF32Tables &t,
                               hipStream_t stream, SaLaunchEv ev);
// the same chain on int16 samples (chain_f32_i16.hip): x = float(sample) * in_scale, then exactly the float32 path
hipError_t sa_launch_chain_f32_i16(const int16_t *in, float in_scale, void *out, int batch, int out_kind, const SaF32Tables &t,
                                   hipStream_t stream, SaLaunchEv ev);

struct SaQ15Tables {
    const int16_t *rom;        // [16384] window ROM
    const uint2 *tw;           // [16384] Q15 twiddles as packed int16 pairs: x = (wr, wi), y = (-wi, wr)
    const uint4 *twrec;        // [2 * kSaTwRecs] the per-lane twiddles of FFT stages 0..2 as 32-byte records {w1, w2, w3, pad}:
                               // records 0..4095: stage 0, exponent bf; 4096..5119: stage 1, exponent 4 j'; 5120..5375: stage 2, 16 j''
};
constexpr int kSaTwRecs = 4096 + 1024 + 256;

hipError_t sa_launch_filter_q15(const int16_t *in, int16_t *out_time, int batch, const SaQ15Params &p,
                                const SaQ15Tables &t, hipStream_t stream, SaLaunchEv ev);
hipError_t sa_launch_fft_q15(const int16_t *in_time, int16_t *out_iq, int batch, bool apply_window,
                             const SaQ15Params &p, const SaQ15Tables &t, hipStream_t stream, SaLaunchEv ev);
