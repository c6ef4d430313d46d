// chain_f32.hip -- fused float signal path for gfx950 (MI355X):
//     Hann window -> 6-section biquad cascade -> 16384-point real FFT -> magnitude
// replacing new/hann8192.vhd -> new/filter_iir12_cust.vhd -> ip/xfft_0 of the reference with one
// pass over HBM (read 64 KiB, write 64 KiB per frame).  One 256-thread workgroup per frame.
//
// Shape of the computation (DESIGN.md sections 3-5):
//   * LDS holds HALF a frame at a time (35 KiB per workgroup) so that four workgroups share a CU
//     (16 waves): every exchange is done in two index-split rounds.  The kernel is latency bound
//     at lower occupancy (profiles/r1_phase_stamps_iir.txt).
//   * thread t owns samples [64t, 64t+64) for the IIR as two chunks of 32 held as float pairs
//     (chunk A in .x, chunk B in .y) so the serial recursion runs on packed-fp32 instructions.
//   * per section: the end state of every chunk from zero state is a dot product ("predict", fused
//     into the previous section's recursion loop); an affine scan over the 512 chunks of the frame
//     (in-row DPP shifts, one LDS hop for the 16 row totals) turns those into true start states;
//     then the exact DF2T recursion of scipy.signal.sosfilt runs from them.
//   * the real FFT is an 8192-point complex FFT of z[m] = x[2m] + i x[2m+1] factored 32 x 16 x 16,
//     each factor in registers (fft_regs.hpp), then the split step X[k] = Xe[k] + W_N^k Xo[k].
//     Taps and scan matrices live in the section's pole coordinates (sa_common.hpp): float32 accuracy
//     then matches a sequential evaluation also for poles next to the real axis.
//   * all 16384 magnitudes are written (upper half mirrored) as aligned 16-byte nontemporal stores.
// The factor 1/2 of the split step is folded into the window table (exact in binary fp).
#include "chain_f32_dev.hpp"

// This file is compiled twice: as it is (float32 frames in: sa_process_f32) and, through chain_f32_i16.hip, with
// SA_F32_INPUT_I16 = 1 (int16 samples in -- what the board's ADC path delivers, imp/dsp_system_top.vhd:435 -- converted
// and scaled in the stage-in: sa_process_f32_i16).  Everything behind the stage-in is the same code.
#ifndef SA_F32_INPUT_I16
#define SA_F32_INPUT_I16 0
#endif

namespace {

#if SA_F32_INPUT_I16
typedef int16_t sa_in_t;
#define SA_IN_SCALE_PARAM const float in_scale,        // only the int16 instantiations carry the scale
#define SA_IN_SCALE_ARG in_scale,
#else
typedef float sa_in_t;
#define SA_IN_SCALE_PARAM
#define SA_IN_SCALE_ARG
#endif

constexpr int kThreads = 256;
constexpr int kLdsComplex = 16 * 272;                 // half-frame exchange image (4352 complex)
constexpr int kScrOff = kLdsComplex * 8;              // scan scratch: 6 sections x 16 rows x float2
constexpr int kSideOff = kScrOff + 6 * 16 * 8;        // one complex side slot (Z[6144])
constexpr int kLaneOff = kSideOff + 16;               // (two complex side slots) then the per-lane matrices P2^i: 6 x 16 x float4
constexpr int kLdsBytes = kLaneOff + 6 * 16 * 16;
constexpr int kLdsOneRound = 65536;                   // the bypassed chain at small batches: a whole float32 frame at once
static_assert(kLdsBytes <= kLdsOneRound, "the exchange images and side slots live inside the one-round image");

// The two-component scan state travels as ONE register pair and every 2x2 matrix is stored column-major (a column is
// an aligned register pair): a matrix-vector product is two packed FMAs, column x broadcast component -- for wave-uniform
// matrices straight from their scalar registers.  (Round 3, tools/ubench/valu_throughput.hip: on gfx950 a plain fp32
// instruction with a scalar or DPP operand costs what a packed one costs, so four scalar FMAs per product were four
// packed-instruction slots.)
//   mv_s / mv_v: r = add + c0 * v.x + c1 * v.y with wave-uniform (SGPR) / per-lane (VGPR) columns;  mv_acc_s: in place
__device__ __forceinline__ v2f mv_s(const v2f c0, const v2f c1, const v2f v, const v2f add)
{
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %3, %4 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
        : "=&v"(r) : "s"(c0), "s"(c1), "v"(v), "v"(add));
    return r;
}
__device__ __forceinline__ v2f mv_v(const v2f c0, const v2f c1, const v2f v, const v2f add)
{
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %3, %4 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
        : "=&v"(r) : "v"(c0), "v"(c1), "v"(v), "v"(add));
    return r;
}
__device__ __forceinline__ v2f mv0_s(const v2f c0, const v2f c1, const v2f v)          // c0 * v.x + c1 * v.y
{
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %3 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
        : "=&v"(r) : "s"(c0), "s"(c1), "v"(v));
    return r;
}
__device__ __forceinline__ void mv_acc_s(v2f &z, const v2f c0, const v2f c1, const v2f v)
{
    asm("v_pk_fma_f32 %0, %1, %3, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
        : "+v"(z) : "s"(c0), "s"(c1), "v"(v));
}

// z <- z + P * shifted(z): one Kogge-Stone level of the affine scan inside a row (two DPP moves, two packed FMAs)
template <int N, typename MatT>
__device__ __forceinline__ void scan_level(v2f &z, const MatT &p)
{
    const v2f u = {row_shr<N>(z.x), row_shr<N>(z.y)};
    mv_acc_s(z, v2f{p[0], p[1]}, v2f{p[2], p[3]}, u);
}

__device__ __forceinline__ float mul_to(float a, float b)
{
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ---------------------------------------------------------------------------------------------
// Stage-in for the IIR, two rounds.  Round h brings chunk h (32 samples = 128 B) of every thread:
// the samples go HBM -> LDS directly (global_load_lds_dwordx4: no VGPRs, no ds_write), 1 KiB = 8 rows
// of 128 B per wave instruction.  The LDS image is linear per instruction, so the XOR swizzle (16-byte
// column c of row r stored at c ^ ((r >> 1) & 7): conflict-free ds_read_b128 at a 128-byte row pitch)
// is applied to the per-lane SOURCE address.  Each thread then reads its own 32 consecutive samples
// and multiplies by the window, which the host stored transposed (wint[g][t] = window[64t + 4g .. +3])
// so that its loads are coalesced in this layout.
// Thread t ends with d[j] = (x[64t + j], x[64t + 32 + j]) * window.
// the LDS-DMA of round h (see stage_in_chunks): 8 x 1 KiB per wave, issued at raised priority so that the
// requests leave ahead of the other workgroups' arithmetic
__device__ __forceinline__ void dma_chunk_half(const float *__restrict__ xin, int h, unsigned char *smem, int lane, int wave)
{
    const int rl = lane >> 3;                                  // row inside the slab
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n = wave * 8 + i;                            // slab: rows 8n .. 8n+7
        const int r = 8 * n + rl;
        const int lc = (lane & 7) ^ ((r >> 1) & 7);
        const float *src = xin + r * 64 + h * 32 + lc * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(smem + n * 1024), 16, 0, SA_DMA_AUX);
    }
    __builtin_amdgcn_s_setprio(0);
}

// int16 samples in (SA_F32_INPUT_I16): a thread's 64 samples are 128 bytes, so ONE round of the same LDS-DMA brings the
// whole frame (32 KiB): row r = thread r's samples, same wave-private slabs, same XOR swizzle of the 16-byte columns.
// Column g of a row holds samples 8g .. 8g+7: g < 4 is chunk A, g >= 4 chunk B.  x = float(sample) * scale is rounded once
// and then takes the window exactly as a float32 input sample does, so the results are those of sa_process_f32 on the
// converted frame, bit for bit.
template <bool WINGEN>
__device__ __forceinline__ void stage_in_chunks(const int16_t *__restrict__ xin, const float in_scale,
                                                const float4 *__restrict__ wint, const SaIirLaneTab *__restrict__ lt,
                                                unsigned char *smem, int t, v2f (&d)[32])
{
    const uint4 *lds4 = reinterpret_cast<const uint4 *>(smem);
    const int lane = t & 63, wave = t >> 6, rl = lane >> 3;
    float4 pq = make_float4(0.f, 0.f, 0.f, 0.f);
    float g0 = 0.f;
    if constexpr (WINGEN) {
        pq = *reinterpret_cast<const float4 *>(&lt->wgen[t][0]);
        g0 = lt->wg0;
    }
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n = wave * 8 + i;                            // slab: rows 8n .. 8n+7
        const int r = 8 * n + rl;
        const int lc = (lane & 7) ^ ((r >> 1) & 7);
        const int16_t *src = xin + r * 64 + lc * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(smem + n * 1024), 16, 0, SA_DMA_AUX);
    }
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const int sw = (t >> 1) & 7;
    const v2f Pw = {pq.x, pq.z}, Qw = {pq.y, pq.w}, G0 = {g0, g0}, sc = {in_scale, in_scale};
#pragma unroll
    for (int g = 0; g < 4; ++g) {                              // samples 8g .. 8g+7 of chunk A and of chunk B
        const uint4 qa = lds4[t * 8 + (g ^ sw)], qb = lds4[t * 8 + ((g + 4) ^ sw)];
        const unsigned ua[4] = {qa.x, qa.y, qa.z, qa.w}, ub[4] = {qb.x, qb.y, qb.z, qb.w};
#pragma unroll
        for (int u = 0; u < 2; ++u) {                          // four samples of either chunk at a time
            float4 wa = make_float4(0.f, 0.f, 0.f, 0.f), wb = wa;
            if constexpr (!WINGEN) {                           // table window: win_t[g'][t] = window at 64 t + 4 g' .. + 3
                wa = wint[(2 * g + u) * 256 + t];
                wb = wint[(8 + 2 * g + u) * 256 + t];
            }
            const float fa[4] = {wa.x, wa.y, wa.z, wa.w}, fb[4] = {wb.x, wb.y, wb.z, wb.w};
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const int e = 4 * u + e4, j = 8 * g + e;
                const int ia = (e & 1) ? (int)ua[e >> 1] >> 16 : (int)(short)(ua[e >> 1] & 0xFFFFu);
                const int ib = (e & 1) ? (int)ub[e >> 1] >> 16 : (int)(short)(ub[e >> 1] & 0xFFFFu);
                const v2f x = v2f{(float)ia, (float)ib} * sc;   // rounded once: the float32 sample
                v2f w;
                if constexpr (WINGEN) {
                    const v2f cs = {lt->wcs[j][0], lt->wcs[j][1]};
                    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
                        "v_pk_fma_f32 %0, %4, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
                        : "=&v"(w) : "v"(Pw), "s"(cs), "v"(G0), "v"(Qw));
                } else {
                    w = v2f{fa[e4], fb[e4]};
                }
                d[j] = x * w;
            }
        }
    }
}

template <bool WINGEN>
__device__ __forceinline__ void stage_in_chunks(const float *__restrict__ xin, const float4 *__restrict__ wint,
                                                const SaIirLaneTab *__restrict__ lt, unsigned char *smem, int t,
                                                v2f (&d)[32])
{
    const float4 *lds4 = reinterpret_cast<const float4 *>(smem);
    const int lane = t & 63, wave = t >> 6;
    // WINGEN: the window is a0 - a1 cos(2 pi n / (N-1)) (Hann, Hamming; what scripts/hann_coeff.py:3-4 generates) and is
    // evaluated in place by the angle-addition formula: with n = 64 t + 32 h + j,
    //   W[n] = G0 + P_h c_j + Q_h s_j,   c_j = cos(theta j), s_j = sin(theta j) wave-uniform (scalar loads),
    //   (P_h, Q_h) = S a1 (-cos, sin)(theta (64 t + 32 h)) per thread and chunk, G0 = S a0, S = 0.5 * cascade gain.
    // The 64 KiB per-frame read of the window table (L2 -> L1, 16 more loads per thread) is gone.  !WINGEN (any other
    // window): the transposed table.
    float4 pq = make_float4(0.f, 0.f, 0.f, 0.f);
    float g0 = 0.f;
    if constexpr (WINGEN) {
        pq = *reinterpret_cast<const float4 *>(&lt->wgen[t][0]);
        g0 = lt->wg0;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        // The image is WAVE-PRIVATE in this phase: wave w requests rows 64w .. 64w+63 (slabs 8w .. 8w+7) and its
        // own threads are the only readers of those rows.  No workgroup barrier then: a wave waits for its own
        // DMA (vmcnt) and, before overwriting the rows with round 1, for its own reads of round 0 (lgkmcnt).
        // Three barriers fewer per frame; a wave delayed on its SIMD no longer holds the other three here.
        if (h == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dma_chunk_half(xin, h, smem, lane, wave);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // nothing of the window arithmetic below may be scheduled above the wait (register-only instructions do
        // cross an asm statement): computed early, the 32 window values of the round sit in registers and spill
        __builtin_amdgcn_sched_barrier(0);
        const int sw = (t >> 1) & 7;
        // WINGEN: round 0 evaluates the window of BOTH chunks as pairs (chunk A, chunk B) -- two packed FMAs per pair with
        // the wave-uniform (c_j, s_j) in a scalar pair -- multiplies chunk A and parks chunk B's factor in the pair's
        // other half, where round 1 multiplies it in place.  (Per sample it used to be two scalar-operand FMAs, each of
        // which costs a packed-instruction slot on gfx950: tools/ubench/valu_throughput.hip.)
        const v2f Pw = {pq.x, pq.z}, Qw = {pq.y, pq.w}, G0 = {g0, g0};
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            // two batches of four units: all eight in flight at once push the kernel over 128 VGPRs
            if (g == 4) __builtin_amdgcn_sched_barrier(0);
            const float4 q = lds4[t * 8 + (g ^ sw)];
            const float qv[4] = {q.x, q.y, q.z, q.w};
            if constexpr (WINGEN) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = 4 * g + e;
                    if (h == 0) {
                        const v2f cs = {lt->wcs[j][0], lt->wcs[j][1]};
                        v2f w;
                        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
                            "v_pk_fma_f32 %0, %4, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
                            : "=&v"(w) : "v"(Pw), "s"(cs), "v"(G0), "v"(Qw));
                        d[j].x = mul_to(qv[e], w.x);
                        d[j].y = w.y;
                    } else {
                        d[j].y = mul_to(qv[e], d[j].y);
                    }
                }
            } else {
                const float4 w = wint[(8 * h + g) * 256 + t];
                // mul_to: one v_mul_f32 straight into its half of the (chunk A, chunk B) pair.  Left to the
                // SLP vectoriser the two rounds become v_pk_mul_f32 on re-paired operands: ~100 v_mov per thread.
                if (h == 0) {
                    d[4 * g + 0].x = mul_to(q.x, w.x);
                    d[4 * g + 1].x = mul_to(q.y, w.y);
                    d[4 * g + 2].x = mul_to(q.z, w.z);
                    d[4 * g + 3].x = mul_to(q.w, w.w);
                } else {
                    d[4 * g + 0].y = mul_to(q.x, w.x);
                    d[4 * g + 1].y = mul_to(q.y, w.y);
                    d[4 * g + 2].y = mul_to(q.z, w.z);
                    d[4 * g + 3].y = mul_to(q.w, w.w);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Predictor taps.  (m1, m2) is one wave-uniform tap pair in an aligned SGPR pair; the chunk-end states of chunk A and
// chunk B (from zero state) accumulate as (z1, z2) pairs: nA += tap * y.x, nB += tap * y.y.  Eight taps in one statement
// (the compiler pads a wait state after every asm statement whose output the next one reads; 32 single-tap statements =
// 32 pads per section).  Two accumulator sets alternate: a dependent FMA every fourth instruction.
__device__ __forceinline__ void tap_fma8(v2f &nAa, v2f &nBa, v2f &nAb, v2f &nBb, const v2f (&tp)[8], const v2f (&y)[8])
{
#define SA_TAP(ACCA, ACCB, T, Y)                                                               \
    "v_pk_fma_f32 %[" ACCA "], %[" T "], %[" Y "], %[" ACCA "] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t" \
    "v_pk_fma_f32 %[" ACCB "], %[" T "], %[" Y "], %[" ACCB "] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
    asm(SA_TAP("a1", "a2", "t0", "y0") SA_TAP("b1", "b2", "t1", "y1") SA_TAP("a1", "a2", "t2", "y2")
            SA_TAP("b1", "b2", "t3", "y3") SA_TAP("a1", "a2", "t4", "y4") SA_TAP("b1", "b2", "t5", "y5")
                SA_TAP("a1", "a2", "t6", "y6") SA_TAP("b1", "b2", "t7", "y7") ""
        : [a1] "+v"(nAa), [a2] "+v"(nBa), [b1] "+v"(nAb), [b2] "+v"(nBb)
        : [t0] "s"(tp[0]), [t1] "s"(tp[1]), [t2] "s"(tp[2]), [t3] "s"(tp[3]), [t4] "s"(tp[4]), [t5] "s"(tp[5]), [t6] "s"(tp[6]),
          [t7] "s"(tp[7]), [y0] "v"(y[0]), [y1] "v"(y[1]), [y2] "v"(y[2]), [y3] "v"(y[3]), [y4] "v"(y[4]), [y5] "v"(y[5]),
          [y6] "v"(y[6]), [y7] "v"(y[7]));
#undef SA_TAP
}

// The predictor of one section over the thread's 32 fresh pairs y = (chunk A, chunk B): chunk end states from zero state,
//   z = A^16 (sum_{j<16} m[j] y[j]) + sum_{j<16} m[j] y[16 + j]     (SaIirSecK::mnext; block Horner over two half chunks:
// the same sixteen tap pairs serve both halves and stay in their scalar registers).
__device__ __forceinline__ void predict_chunk_ends(const v2f (&tp)[16], const v2f q0, const v2f q1, const v2f (&d)[32],
                                                   v2f &zA, v2f &zB)
{
    // four accumulators: each chain sees a dependent FMA every fourth instruction
    v2f nAa = {0.f, 0.f}, nBa = {0.f, 0.f}, nAb = {0.f, 0.f}, nBb = {0.f, 0.f};
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int j = 0; j < 16; j += 8) {
            const v2f t8[8] = {tp[j], tp[j + 1], tp[j + 2], tp[j + 3], tp[j + 4], tp[j + 5], tp[j + 6], tp[j + 7]};
            const int o = 16 * hh + j;
            const v2f y8[8] = {d[o], d[o + 1], d[o + 2], d[o + 3], d[o + 4], d[o + 5], d[o + 6], d[o + 7]};
            tap_fma8(nAa, nBa, nAb, nBb, t8, y8);
        }
        if (hh == 0) {
            nAa = mv0_s(q0, q1, nAa + nAb);
            nBa = mv0_s(q0, q1, nBa + nBb);
            nAb = v2f{0.f, 0.f};
            nBb = v2f{0.f, 0.f};
        }
    }
    zA = nAa + nAb;
    zB = nBa + nBb;
}

// The wave-uniform constants of one section, read one section ahead (while the previous section's loops run)
// so that their scalar-load latency is not on the path between two sections.
struct SecConsts {
    v2f pc0, pc1, mb0, mb1;            // columns of Pc and of T^-1
    float b0, b1, b2, a1, a2, flags_bits;     // flags travel as raw bits
};
template <typename SecT>
__device__ __forceinline__ SecConsts load_consts(const SecT &k)
{
    return {v2f{k.pc[0], k.pc[1]}, v2f{k.pc[2], k.pc[3]}, v2f{k.mback[0], k.mback[1]}, v2f{k.mback[2], k.mback[3]},
            k.c[0], k.c[1], k.c[2], k.c[3], k.c[4], __builtin_bit_cast(float, k.flags)};
}
__device__ __forceinline__ void pin_consts(const SecConsts &c)
{
    asm volatile("" ::"s"(c.pc0), "s"(c.pc1), "s"(c.mb0), "s"(c.mb1), "s"(c.b1), "s"(c.a1), "s"(c.a2), "s"(c.flags_bits));
}

// One cascade section, in place on the thread's two chunks.
//   zA, zB (in) : predicted end states (z1, z2) of chunk A and chunk B from zero state, pole coordinates
//   zA, zB (out): the same for the NEXT section
//   c  (in)    : this section's constants;   cn (out): the next section's, requested here
// Two loops: the recursion (3 scalar constants), then the next section's predictor over the fresh outputs (its
// 16 tap pairs and half-chunk matrix, requested before the recursion so that they arrive under it and resident in
// 36 scalar registers until the predictor is done: predict_chunk_ends).
// (-DSA_STAMP_IIR, diagnostic builds: stamps 3..8 mark the inside of section 2 instead of the FFT passes)
#ifdef SA_STAMP_IIR
#define SA_STAMP_SEC(i) do { if constexpr (SIDX == 2) SA_STAMP(i); } while (0)
#define SA_STAMP_FFT(i) do {} while (0)
#else
#define SA_STAMP_SEC(i) do {} while (0)
#define SA_STAMP_FFT(i) SA_STAMP(i)
#endif
template <bool PREDICT_NEXT, bool UNIT, int SIDX, typename SecT>
__device__ __forceinline__ void iir_section(v2f (&d)[32], const SecT &k, const SecT &knext, const SecConsts c,
                                            SecConsts &cn, const float4 *lanep_lds, float2 *scr_s, int lane, int wave,
                                            v2f &zA, v2f &zB)
{
    SA_STAMP_SEC(3);
    // state after both chunks of this thread, from zero state: T = Pc zA + zB
    v2f T = mv_s(c.pc0, c.pc1, zA, zB);
    // inclusive affine scan inside the 16-lane row; levels whose transition power has decayed below
    // float resolution are skipped (wave-uniform flags from the host)
    const int flags = __builtin_bit_cast(int, c.flags_bits);
    if (!(flags & 1)) scan_level<1>(T, k.plev[0]);
    if (!(flags & 2)) scan_level<2>(T, k.plev[1]);
    if (!(flags & 4)) scan_level<4>(T, k.plev[2]);
    if (!(flags & 8)) scan_level<8>(T, k.plev[3]);
    const int row = 4 * wave + (lane >> 4);
    if ((lane & 15) == 15) scr_s[row] = make_float2(T.x, T.y);
    const v2f e = {row_shr<1>(T.x), row_shr<1>(T.y)};            // exclusive: state before this thread, row-local
    SA_STAMP_SEC(4);
    lds_barrier();
    SA_STAMP_SEC(5);
    v2f cst;
    if (flags & SA_IIR_SKIP_ROWSCAN) {
        // a row (1024 samples) outlasts the section's memory: the row starts from the previous row's total
        const float2 tt = scr_s[(row - 1) & 15];
        cst = v2f{tt.x, tt.y};
    } else {
        // scan over the 16 row totals (every row of every wave repeats it: 16 lanes, 4 DPP levels)
        const float2 tt = scr_s[lane & 15];
        v2f r = {tt.x, tt.y};
        scan_level<1>(r, k.prow[0]);
        scan_level<2>(r, k.prow[1]);
        scan_level<4>(r, k.prow[2]);
        scan_level<8>(r, k.prow[3]);
        // state at the start of this lane's row = inclusive result of the previous row
        const int src = (lane & 48) | ((row - 1) & 15);
        cst = v2f{lane_get(r.x, src), lane_get(r.y, src)};
    }
    if (row == 0) cst = v2f{0.f, 0.f};
    // start state of chunk A: row-local part + P2^i * (row start state); chunk B: Pc sA + zA
    const float4 lanep = *lanep_lds;                     // read behind this section's barrier (the copy of section 0 is then visible)
    const v2f aS = mv_v(v2f{lanep.x, lanep.y}, v2f{lanep.z, lanep.w}, cst, e);
    const v2f bS = mv_s(c.pc0, c.pc1, aS, zA);
    // pole coordinates -> DF2T states of the recursion (sa_common.hpp), re-paired as (chunk A, chunk B)
    const v2f q1 = {aS.x, bS.x}, q2 = {aS.y, bS.y};
    v2f s1 = c.mb0.x * q1 + c.mb1.x * q2, s2 = c.mb0.y * q1 + c.mb1.y * q2;
    // the next section's tap pairs: requested now, consumed after the recursion
    v2f tp[16], h0 = {0.f, 0.f}, h1 = {0.f, 0.f};
    if constexpr (PREDICT_NEXT) {
#pragma unroll
        for (int j = 0; j < 16; ++j) tp[j] = v2f{k.mnext[j][0], k.mnext[j][1]};
        h0 = v2f{k.p16next[0], k.p16next[1]};
        h1 = v2f{k.p16next[2], k.p16next[3]};
    }
    const float b0 = c.b0, b1 = c.b1, b2 = c.b2, na1 = -c.a1, na2 = -c.a2;
    SA_STAMP_SEC(6);
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const v2f x = d[j];
        v2f y;
        if constexpr (UNIT) {                 // b = [1, r1, 1]: the cascade gain sits in the window table
            y = x + s1;
            s1 = na1 * y + (b1 * x + s2);
            s2 = na2 * y + x;
        } else {
            y = b0 * x + s1;
            s1 = na1 * y + (b1 * x + s2);
            s2 = na2 * y + b2 * x;
        }
        d[j] = y;
    }
    SA_STAMP_SEC(7);
    if constexpr (PREDICT_NEXT) {
        cn = load_consts(knext);
        predict_chunk_ends(tp, h0, h1, d, zA, zB);
        pin_consts(cn);
    }
    SA_STAMP_SEC(8);
}

// All NSEC sections run unconditionally (the host pads shorter cascades with identity sections,
// which are exact: y = 1*x + 0).  A run-time section count would carry the 64 data registers
// through control-flow merges and cost ~190 register copies.
template <int S, int NSEC, bool UNIT, typename PlanT>
__device__ __forceinline__ void iir_sections(v2f (&d)[32], const PlanT &ka, const SaIirLaneTab *__restrict__ lt,
                                             float2 *scr, int lane, int wave, v2f &zA, v2f &zB, const SecConsts c)
{
    if constexpr (S < NSEC) {
        // the per-lane matrices sit in LDS (iir_cascade copies them once): a 64-bit global address per thread held through
        // the whole cascade was among the values the tightest variants spilled
        const float4 *lanep = reinterpret_cast<const float4 *>(reinterpret_cast<const unsigned char *>(scr) + (kLaneOff - kScrOff)) +
                              16 * S + (lane & 15);
        SecConsts cn = c;
        iir_section<(S + 1 < NSEC), UNIT, S>(d, ka.sec[S], ka.sec[S + 1 < NSEC ? S + 1 : S], c, cn, lanep, scr + 16 * S, lane,
                                          wave, zA, zB);
        iir_sections<S + 1, NSEC, UNIT>(d, ka, lt, scr, lane, wave, zA, zB, cn);
    }
}

template <int NSEC, bool UNIT, typename PlanT>
__device__ __forceinline__ void iir_cascade(v2f (&d)[32], const PlanT &ka, const SaIirLaneTab *__restrict__ lt,
                                            float2 *scr, int t)
{
    // the per-lane matrices P2^i of all sections into LDS (96 x 16 bytes; read behind each section's scan barrier)
    if (t < 16 * NSEC)
        reinterpret_cast<float4 *>(reinterpret_cast<unsigned char *>(scr) + (kLaneOff - kScrOff))[t] =
            *reinterpret_cast<const float4 *>(&lt->p[t >> 4][t & 15][0]);
    // predictor for the first section (later ones run after the previous section's recursion)
    const SecConsts c0 = load_consts(ka.sec[0]);
    v2f tp[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) tp[j] = v2f{ka.m0[j][0], ka.m0[j][1]};
    v2f zA, zB;
    predict_chunk_ends(tp, v2f{ka.p16_0[0], ka.p16_0[1]}, v2f{ka.p16_0[2], ka.p16_0[3]}, d, zA, zB);
    pin_consts(c0);
    iir_sections<0, NSEC, UNIT>(d, ka, lt, scr, t & 63, t >> 6, zA, zB, c0);
}

// Half-spectrum outputs (SA_OUT_MAG_HALF, SA_OUT_SPEC_HALF: the numpy.fft.rfft layout, rows of 8193 elements).  A row is
// 4- or 8-byte aligned only (every other row of complex values starts 8 bytes off a 16-byte boundary), so 16-byte stores
// straight from the registers are not possible and element-wise stores put two half-written lines into every wave
// instruction: round 2 measured twice the L2 write requests and +46 % HBM write bytes against the full-magnitude output.
// Here the 2048 consecutive bins a round produces per segment go through LDS in natural order and leave as 16-byte stores
// at ABSOLUTE 16-byte boundaries, 1 KiB contiguous per wave instruction; only the (at most n - 1) elements in front of the
// first and behind the last boundary of the segment are stored one by one.
//   seg: the segment's 2048 elements in LDS;  g: where its first element goes in the output row
template <typename Et>
__device__ __forceinline__ void stream_out_segment(const Et *__restrict__ seg, Et *__restrict__ g, int t)
{
    constexpr int n = 16 / (int)sizeof(Et);                         // elements per 16-byte unit
    const int h = (int)(((size_t)g / sizeof(Et)) & (size_t)(n - 1)); // unit u holds elements n u - h .. n u - h + n - 1
    constexpr int per_thread = 2048 / n / kThreads;
#pragma unroll
    for (int i = 0; i <= per_thread; ++i) {
        const int u = t + kThreads * i;
        if (i == per_thread && (h == 0 || t != 0)) break;           // the unit behind the last full one: thread 0, if any
        const int e0 = n * u - h;
        if (e0 >= 0 && e0 + n <= 2048) {
            float v[4];
            if constexpr (n == 2) {
                const float2 a = reinterpret_cast<const float2 *>(seg)[e0], b = reinterpret_cast<const float2 *>(seg)[e0 + 1];
                v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = reinterpret_cast<const float *>(seg)[e0 + j];
            }
            store_nt(reinterpret_cast<float *>(g + e0), v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int j = 0; j < n; ++j)
                if (e0 + j >= 0 && e0 + j < 2048) {
                    if constexpr (n == 2) {
                        const float2 a = reinterpret_cast<const float2 *>(seg)[e0 + j];
                        store_nt(reinterpret_cast<float2 *>(g + e0 + j), a.x, a.y);
                    } else {
                        store_nt(reinterpret_cast<float *>(g + e0 + j), reinterpret_cast<const float *>(seg)[e0 + j]);
                    }
                }
        }
    }
}

// Position of Z[k] inside the half image of the natural-order exchange.  Round r holds the rows
// d = k >> 9 of {0..3, 12..15} (r = 0) or {4..11} (r = 1), compacted to d' = (d + 4r) & 7; inside a row
// the 512 entries are padded by one per 32.  With q = k - 2048 r for the low member of a pair and
// w = 2048 - q for its partner 8192 - k, the compacted row is the same expression in both rounds:
//   low  member: row = q >> 9            partner: row = (4 + (w >> 9)) & 7
__device__ __forceinline__ int zrow_pos(int within, int row)
{
    const int rest = within & 511;
    return rest + (rest >> 5) + 528 * row;
}
__device__ __forceinline__ int zpos_low(int q) { return zrow_pos(q, q >> 9); }
__device__ __forceinline__ int zpos_partner(int w) { return zrow_pos(w, (4 + (w >> 9)) & 7); }

// ---------------------------------------------------------------------------------------------
// One frame: window -> IIR -> FFT -> split -> store.
// ONE_ROUND (bypassed chain on float32 frames, small batches only: sa_launch_chain_f32): the whole 64 KiB frame is
// requested at once into a 64 KiB LDS image instead of two half-frame rounds -- one HBM round trip and one barrier
// fewer per frame, at two workgroups per CU instead of four, which costs nothing while the batch leaves the CUs
// half empty anyway (B <= 512: at most two workgroups per CU either way).
template <int NSEC, bool UNIT, int OUT, bool WINGEN, bool ONE_ROUND, typename PlanT>
__device__ __forceinline__ void chain_frame(const sa_in_t *__restrict__ in, SA_IN_SCALE_PARAM void *__restrict__ out,
                                            const int f, unsigned char *smem,
                                            const float4 *__restrict__ winb, const float4 *__restrict__ twT,
                                            const float4 *__restrict__ twB, const float2 *__restrict__ twC,
                                            const SaIirLaneTab *__restrict__ lanetab, const PlanT &ka)
{
    cf *ldc = reinterpret_cast<cf *>(smem);
    float2 *scr = reinterpret_cast<float2 *>(smem + kScrOff);
    cf *side = reinterpret_cast<cf *>(smem + kSideOff);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int lo = lane & 15;          // b in pass B, c in pass C
    const int kq = lane >> 4;
    const sa_in_t *xin = in + (size_t)f * SA_NPTS;
    constexpr bool IIR = NSEC > 0;
    cf a[32];
#ifdef SA_STAMPS
    if (threadIdx.x == 0 && g_sa_stamps) g_sa_stamps[(size_t)f * 16 + 13] = __builtin_amdgcn_s_memrealtime();
#endif
    SA_STAMP(0);

    if constexpr (IIR) {
        v2f d[32];
#if SA_F32_INPUT_I16
        stage_in_chunks<WINGEN>(xin, in_scale, reinterpret_cast<const float4 *>(lanetab->win_t), lanetab, smem, t, d);
#else
        stage_in_chunks<WINGEN>(xin, reinterpret_cast<const float4 *>(lanetab->win_t), lanetab, smem, t, d);
#endif
        SA_STAMP(1);
        iir_cascade<NSEC, UNIT>(d, ka, lanetab, scr, t);
        SA_STAMP(2);
        // exchange to the pass-A layout in two rounds (m1 < 16, m1 >= 16): the owners of the half
        // write z[32 t' + j] = (x[2j], x[2j+1]) at 33 t' + j; everybody reads z[256 m1 + t].  Real and
        // imaginary part (even / odd sample) sit in different register pairs of d[], so each complex value is
        // stored as two dwords at adjacent addresses (one ds_write2_b32, no register copies); the reader then
        // gets an aligned (re, im) pair per 8-byte read: half the LDS read instructions of two float planes.
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            lds_barrier();
            if ((t >> 7) == h) {
                // written out: left to itself the compiler merges the two dword stores into one 64-bit store and
                // copies the two halves into a register pair first (128 v_mov per thread)
                const unsigned zw = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)(smem) +
                                    8u * 33u * (unsigned)(t & 127);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(zw), "v"(d[2 * j].x), "v"(d[2 * j + 1].x),
                                 "i"(2 * j), "i"(2 * j + 1) : "memory");
                    asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(zw), "v"(d[2 * j].y), "v"(d[2 * j + 1].y),
                                 "i"(2 * (16 + j)), "i"(2 * (16 + j) + 1) : "memory");
                }
            }
            lds_barrier();
#pragma unroll
            for (int m = 0; m < 16; ++m) a[safft::brev(16 * h + m, 5)] = ldc[264 * m + 33 * (t >> 5) + (t & 31)];
        }
    } else {
        // No IIR: the frame goes HBM -> LDS in natural order (two rounds of 32 KiB, LDS-DMA), and the
        // thread picks z[256 m1 + t] straight out of the image; the window comes as 16-byte loads of
        // the pass-A layout (winb[p][t] = window at samples 512(2p)+2t, +1, 512(2p+1)+2t, +1).
        // 8 + 8 vector-memory instructions per wave and round instead of 32 8-byte loads.
#if SA_F32_INPUT_I16
        {
            // int16 samples: the whole frame (32 KiB) in one round; z[256 m1 + t] = (x[2 i], x[2 i + 1]) is one dword
            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int n = wave * 8 + i;
                const int16_t *src = xin + n * 512 + lane * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(smem + n * 1024), 16, 0, SA_DMA_AUX);
            }
            __builtin_amdgcn_s_setprio(0);
            __syncthreads();
            const unsigned *ldu = reinterpret_cast<const unsigned *>(smem);
#pragma unroll
            for (int pp = 0; pp < 16; ++pp) {
                const float4 w = winb[pp * 256 + t];
                const unsigned u0 = ldu[256 * (2 * pp) + t], u1 = ldu[256 * (2 * pp + 1) + t];
                const cf z0 = cf{(float)(int)(short)(u0 & 0xFFFFu), (float)((int)u0 >> 16)} * cf{in_scale, in_scale};
                const cf z1 = cf{(float)(int)(short)(u1 & 0xFFFFu), (float)((int)u1 >> 16)} * cf{in_scale, in_scale};
                a[safft::brev(2 * pp, 5)] = {z0.x * w.x, z0.y * w.y};
                a[safft::brev(2 * pp + 1, 5)] = {z1.x * w.z, z1.y * w.w};
            }
        }
#else
        constexpr int kRounds = ONE_ROUND ? 1 : 2;
        constexpr int kSlabs = 32 / kRounds / 2;           // 1 KiB DMA requests per wave and round: 8 (two rounds) or 16
        constexpr int kPairs = 16 / kRounds;               // window quads (two complex points each) per thread and round
#pragma unroll
        for (int h = 0; h < kRounds; ++h) {
            if (h == 1) __syncthreads();
            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int i = 0; i < kSlabs; ++i) {
                const int n = wave * kSlabs + i;
                const float *src = xin + h * 8192 + n * 256 + lane * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(smem + n * 1024), 16, 0, SA_DMA_AUX);
            }
            __builtin_amdgcn_s_setprio(0);
            __syncthreads();          // (the round's window values requested in front of this barrier: 2 % slower, round 4)
#pragma unroll
            for (int pp = 0; pp < kPairs; ++pp) {
                const float4 w = winb[(kPairs * h + pp) * 256 + t];
                const cf z0 = ldc[256 * (2 * pp) + t];
                const cf z1 = ldc[256 * (2 * pp + 1) + t];
                a[safft::brev(2 * kPairs * h + 2 * pp, 5)] = {z0.x * w.x, z0.y * w.y};
                a[safft::brev(2 * kPairs * h + 2 * pp + 1, 5)] = {z1.x * w.z, z1.y * w.w};
            }
        }
#endif
    }

    // ---- pass A: 32-point FFT over m1 (stride 256), then twiddle W_8192^(k1*m2), m2 = t
    SA_STAMP_FFT(3);
    // the thread's twiddle anchors (requested before the butterflies, consumed after them): W^(b t) for
    // b = 1..7 and W^(8 a t) for a = 1..3 with W = W_8192, plus W_16384^(4 t) for the split step.  The 31
    // factors W^(k1 t), k1 = 8a + b, are applied as two complex products per point; the 64 KiB table of all
    // of them (one 16-byte load per two points, every frame, through L2 -> L1) is what this replaces:
    // 24 KiB of anchors per frame, and the loads no longer sit between the butterflies and the exchange.
    float4 an[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) an[i] = twT[i * 256 + t];
    // The split-step anchors W_16384^(4 t), W_16384^(4 (t + 1)) ride along for the full-spectrum output (round 4: requested
    // right before the split step their L2 round trip was exposed once per frame -- 103.8 -> 97.0 us on the bypassed chain
    // at B = 4096, 11.5 -> 10.8 us at B = 256, -1 % with the cascade, gpurun_out/ab_an5.txt).  The half-spectrum variants
    // keep the late request: four more registers through three FFT passes make them spill.
    float4 an5_early = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (OUT == SA_OUT_MAG_FULL) an5_early = twT[5 * 256 + t];
    safft::fft_dit<32>(a);
    {
        const cf wb[8] = {{1.f, 0.f}, {an[0].x, an[0].y}, {an[0].z, an[0].w}, {an[1].x, an[1].y},
                          {an[1].z, an[1].w}, {an[2].x, an[2].y}, {an[2].z, an[2].w}, {an[3].x, an[3].y}};
        const cf wa[4] = {{1.f, 0.f}, {an[3].z, an[3].w}, {an[4].x, an[4].y}, {an[4].z, an[4].w}};
#pragma unroll
        for (int k1 = 1; k1 < 32; ++k1) {
            if ((k1 & 7) != 0) a[k1] = safft::cmul(a[k1], wb[k1 & 7]);
            if ((k1 >> 3) != 0) a[k1] = safft::cmul(a[k1], wa[k1 >> 3]);
        }
    }
    SA_STAMP_FFT(4);
    // ---- exchange A -> B in two rounds of 16 rows; FFT q of a thread lives in round q:
    //      k1 = 16q + 4 wave + kq, b = lo; inputs ldc[row][16 a + b] with row pitch 272
    cf p[2][16];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        lds_barrier();                                   // previous image fully consumed
#pragma unroll
        for (int r = 0; r < 16; ++r) ldc[r * 272 + t] = a[16 * q + r];
        lds_barrier();
        const int row = 4 * wave + kq;
#pragma unroll
        for (int aa = 0; aa < 16; ++aa) p[q][safft::brev(aa, 4)] = ldc[row * 272 + 16 * aa + lo];
    }
    SA_STAMP_FFT(5);
    // ---- pass B: 16-point FFT over a, twiddle W_256^(b*c)
    safft::fft_dit<16>(p[0]);
    safft::fft_dit<16>(p[1]);
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) {                       // twB4[pp][b] = (W_256^(2pp * b), W_256^((2pp+1) * b))
        const float4 w = twB[pp * 16 + lo];
        if (pp > 0) {
            p[0][2 * pp] = safft::cmul(p[0][2 * pp], {w.x, w.y});
            p[1][2 * pp] = safft::cmul(p[1][2 * pp], {w.x, w.y});
        }
        p[0][2 * pp + 1] = safft::cmul(p[0][2 * pp + 1], {w.z, w.w});
        p[1][2 * pp + 1] = safft::cmul(p[1][2 * pp + 1], {w.z, w.w});
    }
    SA_STAMP_FFT(6);
    // ---- exchange B -> C: a 16x16 transpose inside each 16-lane group, through the row this group
    //      just read (pitch 17).  Only these 16 lanes touch the row: no workgroup barrier; the LDS
    //      executes a wave's accesses in order.
    {
        const int base = (4 * wave + kq) * 272;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int c = 0; c < 16; ++c) ldc[base + c * 17 + lo] = p[q][c];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int b = 0; b < 16; ++b) p[q][safft::brev(b, 4)] = ldc[base + lo * 17 + b];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    SA_STAMP_FFT(7);
    // ---- pass C: 16-point FFT over b -> d;  Z[k1 + 32c + 512d], k1 = 16q + 4 wave + kq, c = lo
    safft::fft_dit<16>(p[0]);
    safft::fft_dit<16>(p[1]);
    SA_STAMP_FFT(8);
    // split-step anchors: W_16384^(4 t) and the right-hand neighbour's W_16384^(4 (t + 1)), (1, 0) for t = 255 (its
    // neighbour is thread 0 of the next block of 1024 bins, whose anchor is W^0).  Half-spectrum outputs request them
    // here, through an opaque copy of the thread index (see an5_early above).
    int ts = t;
    asm volatile("" : "+v"(ts));
    float4 an5 = an5_early;
    if constexpr (OUT != SA_OUT_MAG_FULL) an5 = twT[5 * 256 + ts];
    const cf wP = {an5.x, an5.y}, wPn = {an5.z, an5.w};
    // ---- natural-order image + split step, two rounds: round 0 = d in {0..3,12..15} (bins k < 2048
    //      and their partners), round 1 = d in {4..11}.  Z[2048] and Z[6144] sit on the seam and
    //      travel through two side slots.
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        lds_barrier();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int k1 = 16 * q + 4 * wave + kq;
#pragma unroll
            for (int dd = 0; dd < 8; ++dd) {
                const int dsel = (r == 0) ? (dd < 4 ? dd : dd + 8) : dd + 4;
                ldc[k1 + 33 * lo + 528 * dd] = p[q][dsel];
            }
        }
        if (r == 0 && t == 0) {                               // k1 = 0, c = 0: d = 12 and d = 4
            side[0] = p[0][12];
            side[1] = p[0][4];
        }
        lds_barrier();
        SA_STAMP(9 + r);
        constexpr bool HALF = OUT == SA_OUT_MAG_HALF || OUT == SA_OUT_SPEC_HALF;
        cf Rs[2][5], Is[2][5];                                 // half-spectrum outputs: both groups wait for the staging pass
        float mps[2][5], mqs[2][5];                            // (magnitudes only for SA_OUT_MAG_HALF: half the registers)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            // (half-spectrum outputs keep both groups' results until the staging pass: the groups must not be interleaved)
            if (HALF && jj == 1) __builtin_amdgcn_sched_barrier(0);
            const int q0 = 4 * (t + 256 * jj);                 // k0 - 2048 r: bins q0 .. q0+4 of this round
            const int k0 = q0 + 2048 * r;
            // W_16384^(k0 + e) = W^(4 t) * W^(2048 r + 1024 jj + e): the second factor is the same for every
            // thread (twC, scalar loads), the first is the thread's anchor -- no per-bin table
            cf w[5];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = cmul_s(wP, twC[(2 * r + jj) * 5 + e]);
            // bin k0 + 4 is bin 0 of the neighbouring group, which also stores it: the mirrored halves of the
            // spectrum stay bit-identical only if both evaluate the same product, so this is the NEIGHBOUR's
            // twiddle for its e = 0, anchor(t + 1) * C[block][0], with the block advancing at t = 255
            {
                const float2 c0 = twC[(2 * r + jj) * 5], c1 = twC[(2 * r + jj + 1) * 5];
                const cf csel = (t == 255) ? cf{c1.x, c1.y} : cf{c0.x, c0.y};
                w[4] = safft::cmul(wPn, csel);
            }
            // a group of four bins never straddles a padding or row boundary, so four positions serve
            // the ten reads: low members q0+e at pa+e (e<4) and pb; partners at pm0, pm4+3, pm4+2, pm4+1, pm4
            const int pa = zpos_low(q0), pb = zpos_low(q0 + 4);
            const int pm0 = zpos_partner(2048 - q0), pm4 = zpos_partner(2044 - q0);
            cf zk[5] = {ldc[pa], ldc[pa + 1], ldc[pa + 2], ldc[pa + 3], ldc[pb]};
            cf zm[5] = {ldc[pm0], ldc[pm4 + 3], ldc[pm4 + 2], ldc[pm4 + 1], ldc[pm4]};
            // the seam pair (2048, 6144): Z[2048] is not in round 0's image, Z[6144] not in round 1's
            if (r == 0 && q0 == 2044) {
                zk[4] = side[1];
                zm[4] = side[0];
            }
            if (r == 1 && q0 == 0) zm[0] = side[0];
            cf R[5], I[5];
#pragma unroll
            for (int e = 0; e < 5; ++e) split_eval(zk[e], zm[e], w[e], R[e], I[e]);
            if constexpr (!HALF) {
                split_store<OUT>(R, I, out, f, k0);
            } else if constexpr (OUT == SA_OUT_SPEC_HALF) {
#pragma unroll
                for (int e = 0; e < 5; ++e) {
                    Rs[jj][e] = R[e];
                    Is[jj][e] = I[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 5; ++e) {
                    const cf m2 = safft::pk_fma(I[e], I[e], R[e] * R[e]);          // (|P|^2, |Q|^2)
                    mps[jj][e] = fast_sqrt(m2.x);
                    mqs[jj][e] = fast_sqrt(m2.y);
                }
            }
        }
        if constexpr (HALF) {
            // stage the round's two runs of 2048 bins in natural order (stream_out_segment):
            //   segment 0 = bins 2048 r .. 2048 r + 2047           P_e = X[k0 + e], e < 4, at q0 + e
            //   segment 1 = bins 6144 - 2048 r .. 8191 - 2048 r    conj Q_e = X[8192 - k0 - e], e = 1..4, at 2048 - q0 - e
            lds_barrier();                                     // every thread is done with the Z image
            if constexpr (OUT == SA_OUT_SPEC_HALF) {
                float4 *s0 = reinterpret_cast<float4 *>(smem), *s1 = reinterpret_cast<float4 *>(smem + 2048 * 8);
                float2 *orow = reinterpret_cast<float2 *>(out) + (size_t)f * (SA_MC + 1);
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int q0 = 4 * (t + 256 * jj);
                    const cf(&R)[5] = Rs[jj];
                    const cf(&I)[5] = Is[jj];
                    s0[q0 / 2] = make_float4(R[0].x, I[0].x, R[1].x, I[1].x);
                    s0[q0 / 2 + 1] = make_float4(R[2].x, I[2].x, R[3].x, I[3].x);
                    s1[(2044 - q0) / 2] = make_float4(R[4].y, -I[4].y, R[3].y, -I[3].y);
                    s1[(2044 - q0) / 2 + 1] = make_float4(R[2].y, -I[2].y, R[1].y, -I[1].y);
                    if (r == 0 && q0 == 0) store_nt(orow + SA_MC, R[0].y, -I[0].y);          // X[8192]
                }
                lds_barrier();
                stream_out_segment(reinterpret_cast<const float2 *>(smem), orow + 2048 * r, t);
                stream_out_segment(reinterpret_cast<const float2 *>(smem + 2048 * 8), orow + 6144 - 2048 * r, t);
            } else {
                float4 *s0 = reinterpret_cast<float4 *>(smem), *s1 = reinterpret_cast<float4 *>(smem + 2048 * 4);
                float *orow = reinterpret_cast<float *>(out) + (size_t)f * (SA_MC + 1);
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int q0 = 4 * (t + 256 * jj);
                    const float(&mp)[5] = mps[jj];
                    const float(&mq)[5] = mqs[jj];
                    s0[q0 / 4] = make_float4(mp[0], mp[1], mp[2], mp[3]);
                    s1[(2044 - q0) / 4] = make_float4(mq[4], mq[3], mq[2], mq[1]);
                    if (r == 0 && q0 == 0) store_nt(orow + SA_MC, mq[0]);
                }
                lds_barrier();
                stream_out_segment(reinterpret_cast<const float *>(smem), orow + 2048 * r, t);
                stream_out_segment(reinterpret_cast<const float *>(smem + 2048 * 4), orow + 6144 - 2048 * r, t);
            }
        }
    }
    SA_STAMP(11);
#ifdef SA_STAMPS
    __builtin_amdgcn_s_waitcnt(0);      // drain the stores so the last stamp sees them retire
#endif
    SA_STAMP(12);
#ifdef SA_STAMPS
    if (threadIdx.x == 0 && g_sa_stamps) {             // placement and wall-clock end (100 MHz counter, the same on every XCD)
        g_sa_stamps[(size_t)f * 16 + 14] = __builtin_amdgcn_s_memrealtime();
        g_sa_stamps[(size_t)f * 16 + 15] =
            (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |
            ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);
    }
#endif
}

template <int NSEC, bool UNIT, int OUT, bool WINGEN, bool ONE_ROUND = false>
__global__ __launch_bounds__(kThreads, 4) void chain_f32_kernel(const sa_in_t *__restrict__ in, SA_IN_SCALE_PARAM
                                                                 void *__restrict__ out, int batch,
                                                                 const float4 *__restrict__ winb,
                                                                 const float4 *__restrict__ twT,
                                                                 const float4 *__restrict__ twB,
                                                                 const float2 *__restrict__ twC,
                                                                 const SaIirLaneTab *__restrict__ lanetab,
                                                                 const SaIirK ka)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int f = blockIdx.x;
    if (f >= batch) return;
    chain_frame<NSEC, UNIT, OUT, WINGEN, ONE_ROUND>(in, SA_IN_SCALE_ARG out, f, smem, winb, twT, twB, twC, lanetab, ka);
}

// Window (+ IIR) only: the FFT input time series (debug / parity output, not a hot path: two workgroups per CU are
// asked for, so the register allocator has 256 registers and spills nothing in any instantiation).
template <int NSEC, bool UNIT>
__global__ __launch_bounds__(kThreads, 2) void time_f32_kernel(const sa_in_t *__restrict__ in, SA_IN_SCALE_PARAM
                                                                float *__restrict__ out, int batch,
                                                                const float4 *__restrict__ wint_plain,
                                                                const SaIirLaneTab *__restrict__ lanetab,
                                                                const SaIirK ka)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *scr = reinterpret_cast<float2 *>(smem + kScrOff);
    const int t = threadIdx.x;
    const int f = blockIdx.x;
    if (f >= batch) return;
    v2f d[32];
    const float4 *wint = NSEC > 0 ? reinterpret_cast<const float4 *>(lanetab->win_t) : wint_plain;
#if SA_F32_INPUT_I16
    stage_in_chunks<false>(in + (size_t)f * SA_NPTS, in_scale, wint, lanetab, smem, t, d);
#else
    stage_in_chunks<false>(in + (size_t)f * SA_NPTS, wint, lanetab, smem, t, d);
#endif
    if constexpr (NSEC > 0) iir_cascade<NSEC, UNIT>(d, ka, lanetab, scr, t);
    // Stage-out, the stage-in run backwards: each thread owns 64 consecutive samples, so storing straight
    // from the registers puts every lane of a store instruction into another 256-byte block (measured 5x
    // slower than the whole spectrum chain).  Round h: the thread writes its chunk h into its 128-byte LDS row
    // (same XOR swizzle as the stage-in), then every wave instruction picks up 1 KiB of LDS in linear order,
    // i.e. eight 128-byte row segments, and stores them with 16 B per lane.
    float4 *lds4 = reinterpret_cast<float4 *>(smem);
    float *o = out + (size_t)f * SA_NPTS;
    const int lane = t & 63, wave = t >> 6, rl = lane >> 3;
    const int sw = (t >> 1) & 7;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        lds_barrier();                       // scan scratch / previous round consumed
#pragma unroll
        for (int g = 0; g < 8; ++g) {        // undo the folded 1/2 (exact)
            const float4 v = h == 0 ? make_float4(2.f * d[4 * g].x, 2.f * d[4 * g + 1].x, 2.f * d[4 * g + 2].x, 2.f * d[4 * g + 3].x)
                                    : make_float4(2.f * d[4 * g].y, 2.f * d[4 * g + 1].y, 2.f * d[4 * g + 2].y, 2.f * d[4 * g + 3].y);
            lds4[t * 8 + (g ^ sw)] = v;
        }
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = wave * 8 + i;                        // slab: rows 8n .. 8n+7
            const int r = 8 * n + rl;
            const int lc = (lane & 7) ^ ((r >> 1) & 7);
            const float4 v = lds4[n * 64 + lane];
            store_nt(o + r * 64 + h * 32 + lc * 4, v.x, v.y, v.z, v.w);
        }
    }
}

template <typename K>
hipError_t set_lds(K kernel)
{
    return sa_set_dyn_lds_once(reinterpret_cast<const void *>(kernel), kLdsBytes);
}

}  // namespace

#if defined(SA_STAMPS) && !SA_F32_INPUT_I16
extern "C" int sa_debug_set_stamps(void *p)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_sa_stamps), &p, sizeof(p));
}
#endif

namespace {

template <int NSEC, bool UNIT>
hipError_t launch_nsec(const sa_in_t *in, const float in_scale, void *out, int batch, int out_kind, const SaF32Tables &tb,
                       const SaIirK &ka, hipStream_t stream, SaLaunchEv ev)
{
    const dim3 grid(batch), block(kThreads);
    hipError_t e = hipSuccess;
#if !SA_F32_INPUT_I16
    // Small batches of the bypassed chain (the board's power-on mode, new/command_control.vhd:31; BASELINE config 2 is
    // B = 256): at most two workgroups per CU are resident whatever the kernel asks for, so the frame comes in as ONE
    // 64 KiB round (chain_frame<ONE_ROUND>).  Same arithmetic, same results.
    // A/B in one process (gpurun_out/ab_oneround.txt): 11.0 -> 10.8 us at B = 256, 15.5 -> 15.3 us at B = 512.
    constexpr int kOneRoundMax = 512;
    if constexpr (NSEC == 0) if (batch <= kOneRoundMax && out_kind != SA_OUT_TIME) {
#define SA_LAUNCH1(OUTK)                                                                               \
    do {                                                                                               \
        auto kern = chain_f32_kernel<0, false, OUTK, false, true>;                                     \
        e = sa_set_dyn_lds_once(reinterpret_cast<const void *>(kern), kLdsOneRound);                   \
        if (e != hipSuccess) return e;                                                                 \
        hipExtLaunchKernelGGL(kern, grid, block, kLdsOneRound, stream, ev.start, ev.stop, 0, in, out, batch, tb.win_b, tb.twT, \
                              tb.twB, tb.twC, tb.lanetab, ka);                                         \
    } while (0)
        switch (out_kind) {
            case SA_OUT_MAG_FULL: SA_LAUNCH1(SA_OUT_MAG_FULL); break;
            case SA_OUT_MAG_HALF: SA_LAUNCH1(SA_OUT_MAG_HALF); break;
            case SA_OUT_SPEC_HALF: SA_LAUNCH1(SA_OUT_SPEC_HALF); break;
            default: return hipErrorInvalidValue;
        }
#undef SA_LAUNCH1
        return hipGetLastError();
    }
#endif
#define SA_LAUNCH(OUTK)                                                                                \
    do {                                                                                               \
        auto kern = ka.wingen ? chain_f32_kernel<NSEC, UNIT, OUTK, (NSEC > 0)>                          \
                              : chain_f32_kernel<NSEC, UNIT, OUTK, false>;                             \
        e = set_lds(kern);                                                                             \
        if (e != hipSuccess) return e;                                                                 \
        hipExtLaunchKernelGGL(kern, grid, block, kLdsBytes, stream, ev.start, ev.stop, 0, in, SA_IN_SCALE_ARG out, batch, tb.win_b, tb.twT, \
                           tb.twB, tb.twC, tb.lanetab, ka);                                            \
    } while (0)
    switch (out_kind) {
        case SA_OUT_MAG_FULL: SA_LAUNCH(SA_OUT_MAG_FULL); break;
        case SA_OUT_MAG_HALF: SA_LAUNCH(SA_OUT_MAG_HALF); break;
        case SA_OUT_SPEC_HALF: SA_LAUNCH(SA_OUT_SPEC_HALF); break;
        case SA_OUT_TIME: {
            auto kern = time_f32_kernel<NSEC, UNIT>;
            e = set_lds(kern);
            if (e != hipSuccess) return e;
            hipExtLaunchKernelGGL(kern, grid, block, kLdsBytes, stream, ev.start, ev.stop, 0, in, SA_IN_SCALE_ARG reinterpret_cast<float *>(out), batch,
                               tb.win_t, tb.lanetab, ka);
            break;
        }
        default: return hipErrorInvalidValue;
    }
#undef SA_LAUNCH
    return hipGetLastError();
}

}  // namespace

// tb.iir->nsec is the PADDED section count (0, 2, 4 or 6; see build_plan in specan_abi.cpp).
#if SA_F32_INPUT_I16
hipError_t sa_launch_chain_f32_i16(const int16_t *in, float in_scale, void *out, int batch, int out_kind, const SaF32Tables &tb,
                                   hipStream_t stream, SaLaunchEv ev)
#else
hipError_t sa_launch_chain_f32(const float *in, void *out, int batch, int out_kind, const SaF32Tables &tb,
                               hipStream_t stream, SaLaunchEv ev)
#endif
{
#if !SA_F32_INPUT_I16
    const float in_scale = 1.f;         // the float32 kernels do not take it
#endif
    if (batch <= 0) return hipSuccess;
    static const SaIirK kNoIir = {};
    const int nsec = tb.iir ? tb.iir->nsec : 0;
    const SaIirK &ka = nsec > 0 ? *tb.iir : kNoIir;
    const bool unit = nsec > 0 && ka.unit != 0;
    switch (nsec) {
        case 0: return launch_nsec<0, false>(in, in_scale, out, batch, out_kind, tb, ka, stream, ev);
        case 2: return unit ? launch_nsec<2, true>(in, in_scale, out, batch, out_kind, tb, ka, stream, ev)
                            : launch_nsec<2, false>(in, in_scale, out, batch, out_kind, tb, ka, stream, ev);
        case 4: return unit ? launch_nsec<4, true>(in, in_scale, out, batch, out_kind, tb, ka, stream, ev)
                            : launch_nsec<4, false>(in, in_scale, out, batch, out_kind, tb, ka, stream, ev);
        case 6: return unit ? launch_nsec<6, true>(in, in_scale, out, batch, out_kind, tb, ka, stream, ev)
                            : launch_nsec<6, false>(in, in_scale, out, batch, out_kind, tb, ka, stream, ev);
        default: return hipErrorInvalidValue;
    }
}
