// probes/chain_f32_w8.hip -- the fused float signal path at EIGHT waves per SIMD (gfx950, MI355X):
//     Hann window -> 6-section biquad cascade -> 16384-point real FFT -> magnitude / half spectrum
// (new/hann8192.vhd -> new/filter_iir12_cust.vhd -> ip/xfft_0 of the reference), one 512-thread workgroup per frame,
// <= 64 VGPRs and 39 KiB of LDS so that four workgroups = 32 waves share a CU.
//
// NOT PART OF THE PRODUCT LIBRARY.  Round 3 built this form (the round-2 review's first item), got it bit-for-bit
// through the whole float GPU suite and measured it: 150-160 us per 4096 frames against 131-134 us for the 256-thread
// kernel of chain_f32.hip on the same box (profiles/r3_w8_vs_w4.txt: A/B runs and counters).  Twice the waves of the SAME
// four workgroups per CU add no independent phases to a SIMD -- the waves of a workgroup wait at the same barriers
// together -- while every barrier now gathers eight waves, the exchanges need 66 % more LDS instructions and the
// 64-register budget leaves spills.  It is compiled only into A/B builds (make ab NAME=w8 W8=1 in csrc/), where the
// handle then launches it for the IIR modes; tools/w8_model.py and tools/w8_debug.py belong to it.
//
// Same algorithm as chain_f32.hip (predict / scan / recurse per section, FFT in registers, split step), other partition:
//   * thread t owns samples [32t, 32t+32) as two consecutive chunks of 16, held as float pairs (chunk A in .x, chunk B
//     in .y): the serial DF2T recursion of scipy.signal.sosfilt and the predictor run on packed-fp32 instructions;
//   * the two-component scan state travels as ONE register pair and every 2x2 matrix is stored column-major, so a
//     matrix-vector product is two packed FMAs (column x broadcast component) -- wave-uniform matrices straight from
//     their scalar registers (tools/ubench/valu_throughput.hip: on gfx950 a plain fp32 instruction with a scalar or DPP
//     operand costs what a packed one costs, so the scan is written in packed form too);
//   * z[m] = x[2m] + i x[2m+1] is transformed as TWO 4096-point FFTs, E over the even and O over the odd z (thread
//     parity selects which), each 16 x 16 x 16 in registers; the missing radix-2 step Z[k] = E[k] + W_8192^k O[k] is merged
//     into the split step of the packed real FFT: the thread that writes bins kappa .. kappa+3 forms exactly the two Z it needs;
//   * LDS holds half a frame (as in chain_f32.hip); the 16x16 transposes between the last two passes go through LDS one
//     float plane at a time so that all 32 sixteen-lane groups have room at once (no workgroup barrier there).
// tools/w8_model.py is the index model of everything below (checked against numpy.fft on the CPU).
#include "../chain_f32_dev.hpp"

namespace {

#ifndef SA8_WAVES_PER_SIMD
#define SA8_WAVES_PER_SIMD 8          // launch bound (A/B builds: 6 = three workgroups per CU at 80 VGPRs)
#endif
constexpr int kT8 = SA8_NTHREADS;
constexpr int kImg8 = 4352;                       // complex slots of the exchange image (34 KiB)
constexpr int kScr8 = kImg8 * 8;                  // scan scratch: 6 sections x 32 rows x float2
constexpr int kSide8 = kScr8 + 6 * SA8_ROWS * 8;  // four complex side slots: E, O at bin 1024 (round 0) / 3072 (round 1)
constexpr int kLane8 = kSide8 + 32;               // the per-lane matrices P2^i of the six sections (SaIirLaneTab8::p), 1.5 KiB:
                                                  // read per section with a 32-bit LDS address instead of a 64-bit global one
constexpr int kTwB8 = kLane8 + SA_MAXSEC * 16 * 16; // the pass-B twiddles (SaF32Tables::twB, 2 KiB): LDS latency needs no early loads
constexpr int kLds8 = kTwB8 + 8 * 16 * 16;          // 39 968 B: four workgroups per CU (40 KiB each)

// r = add + c0 * v.x + c1 * v.y : a 2x2 matrix (columns c0, c1) times v, plus add.  _s: wave-uniform columns in scalar
// register pairs; _v: per-lane columns.  One asm statement each (no compiler pad between the dependent FMAs).
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

// z += c0 * v.x + c1 * v.y in place (no copy of the result through a control-flow merge)
__device__ __forceinline__ void mv_acc_s(v2f &z, const v2f c0, const v2f c1, const v2f v)
{
    asm("v_pk_fma_f32 %0, %1, %3, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
        : "+v"(z) : "s"(c0), "s"(c1), "v"(v));
}

// z <- z + P * shifted(z): one Kogge-Stone level of the affine scan inside a 16-lane row (two DPP moves, two packed FMAs)
template <int N>
__device__ __forceinline__ void scan_level8(v2f &z, const float (&p)[4])
{
    const v2f u = {row_shr<N>(z.x), row_shr<N>(z.y)};
    mv_acc_s(z, v2f{p[0], p[1]}, v2f{p[2], p[3]}, u);
}

typedef float v4f __attribute__((ext_vector_type(4)));

// Eight 8-byte / sixteen 4-byte LDS reads at compile-time strides AND the wait for them in one asm statement: the outputs are
// valid when the statement ends.  (Reads left to the compiler are merged into two-element reads whose halves are then
// copied to their places; reads in separate asm statements may have their outputs copied before the data has arrived.)
template <int STRIDE>
__device__ __forceinline__ void lds_read8_b64(cf &r0, cf &r1, cf &r2, cf &r3, cf &r4, cf &r5, cf &r6, cf &r7, unsigned addr)
{
    asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:%9\n\tds_read_b64 %2, %8 offset:%10\n\t"
                 "ds_read_b64 %3, %8 offset:%11\n\tds_read_b64 %4, %8 offset:%12\n\tds_read_b64 %5, %8 offset:%13\n\t"
                 "ds_read_b64 %6, %8 offset:%14\n\tds_read_b64 %7, %8 offset:%15\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                 : "v"(addr), "i"(STRIDE), "i"(2 * STRIDE), "i"(3 * STRIDE), "i"(4 * STRIDE), "i"(5 * STRIDE), "i"(6 * STRIDE),
                   "i"(7 * STRIDE)
                 : "memory");
}
__device__ __forceinline__ void lds_read16_b32(float (&r)[16], unsigned addr)
{
    asm volatile("ds_read_b32 %0, %16\n\tds_read_b32 %1, %16 offset:4\n\tds_read_b32 %2, %16 offset:8\n\t"
                 "ds_read_b32 %3, %16 offset:12\n\tds_read_b32 %4, %16 offset:16\n\tds_read_b32 %5, %16 offset:20\n\t"
                 "ds_read_b32 %6, %16 offset:24\n\tds_read_b32 %7, %16 offset:28\n\tds_read_b32 %8, %16 offset:32\n\t"
                 "ds_read_b32 %9, %16 offset:36\n\tds_read_b32 %10, %16 offset:40\n\tds_read_b32 %11, %16 offset:44\n\t"
                 "ds_read_b32 %12, %16 offset:48\n\tds_read_b32 %13, %16 offset:52\n\tds_read_b32 %14, %16 offset:56\n\t"
                 "ds_read_b32 %15, %16 offset:60\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]),
                   "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15])
                 : "v"(addr)
                 : "memory");
}

// e + o * u and e + o * conj(u): complex multiply-add as two packed FMAs (broadcast, swap and negation are operand modifiers)
__device__ __forceinline__ cf cfma(const cf e, const cf o, const cf u)
{
    cf r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"                                // e + o.x * (u.x, u.y)
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"                       // + o.y * (-u.y, u.x)
        : "=&v"(r) : "v"(o), "v"(u), "v"(e));
    return r;
}
__device__ __forceinline__ cf cfma_conj(const cf e, const cf o, const cf u)
{
    cf r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]\n\t"                 // e + o.x * (u.x, -u.y)
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]"                                      // + o.y * (u.y, u.x)
        : "=&v"(r) : "v"(o), "v"(u), "v"(e));
    return r;
}

// nA += tap * y.x, nB += tap * y.y with tap = (m1, m2) wave-uniform in an aligned scalar pair: the chunk-end states of
// chunk A and chunk B (from zero state) accumulate as (z1, z2) pairs
__device__ __forceinline__ void tap8(v2f &nA, v2f &nB, const v2f tap, const v2f y)
{
    asm("v_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %1, %2, %3, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
        : "+v"(nA), "+v"(nB) : "s"(tap), "v"(y));
}
// the first tap of an accumulator: a product, so that the accumulators are never zeroed
__device__ __forceinline__ void tap8_first(v2f &nA, v2f &nB, const v2f tap, const v2f y)
{
    asm("v_pk_mul_f32 %0, %2, %3 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %1, %2, %3 op_sel:[0,1] op_sel_hi:[1,1]"
        : "=&v"(nA), "=&v"(nB) : "s"(tap), "v"(y));
}

// ---------------------------------------------------------------------------------------------
// Stage-in.  Wave w owns rows 64w .. 64w+63 of the frame (a row = one thread's 32 samples = 128 B) and a private 4 KiB
// of LDS = 32 rows: two sub-rounds, the lanes 32s .. 32s+31 pick their rows up in sub-round s.  No workgroup barrier:
// a wave waits for its own LDS-DMA (vmcnt) and, before overwriting the rows, for its own reads (lgkmcnt).  The DMA
// writes LDS linearly per instruction (1 KiB = 8 rows), so the XOR swizzle that makes the per-lane ds_read_b128
// conflict-free (16-byte column c of row r at c ^ ((r >> 1) & 7)) is applied to the per-lane SOURCE address.
// Window: evaluated in place (WINGEN; W = G0 + P_h c_j + Q_h s_j, the angle-addition form of a0 - a1 cos(theta n) --
// scripts/hann_coeff.py:3-4 -- as two packed FMAs per sample pair) or loaded from the plan's table.
// Thread t ends with d[j] = (x[32t + j], x[32t + 16 + j]) * window.
template <bool WINGEN>
__device__ __forceinline__ void stage_in8(const float *__restrict__ xin, const SaIirLaneTab8 *__restrict__ lt,
                                          unsigned char *smem, int t, v2f (&d)[16])
{
    const int lane = t & 63, wave = t >> 6;
    unsigned char *region = smem + wave * 4096;
    const float4 *lds4 = reinterpret_cast<const float4 *>(region);
    float4 raw[8];
    // defined on every path without an instruction: the lanes of the other sub-round never read what they hold
#pragma unroll
    for (int g = 0; g < 8; ++g) asm volatile("" : "=v"(raw[g].x), "=v"(raw[g].y), "=v"(raw[g].z), "=v"(raw[g].w));
    float4 pq = make_float4(0.f, 0.f, 0.f, 0.f);
    float g0 = 0.f;
    if constexpr (WINGEN) {
        pq = *reinterpret_cast<const float4 *>(&lt->wgen[t][0]);
        g0 = lt->wg0;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (s == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(3);
        // slab i = rows 8i .. 8i+7 of the sub-round; the swizzle term of row rr = 8i + (lane >> 3) is ((rr >> 1) & 7) =
        // (4 i + (lane >> 4)) & 7: slabs i and i + 2 differ by 2048 bytes of source (the instruction's offset field)
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int rr = 8 * par + (lane >> 3);
            const int lc = (lane & 7) ^ ((rr >> 1) & 7);
            const float *src = xin + (64 * wave + 32 * s + rr) * 32 + lc * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(region + par * 1024), 16, 0, SA_DMA_AUX);
            // (the instruction's offset field moves the source AND the LDS destination: slab par + 2 = slab par + 2048 bytes)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(region + par * 1024), 16, 2048,
                                             SA_DMA_AUX);
        }
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if ((lane >> 5) == s) {
            const int rr = lane & 31, sw = (rr >> 1) & 7;
#pragma unroll
            for (int g = 0; g < 8; ++g) raw[g] = lds4[rr * 8 + (g ^ sw)];
        }
    }
    const v2f P = {pq.x, pq.y}, Q = {pq.z, pq.w}, G0 = {g0, g0};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float xa[4] = {raw[g].x, raw[g].y, raw[g].z, raw[g].w};
        const float xb[4] = {raw[4 + g].x, raw[4 + g].y, raw[4 + g].z, raw[4 + g].w};
        float4 wt[2];
        if constexpr (!WINGEN) {
            wt[0] = reinterpret_cast<const float4 *>(lt->win_t)[(2 * g) * kT8 + t];
            wt[1] = reinterpret_cast<const float4 *>(lt->win_t)[(2 * g + 1) * kT8 + t];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = 4 * g + e;
            v2f w;
            if constexpr (WINGEN) {
                const v2f cs = {lt->wcs[j][0], lt->wcs[j][1]};                       // wave-uniform: SGPR pair
                asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
                    "v_pk_fma_f32 %0, %4, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
                    : "=&v"(w) : "v"(P), "s"(cs), "v"(G0), "v"(Q));
            } else {
                const float4 q = wt[e >> 1];
                w = (e & 1) ? v2f{q.z, q.w} : v2f{q.x, q.y};
            }
            // one plain multiply per half, straight into its place in the pair (VGPR-only operands: the fast class).
            // Left to the SLP vectoriser this becomes a packed multiply on pairs assembled with two copies each.
            asm("v_mul_f32 %0, %1, %2" : "=v"(d[j].x) : "v"(xa[e]), "v"(w.x));
            asm("v_mul_f32 %0, %1, %2" : "=v"(d[j].y) : "v"(xb[e]), "v"(w.y));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The wave-uniform constants of one section, requested one section ahead.
struct Sec8 {
    v2f pc0, pc1, mb0, mb1;
    float b0, b1, b2, a1, a2;
    int flags;
};
template <typename SecT>
__device__ __forceinline__ Sec8 load_sec8(const SecT &k)
{
    return {v2f{k.pc[0], k.pc[1]}, v2f{k.pc[2], k.pc[3]}, v2f{k.mback[0], k.mback[1]}, v2f{k.mback[2], k.mback[3]},
            k.c[0], k.c[1], k.c[2], k.c[3], k.c[4], k.flags};
}

// One cascade section, in place on the thread's two chunks.
//   zA, zB (in) : predicted end states (z1, z2) of chunk A and chunk B from zero state, pole coordinates
//   zA, zB (out): the same for the NEXT section (accumulated over this section's outputs)
template <bool PREDICT_NEXT, bool UNIT, typename SecT>
__device__ __forceinline__ void iir_section8(v2f (&d)[16], const SecT &k, const SecT &knext, const Sec8 c, Sec8 &cn,
                                             const float4 *lanetab_lds, float2 *scr_s, int lane, int wave, v2f &zA, v2f &zB)
{
    // state after both chunks of this thread, from zero state: T = Pc zA + zB
    v2f T = mv_s(c.pc0, c.pc1, zA, zB);
    const int flags = c.flags;
    // inclusive affine scan inside the 16-lane row; levels whose transition power is below float resolution are skipped
    if (!(flags & 1)) scan_level8<1>(T, k.plev[0]);
    if (!(flags & 2)) scan_level8<2>(T, k.plev[1]);
    if (!(flags & 4)) scan_level8<4>(T, k.plev[2]);
    if (!(flags & 8)) scan_level8<8>(T, k.plev[3]);
    const int row = 4 * wave + (lane >> 4);                      // 0 .. 31
    if ((lane & 15) == 15) scr_s[row] = make_float2(T.x, T.y);
    const v2f e = {row_shr<1>(T.x), row_shr<1>(T.y)};            // exclusive: state before this thread, row-local
    lds_barrier();
    const float4 lanep = lanetab_lds[lane & 15];                 // P2^i (in LDS since the kernel's first instructions)
    v2f cst;
    if (flags & SA_IIR_SKIP_ROWSCAN) {
        // a row (512 samples) outlasts the section's memory: the row starts from the previous row's total
        const float2 tt = scr_s[(row - 1) & 31];
        cst = v2f{tt.x, tt.y};
    } else {
        // scan over the 32 row totals, two per lane: Q_i = Prow R_2i + R_2i+1, inclusive scan of the Q with the powers of
        // Prow^2 (every 16-lane row of every wave repeats it), then the state in front of row r:
        //   r = 2k: J_(k-1)      r = 2k+1: Prow J_(k-1) + R_2k
        const float4 rr = reinterpret_cast<const float4 *>(scr_s)[lane & 15];
        v2f Qs = mv_s(v2f{k.prow[0][0], k.prow[0][1]}, v2f{k.prow[0][2], k.prow[0][3]}, v2f{rr.x, rr.y}, v2f{rr.z, rr.w});
        scan_level8<1>(Qs, k.prow[1]);
        scan_level8<2>(Qs, k.prow[2]);
        scan_level8<4>(Qs, k.prow[3]);
        scan_level8<8>(Qs, k.prow[4]);
        const int kk = row >> 1;
        const int src = (lane & 48) | ((kk - 1) & 15);
        v2f J = {lane_get(Qs.x, src), lane_get(Qs.y, src)};
        if (kk == 0) J = v2f{0.f, 0.f};
        const float2 rm = scr_s[(row - 1) & 31];
        const v2f odd = mv_s(v2f{k.prow[0][0], k.prow[0][1]}, v2f{k.prow[0][2], k.prow[0][3]}, J, v2f{rm.x, rm.y});
        cst = (row & 1) ? odd : J;
    }
    if (row == 0) cst = v2f{0.f, 0.f};
    // start state of chunk A: row-local part + P2^i * (row start state); chunk B: Pc sA + zA
    const v2f aS = mv_v(v2f{lanep.x, lanep.y}, v2f{lanep.z, lanep.w}, cst, e);
    const v2f bS = mv_s(c.pc0, c.pc1, aS, zA);
    // pole coordinates -> DF2T states of the recursion, re-paired as (chunk A, chunk B)
    const v2f q1 = {aS.x, bS.x}, q2 = {aS.y, bS.y};
    v2f s1 = c.mb0.x * q1 + c.mb1.x * q2, s2 = c.mb0.y * q1 + c.mb1.y * q2;
    const float b0 = c.b0, b1 = c.b1, b2 = c.b2, na1 = -c.a1, na2 = -c.a2;
#pragma unroll
    for (int j = 0; j < SA8_CHUNK; ++j) {
        const v2f x = d[j];
        v2f y;
        if constexpr (UNIT) {                 // b = [1, r1, 1]: the cascade gain sits in the window
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
    if constexpr (PREDICT_NEXT) {
        // the next section's tap pairs and constants: requested here, behind the recursion -- eight waves per SIMD cover the
        // scalar-load latency, and 32 + 14 scalar registers less are live across the recursion (the kernel must stay within
        // 80 SGPRs for eight waves per SIMD)
        __builtin_amdgcn_sched_barrier(0);
        v2f tp[SA8_CHUNK];
#pragma unroll
        for (int j = 0; j < SA8_CHUNK; ++j) tp[j] = v2f{k.mnext[j][0], k.mnext[j][1]};
        cn = load_sec8(knext);
        // one accumulator pair per chunk: with eight waves on the SIMD the other waves cover the dependent FMAs, and
        // a second pair costs four registers of the 64
        tap8_first(zA, zB, tp[0], d[0]);
#pragma unroll
        for (int j = 1; j < SA8_CHUNK; ++j) tap8(zA, zB, tp[j], d[j]);
    }
}

template <int S, int NSEC, bool UNIT, typename PlanT>
__device__ __forceinline__ void iir_sections8(v2f (&d)[16], const PlanT &ka, const SaIirLaneTab8 *__restrict__ lt, float2 *scr,
                                              int lane, int wave, v2f &zA, v2f &zB, const Sec8 c)
{
    if constexpr (S < NSEC) {
        const float4 *lanep = reinterpret_cast<const float4 *>(reinterpret_cast<const unsigned char *>(scr) - kScr8 + kLane8) + S * 16;
        Sec8 cn = c;
        iir_section8<(S + 1 < NSEC), UNIT>(d, ka.sec[S], ka.sec[S + 1 < NSEC ? S + 1 : S], c, cn, lanep, scr + SA8_ROWS * S, lane,
                                           wave, zA, zB);
        iir_sections8<S + 1, NSEC, UNIT>(d, ka, lt, scr, lane, wave, zA, zB, cn);
    }
}

template <int NSEC, bool UNIT, typename PlanT>
__device__ __forceinline__ void iir_cascade8(v2f (&d)[16], const PlanT &ka, const SaIirLaneTab8 *__restrict__ lt, float2 *scr, int t)
{
    const Sec8 c0 = load_sec8(ka.sec[0]);
    v2f zA, zB;
    tap8_first(zA, zB, v2f{ka.m0[0][0], ka.m0[0][1]}, d[0]);
#pragma unroll
    for (int j = 1; j < SA8_CHUNK; ++j) tap8(zA, zB, v2f{ka.m0[j][0], ka.m0[j][1]}, d[j]);
    iir_sections8<0, NSEC, UNIT>(d, ka, lt, scr, t & 63, t >> 6, zA, zB, c0);
}

template <int NSEC, bool UNIT, int OUT, bool WINGEN>
__global__ __launch_bounds__(kT8, SA8_WAVES_PER_SIMD) void chain_f32_w8_kernel(const float *__restrict__ in, void *__restrict__ out, int batch,
                                                              const float4 *__restrict__ twT8, const float4 *__restrict__ twB,
                                                              const float2 *__restrict__ twC,
                                                              const SaIirLaneTab8 *__restrict__ lanetab, const SaIirK8 ka)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int f = blockIdx.x;
    if (f >= batch) return;
    cf *ldc = reinterpret_cast<cf *>(smem);
    float *ldf = reinterpret_cast<float *>(smem);
    float2 *scr = reinterpret_cast<float2 *>(smem + kScr8);
    cf *side = reinterpret_cast<cf *>(smem + kSide8);
    const int t = threadIdx.x;
    const float *xin = in + (size_t)f * SA_NPTS;
    cf a[16];
    SA_STAMP(0);
    {
        // the lane matrices of the plan go to LDS (first read behind the first scan barrier)
        if (t < NSEC * 16) reinterpret_cast<float4 *>(smem + kLane8)[t] = reinterpret_cast<const float4 *>(lanetab->p)[t];
        if (t >= 128 && t < 256) reinterpret_cast<float4 *>(smem + kTwB8)[t - 128] = twB[t - 128];
        v2f d[16];
        stage_in8<WINGEN>(xin, lanetab, smem, t, d);
        SA_STAMP(1);
#if !defined(SA_W8_DEBUG) || SA_W8_DEBUG >= 2
        iir_cascade8<NSEC, UNIT>(d, ka, lanetab, scr, t);
#endif
        SA_STAMP(2);
#ifdef SA_W8_DEBUG          // diagnostic builds only (tools/w8_debug.py): 1 = the windowed samples, 2 = the cascade's output
        {
            float *o = reinterpret_cast<float *>(out) + (size_t)f * SA_NPTS + 32 * t;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                o[j] = 2.f * d[j].x;
                o[16 + j] = 2.f * d[j].y;
            }
            return;
        }
#endif
        // ---- exchange to the pass-A layout, two rounds by writer half: the owners write z[16 t' + i] = (x[2i], x[2i+1]) at
        //      17 t' + i (real and imaginary part sit in different register pairs of d[]: two dwords at adjacent addresses,
        //      one ds_write2_b32, no register copies); everybody reads z[512 m1 + t]
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            lds_barrier();
            if ((t >> 8) == h) {
                const unsigned zw = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)(smem) +
                                    8u * 17u * (unsigned)(t & 255);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(zw), "v"(d[2 * i].x), "v"(d[2 * i + 1].x),
                                 "i"(2 * i), "i"(2 * i + 1) : "memory");
                    asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(zw), "v"(d[2 * i].y), "v"(d[2 * i + 1].y),
                                 "i"(2 * (8 + i)), "i"(2 * (8 + i) + 1) : "memory");
                }
            }
            lds_barrier();
#pragma unroll
            for (int m = 0; m < 8; ++m) a[safft::brev(8 * h + m, 4)] = ldc[544 * m + 17 * (t >> 4) + (t & 15)];
        }
    }
    // ---- pass A: 16-point FFT over m1 (stride 512), then twiddle W_4096^(k1 u), u = t >> 1, from six per-thread
    //      anchors W^(b u), b = 1..3, and W^(4 a u), a = 1..3 (one or two complex products per point)
    SA_STAMP(3);
    float4 an[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) an[i] = twT8[i * kT8 + t];
    safft::fft_dit<16>(a);
    {
        const cf wb[4] = {{1.f, 0.f}, {an[0].x, an[0].y}, {an[0].z, an[0].w}, {an[1].x, an[1].y}};
        const cf wa[4] = {{1.f, 0.f}, {an[1].z, an[1].w}, {an[2].x, an[2].y}, {an[2].z, an[2].w}};
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) {
            if ((k1 & 3) != 0) a[k1] = safft::cmul(a[k1], wb[k1 & 3]);
            if ((k1 >> 2) != 0) a[k1] = safft::cmul(a[k1], wa[k1 >> 2]);
        }
    }
    SA_STAMP(4);
    // ---- exchange A -> B in two rounds.  The pass-B thread tb = 32 k1 + 16 p + b needs A'[k1][u = 16 a + b], a = 0..15, of its
    //      sub-FFT p, from the writers 2u + p = 32 a + 2 b + p.  Registers and LDS hold the data exactly once, so every thread
    //      must hand over as many values per round as it takes in: eight.  Round q pairs writers and readers whose halves
    //      (writer: a >> 3 = t >> 8; reader: k1 >> 3 = tb >> 8) differ by q: the writer stores its k1 = 8 (hi ^ q) + r at
    //      512 r + t, the reader takes a = 8 (hi ^ q) + a' from row k1 & 7.  The half is wave-uniform: both arms keep
    //      compile-time register indices.
    const int k1B = t >> 5, pB = (t >> 4) & 1, lo = t & 15;
    {
        cf bq[16];
        const bool hi = __builtin_amdgcn_readfirstlane(t >> 8) != 0;
        const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)(smem);
        const unsigned wr = lbase + 8u * (unsigned)t;                                            // + 4096 per row
        const unsigned rdb = lbase + 8u * (unsigned)(512 * (k1B & 7) + 2 * lo + pB);             // + 256 per a
        // LDS accesses written out: left to the compiler, the two arms' stores are sunk into one sequence behind sixteen
        // register copies (and the copies spill)
#define SA8_WR(SLOT, ROW) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(wr), "v"(a[SLOT]), "i"(4096 * (ROW)) : "memory")
#define SA8_RD(A0) lds_read8_b64<256>(bq[safft::brev((A0), 4)], bq[safft::brev((A0) + 1, 4)], bq[safft::brev((A0) + 2, 4)], \
                                 bq[safft::brev((A0) + 3, 4)], bq[safft::brev((A0) + 4, 4)], bq[safft::brev((A0) + 5, 4)], \
                                 bq[safft::brev((A0) + 6, 4)], bq[safft::brev((A0) + 7, 4)], rdb + 256u * (A0))
        // each half runs BOTH rounds inside its own arm (every wave meets the same four barriers): merged after each
        // round, all sixteen registers of bq would count as live from the first round on
        if (hi) {
            lds_barrier();
#pragma unroll
            for (int r = 0; r < 8; ++r) SA8_WR(8 + r, r);
            lds_barrier();
            SA8_RD(8);
            lds_barrier();
#pragma unroll
            for (int r = 0; r < 8; ++r) SA8_WR(r, r);
            lds_barrier();
            SA8_RD(0);
        } else {
            lds_barrier();
#pragma unroll
            for (int r = 0; r < 8; ++r) SA8_WR(r, r);
            lds_barrier();
            SA8_RD(0);
            lds_barrier();
#pragma unroll
            for (int r = 0; r < 8; ++r) SA8_WR(8 + r, r);
            lds_barrier();
            SA8_RD(8);
        }
#undef SA8_WR
#undef SA8_RD
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = bq[i];
    }
    SA_STAMP(5);
    // ---- pass B: 16-point FFT over a, twiddle W_256^(b k2)
    safft::fft_dit<16>(a);
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) {                       // twB[pp][b] = (W_256^(2pp * b), W_256^((2pp+1) * b))
        const float4 w = reinterpret_cast<const float4 *>(smem + kTwB8)[pp * 16 + lo];
        if (pp > 0) a[2 * pp] = safft::cmul(a[2 * pp], {w.x, w.y});
        a[2 * pp + 1] = safft::cmul(a[2 * pp + 1], {w.z, w.w});
    }
    SA_STAMP(6);
    // ---- exchange B -> C: a 16x16 transpose inside each 16-lane group, one float plane at a time (272 floats per group,
    //      pitch 17): all 32 groups fit the image at once.  Only the group's own lanes touch its region and the LDS
    //      executes a wave's accesses in order: no workgroup barrier inside; one in front (round 1 of the previous
    //      exchange is still being read by the other half of the workgroup).
    lds_barrier();            // the other half of the workgroup may still be reading the image of round 1
    {
        // plane 0: the real parts leave, the transposed ones come back as sixteen single registers; plane 1: the same for
        // the imaginary parts; then (re, im) pairs.  Single-dword reads in one asm statement each: the register allocator
        // places re[b], im[b] side by side and no value is copied.
        float *gb = ldf + (t >> 4) * 272;
        const unsigned rd = (unsigned)(size_t)(__attribute__((address_space(3))) float *)(gb + lo * 17);
        float re[16], im[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) gb[c * 17 + lo] = a[c].x;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        lds_read16_b32(re, rd);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int c = 0; c < 16; ++c) gb[c * 17 + lo] = a[c].y;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        lds_read16_b32(im, rd);
#pragma unroll
        for (int b = 0; b < 16; ++b) a[safft::brev(b, 4)] = cf{re[b], im[b]};
    }
    SA_STAMP(7);
    // ---- pass C: 16-point FFT over b -> k3;  F_p[k1 + 16 k2 + 256 k3], k2 = lo
    safft::fft_dit<16>(a);
    SA_STAMP(8);
    // ---- natural-order image of (E, O) pairs + output stage, two rounds: round 0 = k3 in {0..3, 12..15} (bins k < 1024 and
    //      their partners 4096 - k), round 1 = k3 in {4..11}.  Bin 1024 (needed by round 0's last group) and bin 3072 (its
    //      partner in round 1) are not in that round's image and travel through four side slots.
    // The thread's group: bins kappa0 .. kappa0+3 (+ one for the mirrored streams), kappa0 = 4g + 1024 r + 4096 half.  It reads
    // E, O at k = kappa mod 4096 and at 4096 - k and forms the two Z of its pair with its OWN twiddle squared:
    //   u = W_16384^(2 kappa):  Z[kappa] = E[k] + u O[k],  Z[8192 - kappa] = E[4096-k] + conj(u) O[4096-k]
    // (for half = 1 that is E - W_8192^k O and E' - conj(W_8192^k) O': the same code).
    // Where the group's pairs sit in the image does not depend on the round (compacted bin of k0 = 4g is 4g in both, of
    // 4096 - k0 - 4 it is 4 (511 - g)): four byte addresses, computed once -- run of four pairs at k0, the pair at k0 + 4,
    // run of four at 4096 - k0 - 4 .. 4096 - k0 - 1, the pair at 4096 - k0.
    // split-step anchors: W_16384^(4g + 4096 half) and the next group's (chain_f32.hip: bit-identical mirrored halves)
    // (an opaque copy of the thread index: everything the output stage derives from it -- the anchor load, five LDS
    //  addresses -- is computed here and not hoisted to the top of the kernel, kept across it and spilled)
    int to = t;
    asm volatile("" : "+v"(to));
    const float4 an3 = twT8[3 * kT8 + to];
    const cf wP = {an3.x, an3.y}, wPn = {an3.z, an3.w};
    const int g = to >> 1, half = to & 1;
    const int g1 = g + 1, gm = 511 - g, gz = 512 - g;
    const int A1 = 64 * g + 16 * (g >> 2), A2 = 64 * g1 + 16 * (g1 >> 2), A3 = 64 * gm + 16 * (gm >> 2);
    const int A4 = 64 * gz + 16 * (gz >> 2);
    const int k1o = to >> 5, po = (to >> 4) & 1, loo = to & 15;
    const int wimg = 8 * (2 * k1o + 32 * loo + po + 2 * ((2 * k1o + 32 * loo + po) >> 5));      // writer: + 4352 bytes per dd
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        lds_barrier();
#pragma unroll
        for (int dd = 0; dd < 8; ++dd) {
            const int k3 = (r == 0) ? (dd < 4 ? dd : dd + 8) : dd + 4;
            *reinterpret_cast<cf *>(smem + wimg + 4352 * dd) = a[k3];
        }
        if (k1o == 0 && loo == 0) side[2 * r + po] = a[r == 0 ? 4 : 12];       // bin 1024 (round 0) / 3072 (round 1)
        lds_barrier();
        SA_STAMP(9 + r);
        const int k0 = 4 * g + 1024 * r;
        // round 0: bin 1024 (g = 255, e = 4) comes from the side slots, the partner of bin 0 is bin 0;
        // round 1: the partner of bin 1024 (g = 0, e = 0) is bin 3072: side slots
        const int A2r = (r == 0 && g == 255) ? kSide8 : A2;
        const int A4r = (g == 0) ? (r == 0 ? 0 : kSide8 + 16) : A4;
        cf R[5], I[5];
        float mp[5], mq[5];
#pragma unroll
        for (int e = 0; e < 5; ++e) {
            const v4f eo = *reinterpret_cast<const v4f *>(smem + (e < 4 ? A1 + 16 * e : A2r));
            const v4f eop = *reinterpret_cast<const v4f *>(smem + (e == 0 ? A4r : A3 + 16 * (4 - e)));
            cf w;
            if (e < 4) {
                w = cmul_s(wP, twC[r * 5 + e]);
            } else {
                // bin kappa0 + 4 is bin 0 of the neighbouring group, which also stores it: the mirrored halves of the
                // spectrum stay bit-identical only if both evaluate the same product (chain_f32.hip)
                const float2 c0 = twC[r * 5], c1 = twC[(r + 1) * 5];
                const cf csel = (g == 255) ? cf{c1.x, c1.y} : cf{c0.x, c0.y};
                w = safft::cmul(wPn, csel);
            }
            const cf u = safft::cmul(w, w);
            const cf zk = cfma(cf{eo.x, eo.y}, cf{eo.z, eo.w}, u);
            const cf zm = cfma_conj(cf{eop.x, eop.y}, cf{eop.z, eop.w}, u);
            split_eval(zk, zm, w, R[e], I[e]);
            if constexpr (OUT == SA_OUT_MAG_FULL) {
                // bins 2048 / 14336 and 6144 / 10240 are written by the two halves of the last group from the two sides of
                // the SAME pair (P of one half, Q of the other): equal in exact arithmetic; to keep the mirrored halves of
                // the spectrum bit-identical both lanes take their Q from the neighbouring lane's P
                if (e == 4 && r == 1 && (to >> 6) == 7) {
                    const float nr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, R[4].x), 0xB1, 0xF, 0xF, true));
                    const float ni = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, I[4].x), 0xB1, 0xF, 0xF, true));
                    if (g == 255) {
                        R[4].y = nr;
                        I[4].y = ni;
                    }
                }
            }
            if constexpr (OUT != SA_OUT_SPEC_HALF) {
                const cf m2 = safft::pk_fma(I[e], I[e], R[e] * R[e]);          // (|P|^2, |Q|^2)
                mp[e] = fast_sqrt(m2.x);
                mq[e] = fast_sqrt(m2.y);
            }
        }
        const int kap0 = k0 + 4096 * half;
        if constexpr (OUT == SA_OUT_MAG_FULL) {
            float *o = reinterpret_cast<float *>(out) + (size_t)f * SA_NPTS;
            store_nt(o + kap0, mp[0], mp[1], mp[2], mp[3]);
            store_nt(o + SA_NPTS - kap0 - 4, mp[4], mp[3], mp[2], mp[1]);
            store_nt(o + SA_MC + kap0, mq[0], mq[1], mq[2], mq[3]);
            store_nt(o + SA_MC - kap0 - 4, mq[4], mq[3], mq[2], mq[1]);
        } else if constexpr (OUT == SA_OUT_MAG_HALF) {
            float *o = reinterpret_cast<float *>(out) + (size_t)f * (SA_MC + 1);     // rows are not 16-byte aligned
#pragma unroll
            for (int e = 0; e < 4; ++e) store_nt(o + kap0 + e, mp[e]);
#pragma unroll
            for (int e = 1; e < 5; ++e) store_nt(o + SA_MC - kap0 - e, mq[e]);
            if (kap0 == 0) store_nt(o + SA_MC, mq[0]);
        } else {
            split_store<OUT>(R, I, out, f, kap0);
        }
    }
    SA_STAMP(11);
}

template <typename K>
hipError_t set_lds8(K kernel)
{
    return sa_set_dyn_lds_once(reinterpret_cast<const void *>(kernel), kLds8);
}

template <int NSEC, bool UNIT>
hipError_t launch8(const float *in, void *out, int batch, int out_kind, const SaF32Tables &tb, const SaIirK8 &ka,
                   hipStream_t stream, SaLaunchEv ev)
{
    const dim3 grid(batch), block(kT8);
    hipError_t e = hipSuccess;
#define SA_LAUNCH8(OUTK)                                                                               \
    do {                                                                                               \
        auto kern = ka.wingen ? chain_f32_w8_kernel<NSEC, UNIT, OUTK, true>                            \
                              : chain_f32_w8_kernel<NSEC, UNIT, OUTK, false>;                          \
        e = set_lds8(kern);                                                                            \
        if (e != hipSuccess) return e;                                                                 \
        hipExtLaunchKernelGGL(kern, grid, block, kLds8, stream, ev.start, ev.stop, 0, in, out, batch, tb.twT8, tb.twB, tb.twC, \
                              tb.lanetab8, ka);                                                        \
    } while (0)
    switch (out_kind) {
        case SA_OUT_MAG_FULL: SA_LAUNCH8(SA_OUT_MAG_FULL); break;
        case SA_OUT_MAG_HALF: SA_LAUNCH8(SA_OUT_MAG_HALF); break;
        case SA_OUT_SPEC_HALF: SA_LAUNCH8(SA_OUT_SPEC_HALF); break;
        default: return hipErrorNotSupported;
    }
#undef SA_LAUNCH8
    return hipGetLastError();
}

}  // namespace

hipError_t sa_launch_chain_f32_w8(const float *in, void *out, int batch, int out_kind, const SaF32Tables &tb,
                                  hipStream_t stream, SaLaunchEv ev)
{
    if (batch <= 0) return hipSuccess;
    if (!tb.iir8 || tb.iir8->nsec <= 0 || out_kind == SA_OUT_TIME) return hipErrorNotSupported;
    const SaIirK8 &ka = *tb.iir8;
    const bool unit = ka.unit != 0;
    switch (ka.nsec) {
        case 2: return unit ? launch8<2, true>(in, out, batch, out_kind, tb, ka, stream, ev)
                            : launch8<2, false>(in, out, batch, out_kind, tb, ka, stream, ev);
        case 4: return unit ? launch8<4, true>(in, out, batch, out_kind, tb, ka, stream, ev)
                            : launch8<4, false>(in, out, batch, out_kind, tb, ka, stream, ev);
        case 6: return unit ? launch8<6, true>(in, out, batch, out_kind, tb, ka, stream, ev)
                            : launch8<6, false>(in, out, batch, out_kind, tb, ka, stream, ev);
        default: return hipErrorNotSupported;
    }
}
