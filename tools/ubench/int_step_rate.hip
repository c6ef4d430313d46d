// Micro-benchmark, round 4: wall-clock time per step of the integer cascades' inner blocks, one wave per SIMD on every
// SIMD of the chip (256 workgroups x 4 waves), eight blocks per asm statement and named registers as in the product
// loops (chain_q15.hip).  No LDS, no refill, no stores: the floor of the recursion itself.
//   * the Q7 seven-instruction block as shipped in round 3 (order H I B A C E F: the neighbour's output is read by a DPP
//     product two instructions after it was written and sits on the recurrence H -> C -> F -> H)
//   * the same arithmetic retimed: every DPP product reads a neighbour output that is one or two blocks old, hi(p0) is
//     folded into the partial sum off the chain, the recurrence is H -> A -> H (new/filter_iir_cust.vhd:96-100)
//   * the nine-instruction block (B1 != 0) as shipped and retimed
//   * the wide Q2.14 block (ten instructions: packed-int16 dot products, split accumulators), serial and with the
//     feed-forward half computed a block ahead
//   * issue / latency probes for v_dot2_i32_i16
//   hipcc -O3 --offload-arch=gfx950 int_step_rate.hip -o int_step_rate && ./int_step_rate
#include <hip/hip_runtime.h>
#include <cstdio>

// ---------------------------------------------------------------------------------------------- Q7, as shipped
#define I_H(Y) "v_add_u32_sdwa " Y ", %[s2], %[t] dst_sel:WORD_0 dst_unused:UNUSED_SEXT src0_sel:DWORD src1_sel:DWORD\n\t"
#define I_I "v_add_u32_sdwa %[s2], %[p2], %[p3] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"
#define I_B(H1) "v_mul_i32_i24_dpp %[p2], " H1 ", %[cB0] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_C(Y) "v_mul_i32_i24_dpp %[p0], " Y ", %[cB2] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_D(Y) "v_mul_i32_i24_dpp %[p1], " Y ", %[cB1] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_A(Y) "v_mad_i32_i24 %[p4], " Y ", %[nA1], %[k]\n\t"
#define I_E(Y) "v_mad_i32_i24 %[p3], " Y ", %[nA0], %[k]\n\t"
#define I_F "v_add_u32_sdwa %[t], %[p0], %[p4] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"
#define I_G "v_add_u32_sdwa %[u], %[p1], %[p2] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"
#define I_I9 "v_add_u32_sdwa %[s2], %[u], %[p3] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
#define OLD7(Y, H1) I_H(Y) I_I I_B(H1) I_A(Y) I_C(Y) I_E(Y) I_F
#define OLD9(Y, H1) I_H(Y) I_G I_B(H1) I_C(Y) I_A(Y) I_I9 I_D(Y) I_F I_E(Y)

// ---------------------------------------------------------------------------------------------- Q7, retimed
// y[n] = sext16(S + hi(p4)),  p4 = y[n-1] * -A1 + k,  S = hi(B2 x[n]) + hi(B0 x[n-2]) + hi(-A0 y[n-2] + k) (+ hi(B1 x[n-1]))
// block e: Y1 = the neighbour-visible output of block e-1, Y2 = of block e-2
#define R_H(Y) "v_add_u32_sdwa " Y ", %[S], %[p4] dst_sel:WORD_0 dst_unused:UNUSED_SEXT src0_sel:DWORD src1_sel:WORD_1\n\t"
#define R_A(Y) "v_mad_i32_i24 %[p4], " Y ", %[nA1], %[k]\n\t"
#define R_E(Y) "v_mad_i32_i24 %[p3], " Y ", %[nA0], %[k]\n\t"
#define R_I "v_add_u32_sdwa %[s2], %[p2], %[p3] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"
#define R_C(Y1) "v_mul_i32_i24_dpp %[p0], " Y1 ", %[cB2] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
#define R_B(Y2) "v_mul_i32_i24_dpp %[p2], " Y2 ", %[cB0] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
#define R_D(Y2) "v_mul_i32_i24_dpp %[p1], " Y2 ", %[cB1] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
#define R_F "v_add_u32_sdwa %[S], %[s2], %[p0] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
#define R_G "v_add_u32_sdwa %[u], %[p0], %[p1] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"
#define R_F9 "v_add_u32 %[S], %[s2], %[u]\n\t"
#define NEW7A(Y, Y1, Y2) R_H(Y) R_I R_A(Y) R_E(Y) R_C(Y1) R_F R_B(Y2)
#define NEW7B(Y, Y1, Y2) R_H(Y) R_I R_A(Y) R_E(Y) R_C(Y1) R_B(Y2) R_F
#define NEW7C(Y, Y1, Y2) R_H(Y) R_A(Y) R_I R_E(Y) R_C(Y1) R_B(Y2) R_F
#define NEW7D(Y, Y1, Y2) R_H(Y) R_C(Y1) R_A(Y) R_I R_E(Y) R_F R_B(Y2)
#define NEW9A(Y, Y1, Y2) R_H(Y) R_I R_A(Y) R_E(Y) R_C(Y1) R_D(Y2) R_B(Y2) R_G R_F9
#define NEW9B(Y, Y1, Y2) R_H(Y) R_I R_A(Y) R_C(Y1) R_D(Y2) R_E(Y) R_G R_B(Y2) R_F9

// the same with the feedback products as plain VOP2 multiplies (4-byte encodings) whose high words are SUBTRACTED:
// -floor(A y / 128) needs no rounding constant that way (the k of the mad form only turns the subtraction into an add)
#define S_H(Y) "v_sub_u32_sdwa " Y ", %[S], %[p4] dst_sel:WORD_0 dst_unused:UNUSED_SEXT src0_sel:DWORD src1_sel:WORD_1\n\t"
#define S_A(Y) "v_mul_i32_i24 %[p4], " Y ", %[nA1]\n\t"
#define S_E(Y) "v_mul_i32_i24 %[p3], " Y ", %[nA0]\n\t"
#define S_I "v_sub_u32_sdwa %[s2], %[p2], %[p3] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"
#define NEW7S(Y, Y1, Y2) S_H(Y) S_I S_A(Y) S_E(Y) R_C(Y1) R_B(Y2) R_F
// round-3 order with the short encodings: H I B A C E F, t = hi(p0) - hi(p4)
#define O_F "v_sub_u32_sdwa %[t], %[p0], %[p4] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"
#define OLD7S(Y, H1) I_H(Y) S_I I_B(H1) S_A(Y) I_C(Y) S_E(Y) O_F

#define G8_OLD(BLK) BLK("v52", "v59") BLK("v53", "v52") BLK("v54", "v53") BLK("v55", "v54") BLK("v56", "v55") BLK("v57", "v56") BLK("v58", "v57") BLK("v59", "v58")
#define G8_NEW(BLK)                                                                                                    \
    BLK("v52", "v59", "v58") BLK("v53", "v52", "v59") BLK("v54", "v53", "v52") BLK("v55", "v54", "v53")                \
    BLK("v56", "v55", "v54") BLK("v57", "v56", "v55") BLK("v58", "v57", "v56") BLK("v59", "v58", "v57")

#define Q7_KERNEL(NAME, GROUP)                                                                                         \
    __global__ __launch_bounds__(256) void NAME(int *out, int ngroups, int cc)                                         \
    {                                                                                                                  \
        int s2 = cc, S = cc + 1, cB2 = cc << 9, cB1 = (cc + 1) << 9, cB0 = (cc + 2) << 9, nA0 = -(cc << 9),            \
            nA1 = -((cc + 3) << 9);                                                                                    \
        const int k127 = 127 << 9;                                                                                     \
        int p0 = 1, p1 = 2, p2 = 3, p3 = 4, p4 = 5, t = 6, u = 7, y = threadIdx.x;                                     \
        asm volatile("v_mov_b32 v52, %[y]\n\tv_mov_b32 v53, %[y]\n\tv_mov_b32 v54, %[y]\n\tv_mov_b32 v55, %[y]\n\t"   \
                     "v_mov_b32 v56, %[y]\n\tv_mov_b32 v57, %[y]\n\tv_mov_b32 v58, %[y]\n\tv_mov_b32 v59, %[y]\n\t"   \
                     ".p2align 6\n1:\n\t" GROUP                                                                        \
                     "s_add_i32 %[cnt], %[cnt], -1\n\ts_cmp_lg_u32 %[cnt], 0\n\ts_cbranch_scc1 1b\n\t"                 \
                     "v_mov_b32 %[y], v59"                                                                             \
                     : [y] "+v"(y), [s2] "+v"(s2), [S] "+v"(S), [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), \
                       [p4] "+v"(p4), [t] "+v"(t), [u] "+v"(u), [cnt] "+s"(ngroups)                                    \
                     : [cB2] "v"(cB2), [cB1] "v"(cB1), [cB0] "v"(cB0), [nA0] "v"(nA0), [nA1] "v"(nA1), [k] "s"(k127)   \
                     : "scc", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59");                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s2 + S + p0 + p1 + p2 + p3 + p4 + t + u + y;                      \
    }

Q7_KERNEL(k_old7, G8_OLD(OLD7))
Q7_KERNEL(k_old9, G8_OLD(OLD9))
Q7_KERNEL(k_new7a, G8_NEW(NEW7A))
Q7_KERNEL(k_new7b, G8_NEW(NEW7B))
Q7_KERNEL(k_new7c, G8_NEW(NEW7C))
Q7_KERNEL(k_new7d, G8_NEW(NEW7D))
Q7_KERNEL(k_new9a, G8_NEW(NEW9A))
Q7_KERNEL(k_new9b, G8_NEW(NEW9B))
Q7_KERNEL(k_new7s, G8_NEW(NEW7S))
Q7_KERNEL(k_old7s, G8_OLD(OLD7S))

// ---------------------------------------------------------------------------------------------- wide Q2.14
// serial form (filter_w14_kernel): PP = own pair of block e-1 (the neighbour's is read from the same register name)
#define W_SER(PP, PC, XC, XP, WC, WP)                                                                                  \
    "v_dot2_i32_i16 %[al], " PP ", %[cfbl], %[k]\n\t"                                                                  \
    "v_dot2_i32_i16 %[ah], " PP ", %[cfbh], 0\n\t"                                                                     \
    "v_mov_b32_dpp " XC ", " PP " row_ror:1 row_mask:0xf bank_mask:0xf\n\t"                                            \
    "v_dot2_i32_i16 %[al], " XC ", %[c01l], %[al]\n\t"                                                                 \
    "v_dot2_i32_i16 %[ah], " XC ", %[c01h], %[ah]\n\t"                                                                 \
    "v_dot2_i32_i16 %[al], " XP ", %[c2l], %[al]\n\t"                                                                  \
    "v_dot2_i32_i16 %[ah], " XP ", %[c2h], %[ah]\n\t"                                                                  \
    "v_ashrrev_i32 %[al], 14, %[al]\n\t"                                                                               \
    "v_add_u32 " WC ", %[ah], %[al]\n\t"                                                                               \
    "v_cvt_pk_i16_i32 " PC ", " WP ", " WC "\n\t"
// retimed form: the feed-forward sums (fl, fh) of block e+1 are formed inside block e from neighbour pairs that are at
// least a block old (P2 = the pair of block e-2), interleaved with the feedback chain of block e
#define W_RET(PP, P2, PC, XC, XP, WC, WP)                                                                              \
    "v_dot2_i32_i16 %[al], " PP ", %[cfbl], %[fl]\n\t"                                                                 \
    "v_mov_b32_dpp " XC ", " P2 " row_ror:1 row_mask:0xf bank_mask:0xf\n\t"                                            \
    "v_dot2_i32_i16 %[ah], " PP ", %[cfbh], %[fh]\n\t"                                                                 \
    "v_dot2_i32_i16 %[fl], " XC ", %[c01l], %[k]\n\t"                                                                  \
    "v_ashrrev_i32 %[al], 14, %[al]\n\t"                                                                               \
    "v_dot2_i32_i16 %[fh], " XC ", %[c01h], 0\n\t"                                                                     \
    "v_add_u32 " WC ", %[ah], %[al]\n\t"                                                                               \
    "v_dot2_i32_i16 %[fl], " XP ", %[c2l], %[fl]\n\t"                                                                  \
    "v_cvt_pk_i16_i32 " PC ", " WP ", " WC "\n\t"                                                                      \
    "v_dot2_i32_i16 %[fh], " XP ", %[c2h], %[fh]\n\t"
// serial form with the four accumulating dot products in their 4-byte VOP2 encoding (v_dot2c_i32_i16: D += S0 . S1)
#define W_SHORT(PP, PC, XC, XP, WC, WP)                                                                                \
    "v_dot2_i32_i16 %[al], " PP ", %[cfbl], %[k]\n\t"                                                                  \
    "v_dot2_i32_i16 %[ah], " PP ", %[cfbh], 0\n\t"                                                                     \
    "v_mov_b32_dpp " XC ", " PP " row_ror:1 row_mask:0xf bank_mask:0xf\n\t"                                            \
    "v_dot2c_i32_i16 %[al], " XC ", %[c01l]\n\t"                                                                       \
    "v_dot2c_i32_i16 %[ah], " XC ", %[c01h]\n\t"                                                                       \
    "v_dot2c_i32_i16 %[al], " XP ", %[c2l]\n\t"                                                                        \
    "v_dot2c_i32_i16 %[ah], " XP ", %[c2h]\n\t"                                                                        \
    "v_ashrrev_i32 %[al], 14, %[al]\n\t"                                                                               \
    "v_add_u32 " WC ", %[ah], %[al]\n\t"                                                                               \
    "v_cvt_pk_i16_i32 " PC ", " WP ", " WC "\n\t"
#define WG_SHORT                                                                                                       \
    W_SHORT("v55", "v56", "%[x0]", "%[x1]", "%[w0]", "%[w1]") W_SHORT("v56", "v52", "%[x1]", "%[x0]", "%[w1]", "%[w0]") \
    W_SHORT("v52", "v57", "%[x0]", "%[x1]", "%[w0]", "%[w1]") W_SHORT("v57", "v53", "%[x1]", "%[x0]", "%[w1]", "%[w0]") \
    W_SHORT("v53", "v58", "%[x0]", "%[x1]", "%[w0]", "%[w1]") W_SHORT("v58", "v54", "%[x1]", "%[x0]", "%[w1]", "%[w0]") \
    W_SHORT("v54", "v59", "%[x0]", "%[x1]", "%[w0]", "%[w1]") W_SHORT("v59", "v55", "%[x1]", "%[x0]", "%[w1]", "%[w0]")
#define WG_SER                                                                                                         \
    W_SER("v55", "v56", "%[x0]", "%[x1]", "%[w0]", "%[w1]") W_SER("v56", "v52", "%[x1]", "%[x0]", "%[w1]", "%[w0]")    \
    W_SER("v52", "v57", "%[x0]", "%[x1]", "%[w0]", "%[w1]") W_SER("v57", "v53", "%[x1]", "%[x0]", "%[w1]", "%[w0]")    \
    W_SER("v53", "v58", "%[x0]", "%[x1]", "%[w0]", "%[w1]") W_SER("v58", "v54", "%[x1]", "%[x0]", "%[w1]", "%[w0]")    \
    W_SER("v54", "v59", "%[x0]", "%[x1]", "%[w0]", "%[w1]") W_SER("v59", "v55", "%[x1]", "%[x0]", "%[w1]", "%[w0]")
#define WG_RET                                                                                                         \
    W_RET("v55", "v59", "v56", "%[x0]", "%[x1]", "%[w0]", "%[w1]") W_RET("v56", "v55", "v52", "%[x1]", "%[x0]", "%[w1]", "%[w0]") \
    W_RET("v52", "v56", "v57", "%[x0]", "%[x1]", "%[w0]", "%[w1]") W_RET("v57", "v52", "v53", "%[x1]", "%[x0]", "%[w1]", "%[w0]") \
    W_RET("v53", "v57", "v58", "%[x0]", "%[x1]", "%[w0]", "%[w1]") W_RET("v58", "v53", "v54", "%[x1]", "%[x0]", "%[w1]", "%[w0]") \
    W_RET("v54", "v58", "v59", "%[x0]", "%[x1]", "%[w0]", "%[w1]") W_RET("v59", "v54", "v55", "%[x1]", "%[x0]", "%[w1]", "%[w0]")

#define W_KERNEL(NAME, GROUP)                                                                                          \
    __global__ __launch_bounds__(256) void NAME(int *out, int ngroups, int cc)                                         \
    {                                                                                                                  \
        unsigned c01l = cc * 0x00010003u, c01h = 0x00010000u, c2l = cc & 0xffff, c2h = 1, cfbl = cc * 0x00050007u,    \
                 cfbh = 0xffff0001u;                                                                                   \
        int w0 = cc, w1 = cc + 1, al = 0, ah = 0, fl = 1, fh = 2, y = threadIdx.x;                                     \
        unsigned x0 = cc, x1 = cc + 2;                                                                                 \
        asm volatile("v_mov_b32 v52, %[y]\n\tv_mov_b32 v53, %[y]\n\tv_mov_b32 v54, %[y]\n\tv_mov_b32 v55, %[y]\n\t"   \
                     "v_mov_b32 v56, %[y]\n\tv_mov_b32 v57, %[y]\n\tv_mov_b32 v58, %[y]\n\tv_mov_b32 v59, %[y]\n\t"   \
                     ".p2align 6\n1:\n\t" GROUP                                                                        \
                     "s_add_i32 %[cnt], %[cnt], -1\n\ts_cmp_lg_u32 %[cnt], 0\n\ts_cbranch_scc1 1b\n\t"                 \
                     "v_mov_b32 %[y], v55"                                                                             \
                     : [y] "+v"(y), [w0] "+v"(w0), [w1] "+v"(w1), [x0] "+v"(x0), [x1] "+v"(x1), [al] "+v"(al), [ah] "+v"(ah), \
                       [fl] "+v"(fl), [fh] "+v"(fh), [cnt] "+s"(ngroups)                                               \
                     : [c01l] "v"(c01l), [c01h] "v"(c01h), [c2l] "v"(c2l), [c2h] "v"(c2h), [cfbl] "v"(cfbl), [cfbh] "v"(cfbh), \
                       [k] "s"(8192)                                                                                   \
                     : "scc", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59");                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = w0 + w1 + al + ah + fl + fh + y + x0 + x1;                        \
    }
W_KERNEL(k_wser, WG_SER)
W_KERNEL(k_wret, WG_RET)
W_KERNEL(k_wshort, WG_SHORT)

// ---------------------------------------------------------------------------------------------- encoding-size probes
// 64 independent instructions per pass, four-byte (VOP2 e32) against eight-byte (VOP3 e64) encodings of the same add,
// on one wave per SIMD and on two (512-thread workgroups): is a lone wave bound by instruction BYTES?
#define REP16(X) X X X X X X X X X X X X X X X X
template <int KIND>
__global__ void k_enc(int *out, int ngroups, int cc)
{
    int a0 = cc, a1 = cc + 1, a2 = cc + 2, a3 = cc + 3, b = threadIdx.x;
    if (cc >= 100 && (threadIdx.x & 32)) return;          // cc >= 100: only the low 32 lanes of every wave stay active
    for (int g = 0; g < ngroups; ++g) {
        if (KIND == 4)
            asm volatile(REP16("v_add_u32_sdwa %0, %4, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\tv_add_u32_sdwa %1, %4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"
                               "v_add_u32_sdwa %2, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\tv_add_u32_sdwa %3, %4, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        if (KIND == 5)
            asm volatile(REP16("v_mad_i32_i24 %0, %4, %0, %0\n\tv_mad_i32_i24 %1, %4, %1, %1\n\tv_mad_i32_i24 %2, %4, %2, %2\n\tv_mad_i32_i24 %3, %4, %3, %3\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        if (KIND == 6)
            asm volatile(REP16("v_dot2_i32_i16 %0, %4, %0, %0\n\tv_dot2_i32_i16 %1, %4, %1, %1\n\tv_dot2_i32_i16 %2, %4, %2, %2\n\tv_dot2_i32_i16 %3, %4, %3, %3\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        if (KIND == 7)
            asm volatile(REP16("v_mov_b32_dpp %0, %4 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %4 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                               "v_mov_b32_dpp %2, %4 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %4 row_ror:1 row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        if (KIND == 0)
            asm volatile(REP16("v_add_u32_e32 %0, %4, %0\n\tv_add_u32_e32 %1, %4, %1\n\tv_add_u32_e32 %2, %4, %2\n\tv_add_u32_e32 %3, %4, %3\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        if (KIND == 1)
            asm volatile(REP16("v_add_u32_e64 %0, %4, %0\n\tv_add_u32_e64 %1, %4, %1\n\tv_add_u32_e64 %2, %4, %2\n\tv_add_u32_e64 %3, %4, %3\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        if (KIND == 2)
            asm volatile(REP16("v_add_u32_e32 %0, %4, %0\n\tv_add_u32_e64 %1, %4, %1\n\tv_add_u32_e32 %2, %4, %2\n\tv_add_u32_e64 %3, %4, %3\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        if (KIND == 3)
            asm volatile(REP16("v_mul_i32_i24_dpp %0, %4, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_mul_i32_i24_dpp %1, %4, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                               "v_mul_i32_i24_dpp %2, %4, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_mul_i32_i24_dpp %3, %4, %3 row_ror:1 row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

// ---------------------------------------------------------------------------------------------- v_dot2_i32_i16 probes
// 80 instructions per loop pass: (a) one dependent chain, (b) two interleaved chains, (c) ten independent accumulators
__global__ __launch_bounds__(256) void k_dot_dep(int *out, int ngroups, int cc)
{
    int a = cc; unsigned p = threadIdx.x * 0x10001u, c = 0x00030005u;
    for (int g = 0; g < ngroups; ++g) {
#pragma unroll
        for (int i = 0; i < 80; ++i) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a) : "v"(p), "v"(c));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ __launch_bounds__(256) void k_dot_two(int *out, int ngroups, int cc)
{
    int a = cc, b = cc + 1; unsigned p = threadIdx.x * 0x10001u, c = 0x00030005u;
    for (int g = 0; g < ngroups; ++g) {
#pragma unroll
        for (int i = 0; i < 40; ++i)
            asm volatile("v_dot2_i32_i16 %0, %2, %3, %0\n\tv_dot2_i32_i16 %1, %2, %3, %1" : "+v"(a), "+v"(b) : "v"(p), "v"(c));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b;
}
__global__ __launch_bounds__(256) void k_mad_dep(int *out, int ngroups, int cc)
{
    int a = cc, p = threadIdx.x, c = 5;
    for (int g = 0; g < ngroups; ++g) {
#pragma unroll
        for (int i = 0; i < 80; ++i) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a) : "v"(p), "v"(c));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ __launch_bounds__(256) void k_cvt_dot(int *out, int ngroups, int cc)     // cvt_pk -> dot2 -> ashr -> add chain only
{
    int a = cc, w = 3; unsigned p = threadIdx.x * 0x10001u, c = 0x00030005u;
    for (int g = 0; g < ngroups; ++g) {
#pragma unroll
        for (int i = 0; i < 20; ++i)
            asm volatile("v_dot2_i32_i16 %0, %2, %3, 0\n\tv_ashrrev_i32 %0, 14, %0\n\tv_add_u32 %1, %1, %0\n\tv_cvt_pk_i16_i32 %2, %1, %1"
                         : "+v"(a), "+v"(w), "+v"(p) : "v"(c));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + w + p;
}

int main()
{
    int *out;
    hipMalloc(&out, 512 * 1024 * 4);
    const int ngroups = 2049;      // 16 392 steps, as one frame
    struct K { const char *name; void (*fn)(int *, int, int); int instr_per_group; int steps_per_group; };
    const K ks[] = {
        {"Q7 7-instr, round 3 (H I B A C E F)", k_old7, 56, 8},
        {"Q7 7-instr retimed A (H I A E C F B)", k_new7a, 56, 8},
        {"Q7 7-instr retimed B (H I A E C B F)", k_new7b, 56, 8},
        {"Q7 7-instr retimed C (H A I E C B F)", k_new7c, 56, 8},
        {"Q7 7-instr retimed D (H C A I E F B)", k_new7d, 56, 8},
        {"Q7 9-instr, round 3 (H G B C A I D F E)", k_old9, 72, 8},
        {"Q7 9-instr retimed A", k_new9a, 72, 8},
        {"Q7 9-instr retimed B", k_new9b, 72, 8},
        {"wide Q2.14 10-instr, serial", k_wser, 80, 8},
        {"wide Q2.14 10-instr, feed-forward ahead", k_wret, 80, 8},
        {"Q7 7-instr retimed, VOP2 feedback products", k_new7s, 56, 8},
        {"Q7 7-instr round-3 order, VOP2 feedback", k_old7s, 56, 8},
        {"wide Q2.14 10-instr, serial, 4x v_dot2c", k_wshort, 80, 8},
        {"probe: 80 dependent v_dot2_i32_i16", k_dot_dep, 80, 8},
        {"probe: two chains of v_dot2_i32_i16", k_dot_two, 80, 8},
        {"probe: 80 dependent v_mad_i32_i24", k_mad_dep, 80, 8},
        {"probe: dot2 > ashr > add > cvt_pk chain", k_cvt_dot, 80, 8},
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 1500; ++rep) hipLaunchKernelGGL(ks[0].fn, dim3(256), dim3(256), 0, 0, out, ngroups, 3);   // leave the idle clocks
    hipDeviceSynchronize();
    for (const K &k : ks) {
        float best = 1e9f;
        for (int rep = 0; rep < 40; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k.fn, dim3(256), dim3(256), 0, 0, out, ngroups, 3);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-44s %7.1f us for 16 392 steps = %5.2f ns per step = %5.2f ns per instruction\n", k.name, best * 1e3,
               best * 1e6 / (ngroups * 8.0), best * 1e6 / (ngroups * (double)k.instr_per_group));
    }
    // the same blocks with TWO workgroups per CU: two waves on every SIMD (what a batch of 8 192 frames, or two frames per
    // wave instead of four, would look like)
    for (const K &k : ks) {
        float best = 1e9f;
        for (int rep = 0; rep < 20; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k.fn, dim3(512), dim3(256), 0, 0, out, ngroups, 3);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("2 waves/SIMD: %-44s %7.1f us for 16 392 steps = %5.2f ns per step per wave\n", k.name, best * 1e3,
               best * 1e6 / (ngroups * 8.0));
    }
    // encoding-size probes: ns per instruction and bytes per ns, one and two waves per SIMD
    struct E { const char *name; void (*fn)(int *, int, int); int bytes; };
    const E es[] = {{"64 x v_add_u32 e32 (4 bytes)", k_enc<0>, 4 * 64}, {"64 x v_add_u32 e64 (8 bytes)", k_enc<1>, 8 * 64},
                    {"64 x alternating e32 / e64", k_enc<2>, 6 * 64}, {"64 x v_mul_i32_i24_dpp (8 bytes)", k_enc<3>, 8 * 64},
                    {"64 x v_add_u32_sdwa (8 bytes)", k_enc<4>, 8 * 64}, {"64 x v_mad_i32_i24 (8 bytes)", k_enc<5>, 8 * 64},
                    {"64 x v_dot2_i32_i16 (8 bytes)", k_enc<6>, 8 * 64}, {"64 x v_mov_b32_dpp (8 bytes)", k_enc<7>, 8 * 64}};
    for (int half = 0; half < 2; ++half)
    for (int threads = 256; threads <= 1024; threads += 256)
        for (const E &e : es) {
            float best = 1e9f;
            if (half) printf("low 32 lanes only: ");
            for (int rep = 0; rep < 30; ++rep) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(e.fn, dim3(256), dim3(threads), 0, 0, out, 4096, half ? 103 : 3);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double ns = best * 1e6 / (4096.0 * 64.0);
            printf("%d wave(s) per SIMD, %-36s %5.2f ns per instruction per wave = %5.2f bytes per ns per wave\n", threads / 256, e.name, ns,
                   e.bytes / 64.0 / ns);
        }
    return 0;
}
