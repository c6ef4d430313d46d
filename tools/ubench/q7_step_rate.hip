// Micro-benchmark: wall-clock time per step of the Q15 cascade's inner loop (chain_q15.hip, SA_Q7_STEP) with one
// wave per SIMD on every SIMD of the chip (256 workgroups x 4 waves), against variants of the same nine
// instructions.  No LDS, no refill, no stores: what is left is the floor of the recursion itself.
//   hipcc -O3 --offload-arch=gfx950 q7_step_rate.hip -o q7_step_rate && ./q7_step_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define STEP_A(Y, H1, H2)                                                                                              \
    "v_mad_i32_i24 %[p4], %[" H1 "], %[nA1], %[k]\n\t"                                                                 \
    "v_mul_i32_i24_dpp %[p2], %[" H2 "], %[cB0] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"                              \
    "v_mul_i32_i24_dpp %[p0], %[" H1 "], %[cB2] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"                              \
    "v_mul_i32_i24_dpp %[p1], %[" H1 "], %[cB1] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"                              \
    "v_mad_i32_i24 %[p3], %[" H1 "], %[nA0], %[k]\n\t"                                                                 \
    "v_add_u32_sdwa %[t], %[p0], %[p4] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"        \
    "v_add_u32_sdwa %[u], %[p1], %[p2] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"        \
    "v_add_u32_sdwa %[" Y "], %[s2], %[t] dst_sel:WORD_0 dst_unused:UNUSED_SEXT src0_sel:DWORD src1_sel:DWORD\n\t"     \
    "v_add_u32_sdwa %[s2], %[u], %[p3] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"

// same data flow, no SDWA / DPP modifiers at all (NOT the same arithmetic: a probe of what the modifiers cost)
#define STEP_PLAIN(Y, H1, H2)                                                                                          \
    "v_mad_i32_i24 %[p4], %[" H1 "], %[nA1], %[k]\n\t"                                                                 \
    "v_mul_i32_i24 %[p2], %[" H2 "], %[cB0]\n\t"                                                                       \
    "v_mul_i32_i24 %[p0], %[" H1 "], %[cB2]\n\t"                                                                       \
    "v_mul_i32_i24 %[p1], %[" H1 "], %[cB1]\n\t"                                                                       \
    "v_mad_i32_i24 %[p3], %[" H1 "], %[nA0], %[k]\n\t"                                                                 \
    "v_add_u32 %[t], %[p0], %[p4]\n\t"                                                                                 \
    "v_add_u32 %[u], %[p1], %[p2]\n\t"                                                                                 \
    "v_add_u32 %[" Y "], %[s2], %[t]\n\t"                                                                              \
    "v_add_u32 %[s2], %[u], %[p3]"

// DPP kept, SDWA replaced by plain adds
#define STEP_DPP(Y, H1, H2)                                                                                            \
    "v_mad_i32_i24 %[p4], %[" H1 "], %[nA1], %[k]\n\t"                                                                 \
    "v_mul_i32_i24_dpp %[p2], %[" H2 "], %[cB0] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"                              \
    "v_mul_i32_i24_dpp %[p0], %[" H1 "], %[cB2] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"                              \
    "v_mul_i32_i24_dpp %[p1], %[" H1 "], %[cB1] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"                              \
    "v_mad_i32_i24 %[p3], %[" H1 "], %[nA0], %[k]\n\t"                                                                 \
    "v_add_u32 %[t], %[p0], %[p4]\n\t"                                                                                 \
    "v_add_u32 %[u], %[p1], %[p2]\n\t"                                                                                 \
    "v_add_u32 %[" Y "], %[s2], %[t]\n\t"                                                                              \
    "v_add_u32 %[s2], %[u], %[p3]"

// SDWA kept, DPP replaced by plain multiplies
#define STEP_SDWA(Y, H1, H2)                                                                                           \
    "v_mad_i32_i24 %[p4], %[" H1 "], %[nA1], %[k]\n\t"                                                                 \
    "v_mul_i32_i24 %[p2], %[" H2 "], %[cB0]\n\t"                                                                       \
    "v_mul_i32_i24 %[p0], %[" H1 "], %[cB2]\n\t"                                                                       \
    "v_mul_i32_i24 %[p1], %[" H1 "], %[cB1]\n\t"                                                                       \
    "v_mad_i32_i24 %[p3], %[" H1 "], %[nA0], %[k]\n\t"                                                                 \
    "v_add_u32_sdwa %[t], %[p0], %[p4] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"        \
    "v_add_u32_sdwa %[u], %[p1], %[p2] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"        \
    "v_add_u32_sdwa %[" Y "], %[s2], %[t] dst_sel:WORD_0 dst_unused:UNUSED_SEXT src0_sel:DWORD src1_sel:DWORD\n\t"     \
    "v_add_u32_sdwa %[s2], %[u], %[p3] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"

// the Y write as a DWORD add followed by nothing (probe: is the WORD_0 + sign-extend destination form the slow one?)
#define STEP_YDW(Y, H1, H2)                                                                                            \
    "v_mad_i32_i24 %[p4], %[" H1 "], %[nA1], %[k]\n\t"                                                                 \
    "v_mul_i32_i24_dpp %[p2], %[" H2 "], %[cB0] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"                              \
    "v_mul_i32_i24_dpp %[p0], %[" H1 "], %[cB2] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"                              \
    "v_mul_i32_i24_dpp %[p1], %[" H1 "], %[cB1] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"                              \
    "v_mad_i32_i24 %[p3], %[" H1 "], %[nA0], %[k]\n\t"                                                                 \
    "v_add_u32_sdwa %[t], %[p0], %[p4] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"        \
    "v_add_u32_sdwa %[u], %[p1], %[p2] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"        \
    "v_add_u32 %[" Y "], %[s2], %[t]\n\t"                                                                              \
    "v_add_u32_sdwa %[s2], %[u], %[p3] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"

#define OPERANDS(YV)                                                                                                   \
    : [y0] "=&v"(YV), [s2] "+v"(s2), [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2), [p3] "=&v"(p3), [p4] "=&v"(p4),   \
      [t] "=&v"(t), [u] "=&v"(u)                                                                                       \
    : [y7] "v"(h1), [y6] "v"(h2), [cB2] "v"(cB2), [cB1] "v"(cB1), [cB0] "v"(cB0), [nA0] "v"(nA0), [nA1] "v"(nA1), [k] "s"(k127)

template <int KIND>
__device__ __forceinline__ void step(int &y, int &s2, int h1, int h2, int cB2, int cB1, int cB0, int nA0, int nA1, int k127)
{
    int p0, p1, p2, p3, p4, t, u, y0;
    if (KIND == 0) asm volatile(STEP_A("y0", "y7", "y6") OPERANDS(y0));
    if (KIND == 1) asm volatile(STEP_PLAIN("y0", "y7", "y6") OPERANDS(y0));
    if (KIND == 2) asm volatile(STEP_DPP("y0", "y7", "y6") OPERANDS(y0));
    if (KIND == 3) asm volatile(STEP_SDWA("y0", "y7", "y6") OPERANDS(y0));
    if (KIND == 4) asm volatile(STEP_YDW("y0", "y7", "y6") OPERANDS(y0));
    y = y0;
}

template <int KIND>
__global__ __launch_bounds__(256) void k(int *out, int ngroups, int c)
{
    int y[8];
    for (int e = 0; e < 8; ++e) y[e] = threadIdx.x + e;
    int s2 = c, cB2 = c << 9, cB1 = (c + 1) << 9, cB0 = (c + 2) << 9, nA0 = -(c << 9), nA1 = -((c + 3) << 9);
    const int k127 = 127 << 9;
    for (int g = 0; g < ngroups; ++g) {
#pragma unroll
        for (int e = 0; e < 8; ++e) step<KIND>(y[e], s2, y[(e + 7) & 7], y[(e + 6) & 7], cB2, cB1, cB0, nA0, nA1, k127);
    }
    int acc = s2;
    for (int e = 0; e < 8; ++e) acc += y[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}


// ---- the same nine instructions in other cyclic orders (block = H first: y[n] from the prepared s2 and t) ------------
// instruction names: A p4 = y*nA1 + k | B p2 = x[n-2]*B0 | C p0 = x*B2 | D p1 = x*B1 | E p3 = y*nA0 + k | F t = hi p0 + hi p4
//                    G u = hi p1 + hi p2 | H y = s2 + t (word, sign-extended) | I s2 = u + hi p3
#define I_A(H1) "v_mad_i32_i24 %[p4], %[" H1 "], %[nA1], %[k]\n\t"
#define I_B(H2) "v_mul_i32_i24_dpp %[p2], %[" H2 "], %[cB0] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_C(H1) "v_mul_i32_i24_dpp %[p0], %[" H1 "], %[cB2] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_D(H1) "v_mul_i32_i24_dpp %[p1], %[" H1 "], %[cB1] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_E(H1) "v_mad_i32_i24 %[p3], %[" H1 "], %[nA0], %[k]\n\t"
#define I_F "v_add_u32_sdwa %[t], %[p0], %[p4] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"
#define I_G "v_add_u32_sdwa %[u], %[p1], %[p2] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"
#define I_H(Y) "v_add_u32_sdwa %[" Y "], %[s2], %[t] dst_sel:WORD_0 dst_unused:UNUSED_SEXT src0_sel:DWORD src1_sel:DWORD\n\t"
#define I_HP(Y) "v_add_u32 %[" Y "], %[s2], %[t]\n\t"
#define I_I "v_add_u32_sdwa %[s2], %[u], %[p3] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
// ten-instruction form: y written by a plain add, sign-extended off the critical cycle (S), own-lane products take
// the low word sign-extended through SDWA (no rounding constant: the feedback terms are subtracted)
#define I_S(Y) "v_bfe_i32 %[ys], %[" Y "], 0, 16\n\t"
#define I_A2(Y) "v_mul_i32_i24_sdwa %[p4], sext(%[" Y "]), %[nA1] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\t"
#define I_E2(Y) "v_mul_i32_i24_sdwa %[p3], sext(%[" Y "]), %[nA0] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\t"
#define I_F2 "v_sub_u32_sdwa %[t], %[p0], %[p4] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"
#define I_I2 "v_sub_u32_sdwa %[s2], %[u], %[p3] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"

#define ROPERANDS                                                                                                      \
    : [y0] "=&v"(y0), [s2] "+v"(s2), [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [p4] "+v"(p4),        \
      [t] "+v"(t), [u] "+v"(u)                                                                                         \
    : [y7] "v"(h1), [cB2] "v"(cB2), [cB1] "v"(cB1), [cB0] "v"(cB0), [nA0] "v"(nA0), [nA1] "v"(nA1), [k] "s"(k127)

struct Carry { int p0, p1, p2, p3, p4, t, u; };

// block: y0 = new output (H first), h1 = the output before it (x[n-2] source for B)
template <int KIND>
__device__ __forceinline__ void rstep(int &y, int &s2, Carry &c, int h1, int cB2, int cB1, int cB0, int nA0, int nA1, int k127)
{
    int y0, p0 = c.p0, p1 = c.p1, p2 = c.p2, p3 = c.p3, p4 = c.p4, t = c.t, u = c.u;
    if (KIND == 5) asm volatile(I_H("y0") I_G I_B("y7") I_C("y0") I_A("y0") I_I I_D("y0") I_F I_E("y0") ROPERANDS);     // F->H 2
    if (KIND == 6) asm volatile(I_H("y0") I_G I_B("y7") I_C("y0") I_A("y0") I_I I_F I_D("y0") I_E("y0") ROPERANDS);     // A->F 2
    if (KIND == 7) asm volatile(I_HP("y0") I_G I_B("y7") I_C("y0") I_A("y0") I_I I_D("y0") I_F I_E("y0") ROPERANDS);    // 5 with a plain y
    if (KIND == 8) asm volatile(I_H("y0") I_B("y7") I_G I_I I_C("y0") I_A("y0") I_D("y0") I_F I_E("y0") ROPERANDS);     // H->C 4, H->A 5, G->I 1
    y = y0; c.p0 = p0; c.p1 = p1; c.p2 = p2; c.p3 = p3; c.p4 = p4; c.t = t; c.u = u;
}

// ten-instruction block; ys = sign-extended copy read by the neighbours one step later than in the nine-instruction form
__device__ __forceinline__ void tstep(int &yw, int &ysn, int &s2, Carry &c, int ys1, int ys2, int cB2, int cB1, int cB0, int nA0, int nA1)
{
    int p0 = c.p0, p1 = c.p1, p2 = c.p2, p3 = c.p3, p4 = c.p4, t = c.t, u = c.u, ys;
    asm volatile(I_HP("y0") I_G "v_mul_i32_i24_dpp %[p0], %[x1], %[cB2] row_ror:1 row_mask:0xf bank_mask:0xf\n\t" I_A2("y0")
                 "v_mul_i32_i24_dpp %[p1], %[x1], %[cB1] row_ror:1 row_mask:0xf bank_mask:0xf\n\t" I_I2 I_F2
                 "v_mul_i32_i24_dpp %[p2], %[x2], %[cB0] row_ror:1 row_mask:0xf bank_mask:0xf\n\t" I_S("y0") I_E2("y0")
                 : [y0] "+v"(yw), [ys] "=&v"(ys), [s2] "+v"(s2), [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [p4] "+v"(p4),
                   [t] "+v"(t), [u] "+v"(u)
                 : [x1] "v"(ys1), [x2] "v"(ys2), [cB2] "v"(cB2), [cB1] "v"(cB1), [cB0] "v"(cB0), [nA0] "v"(nA0), [nA1] "v"(nA1));
    ysn = ys; c.p0 = p0; c.p1 = p1; c.p2 = p2; c.p3 = p3; c.p4 = p4; c.t = t; c.u = u;
}

template <int KIND>
__global__ __launch_bounds__(256) void kr(int *out, int ngroups, int cc)
{
    int y[8];
    for (int e = 0; e < 8; ++e) y[e] = threadIdx.x + e;
    int s2 = cc, cB2 = cc << 9, cB1 = (cc + 1) << 9, cB0 = (cc + 2) << 9, nA0 = -(cc << 9), nA1 = -((cc + 3) << 9);
    const int k127 = 127 << 9;
    Carry c = {1, 2, 3, 4, 5, 6, 7};
    int yw = 3;
    for (int g = 0; g < ngroups; ++g) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (KIND == 9) tstep(yw, y[e], s2, c, y[(e + 7) & 7], y[(e + 6) & 7], cB2, cB1, cB0, nA0, nA1);
            else rstep<KIND>(y[e], s2, c, y[(e + 7) & 7], cB2, cB1, cB0, nA0, nA1, k127);
        }
    }
    int acc = s2 + c.p0 + c.p1 + c.p2 + c.p3 + c.p4 + c.t + c.u + yw;
    for (int e = 0; e < 8; ++e) acc += y[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// eight blocks of the order H G B C A I D F E in ONE asm statement (no compiler padding between blocks)
#define BLK(Y, H1) I_H(Y) I_G I_B(H1) I_C(Y) I_A(Y) I_I I_D(Y) I_F I_E(Y)
__global__ __launch_bounds__(256) void kmerged(int *out, int ngroups, int cc)
{
    int y0 = threadIdx.x, y1 = y0 + 1, y2 = y0 + 2, y3 = y0 + 3, y4 = y0 + 4, y5 = y0 + 5, y6 = y0 + 6, y7 = y0 + 7;
    int s2 = cc, cB2 = cc << 9, cB1 = (cc + 1) << 9, cB0 = (cc + 2) << 9, nA0 = -(cc << 9), nA1 = -((cc + 3) << 9);
    const int k127 = 127 << 9;
    int p0 = 1, p1 = 2, p2 = 3, p3 = 4, p4 = 5, t = 6, u = 7;
    for (int g = 0; g < ngroups; ++g) {
        asm volatile(BLK("y0", "y7") BLK("y1", "y0") BLK("y2", "y1") BLK("y3", "y2") BLK("y4", "y3") BLK("y5", "y4") BLK("y6", "y5") BLK("y7", "y6")
                     : [y0] "+v"(y0), [y1] "+v"(y1), [y2] "+v"(y2), [y3] "+v"(y3), [y4] "+v"(y4), [y5] "+v"(y5), [y6] "+v"(y6), [y7] "+v"(y7),
                       [s2] "+v"(s2), [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [p4] "+v"(p4), [t] "+v"(t), [u] "+v"(u)
                     : [cB2] "v"(cB2), [cB1] "v"(cB1), [cB0] "v"(cB0), [nA0] "v"(nA0), [nA1] "v"(nA1), [k] "s"(k127));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s2 + p0 + p1 + p2 + p3 + p4 + t + u + y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7;
}

int main()
{
    int *out;
    hipMalloc(&out, 256 * 256 * 4);
    const int ngroups = 2049;      // 16 392 steps, as one frame
    const int NK = 11;
    const char *names[NK] = {"SA_Q7_STEP as shipped", "no DPP, no SDWA (data flow only)", "DPP, plain adds", "SDWA, plain multiplies", "DPP + SDWA, y written as DWORD",
                             "order H G B C A I D F E", "order H G B C A I F D E", "order H G B C A I D F E, plain y", "order H B G I C A D F E",
                             "ten instructions (plain y + bfe)", "order H G B C A I D F E, 8 blocks in one asm"};
    void (*ks[NK])(int *, int, int) = {k<0>, k<1>, k<2>, k<3>, k<4>, kr<5>, kr<6>, kr<7>, kr<8>, kr<9>, kmerged};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // the chip leaves its idle clocks only after ~100 ms of load
    for (int rep = 0; rep < 1500; ++rep) hipLaunchKernelGGL(ks[0], dim3(256), dim3(256), 0, 0, out, ngroups, 3);
    hipDeviceSynchronize();
    for (int round = 0; round < 1; ++round)
    for (int kind = 0; kind < NK; ++kind) {
        float best = 1e9f;
        for (int rep = 0; rep < 50; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(ks[kind], dim3(256), dim3(256), 0, 0, out, ngroups, 3);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-36s %7.1f us for 16 392 steps = %5.2f ns per step = %5.2f ns per instruction\n", names[kind], best * 1e3,
               best * 1e6 / (ngroups * 8.0), best * 1e6 / (ngroups * (kind == 9 ? 80.0 : 72.0)));
    }
    return 0;
}
