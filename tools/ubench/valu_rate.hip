// Micro-benchmark, round 4: pass rate of the integer vector instructions the Q15 kernels are made of, per SIMD, at 1, 2, 4
// and 8 waves per SIMD (256 workgroups x 256..2048 threads... two workgroups of 1024 for 8).  64 independent instructions per
// loop pass (four accumulators), so the figure is issue / pass rate and not latency.  Printed: ns per wave-instruction per
// SIMD (aggregate: launch time / instructions each SIMD executed).
//   hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X X X X X X X X X X X X X X X X
#define OPK(NAME, I0, I1, I2, I3)                                                                                        \
    __global__ __launch_bounds__(1024) void NAME(int *out, int n, int cc)                                                \
    {                                                                                                                    \
        int a0 = cc, a1 = cc + 1, a2 = cc + 2, a3 = cc + 3, b = threadIdx.x, c = threadIdx.x * 3 + 1;                    \
        for (int g = 0; g < n; ++g)                                                                                      \
            asm volatile(REP16(I0 "\n\t" I1 "\n\t" I2 "\n\t" I3 "\n\t")                                                  \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;                                                  \
    }
#define OP2(NAME, OP) OPK(NAME, OP " %0, %4, %0", OP " %1, %4, %1", OP " %2, %4, %2", OP " %3, %4, %3")
#define OP3(NAME, OP) OPK(NAME, OP " %0, %4, %5, %0", OP " %1, %4, %5, %1", OP " %2, %4, %5, %2", OP " %3, %4, %5, %3")
#define OP2X(NAME, OP, SFX) OPK(NAME, OP " %0, %4, %0 " SFX, OP " %1, %4, %1 " SFX, OP " %2, %4, %2 " SFX, OP " %3, %4, %3 " SFX)
#define OP3X(NAME, OP, SFX)                                                                                              \
    OPK(NAME, OP " %0, %4, %5, %0 " SFX, OP " %1, %4, %5, %1 " SFX, OP " %2, %4, %5, %2 " SFX, OP " %3, %4, %5, %3 " SFX)

OP2(k_add, "v_add_u32_e32")
OP2(k_sub, "v_sub_u32_e32")
OP2(k_ashr, "v_ashrrev_i32_e32")
OP2(k_lshl, "v_lshlrev_b32_e32")
OP2(k_and, "v_and_b32_e32")
OP2(k_xor, "v_xor_b32_e32")
OP2(k_max, "v_max_i32_e32")
OP2(k_mul24, "v_mul_i32_i24_e32")
OP2(k_mulhi24, "v_mul_hi_i32_i24_e32")
OP2(k_dot2c, "v_dot2c_i32_i16_e32")
OP2(k_cvtpk, "v_cvt_pk_i16_i32")
OP2(k_pack, "v_pack_b32_f16")
OP2(k_mullo, "v_mul_lo_u32")
OP3(k_bfe, "v_bfe_i32")
OP3(k_bfi, "v_bfi_b32")
OP3(k_perm, "v_perm_b32")
OP3(k_alignbit, "v_alignbit_b32")
OP3(k_add3, "v_add3_u32")
OP3(k_lshladd, "v_lshl_add_u32")
OP3(k_addlshl, "v_add_lshl_u32")
OP3(k_andor, "v_and_or_b32")
OP3(k_lshlor, "v_lshl_or_b32")
OP3(k_med3, "v_med3_i32")
OP3(k_mad24, "v_mad_i32_i24")
OP3(k_madi16, "v_mad_i32_i16")
OP3X(k_madi16_sel, "v_mad_i32_i16", "op_sel:[1,0,0]")
OP3(k_dot2, "v_dot2_i32_i16")
OP3(k_dot4, "v_dot4_i32_i8")
OP3(k_sad16, "v_sad_u16")
OP3(k_fma, "v_fma_f32")
OP2(k_pkadd, "v_pk_add_i16")
OP2(k_pksub, "v_pk_sub_i16")
OP2X(k_pkadd_sel, "v_pk_add_i16", "op_sel:[1,0] op_sel_hi:[0,1]")
OP2(k_pkashr, "v_pk_ashrrev_i16")
OP2(k_pklshl, "v_pk_lshlrev_b16")
OP2(k_pkmax, "v_pk_max_i16")
OP2(k_pkmullo, "v_pk_mul_lo_u16")
OP3(k_pkmad, "v_pk_mad_i16")
OP2X(k_add_sdwa, "v_add_u32_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0")
OP2X(k_ashr_sdwa, "v_ashrrev_i32_sdwa", "dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD")
OP2X(k_mul24_sdwa, "v_mul_i32_i24_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0")
OP2X(k_add_dpp, "v_add_u32_dpp", "row_ror:1 row_mask:0xf bank_mask:0xf")
OPK(k_mov_dpp, "v_mov_b32_dpp %0, %4 row_ror:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %1, %4 row_ror:1 row_mask:0xf bank_mask:0xf",
    "v_mov_b32_dpp %2, %4 row_ror:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %3, %4 row_ror:1 row_mask:0xf bank_mask:0xf")
OPK(k_mov, "v_mov_b32_e32 %0, %4", "v_mov_b32_e32 %1, %5", "v_mov_b32_e32 %2, %4", "v_mov_b32_e32 %3, %5")

int main()
{
    int *out;
    (void)hipMalloc(&out, 512 * 1024 * 4);
    struct K { const char *name; void (*fn)(int *, int, int); };
#define E(k) {#k, k}
    const K ks[] = {E(k_add), E(k_sub), E(k_ashr), E(k_lshl), E(k_and), E(k_xor), E(k_max), E(k_mov), E(k_mul24), E(k_mulhi24), E(k_dot2c),
                    E(k_cvtpk), E(k_pack), E(k_mullo), E(k_bfe), E(k_bfi), E(k_perm), E(k_alignbit), E(k_add3), E(k_lshladd), E(k_addlshl),
                    E(k_andor), E(k_lshlor), E(k_med3), E(k_mad24), E(k_madi16), E(k_madi16_sel), E(k_dot2), E(k_dot4),
                    E(k_sad16), E(k_fma), E(k_pkadd), E(k_pksub), E(k_pkadd_sel), E(k_pkashr), E(k_pklshl), E(k_pkmax),
                    E(k_pkmullo), E(k_pkmad), E(k_add_sdwa), E(k_ashr_sdwa), E(k_mul24_sdwa), E(k_add_dpp), E(k_mov_dpp)};
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int n = 2048;
    for (int rep = 0; rep < 300; ++rep) hipLaunchKernelGGL(k_add, dim3(512), dim3(1024), 0, 0, out, n, 3);
    (void)hipDeviceSynchronize();
    printf("%-14s %8s %8s %8s %8s   ns per wave-instruction per SIMD at 1 / 2 / 4 / 8 waves per SIMD\n", "", "w1", "w2", "w4", "w8");
    for (const K &k : ks) {
        printf("%-14s", k.name + 2);
        const int grid[4] = {256, 256, 256, 512}, thr[4] = {256, 512, 1024, 1024}, wps[4] = {1, 2, 4, 8};
        for (int m = 0; m < 4; ++m) {
            float best = 1e9f;
            for (int rep = 0; rep < 12; ++rep) {
                (void)hipEventRecord(e0, 0);
                hipLaunchKernelGGL(k.fn, dim3(grid[m]), dim3(thr[m]), 0, 0, out, n, 3);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf(" %8.2f", best * 1e6 / (n * 64.0 * wps[m]));
        }
        printf("\n");
    }
    return 0;
}
