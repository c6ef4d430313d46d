// Micro-benchmark: rate at which ONE wave (alone on its SIMD) issues independent instructions of the kinds
// the Q15 cascade step is made of.  hipcc -O3 --offload-arch=gfx950 issue_rate.hip -o issue_rate && ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X X X X X X X X
template <int KIND>
__global__ void k(int *out, unsigned long long *cyc, int n, int c)
{
    int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, w = c;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        if (KIND == 0)
            asm volatile(REP8("v_mul_i32_i24 %0, %8, %8\n v_mul_i32_i24 %1, %8, %8\n v_mul_i32_i24 %2, %8, %8\n v_mul_i32_i24 %3, %8, %8\n"
                              "v_mul_i32_i24 %4, %8, %8\n v_mul_i32_i24 %5, %8, %8\n v_mul_i32_i24 %6, %8, %8\n v_mul_i32_i24 %7, %8, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));
        if (KIND == 1)
            asm volatile(REP8("v_mul_i32_i24_dpp %0, %8, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mul_i32_i24_dpp %1, %8, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mul_i32_i24_dpp %2, %8, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mul_i32_i24_dpp %3, %8, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mul_i32_i24_dpp %4, %8, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mul_i32_i24_dpp %5, %8, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mul_i32_i24_dpp %6, %8, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mul_i32_i24_dpp %7, %8, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));
        if (KIND == 2)
            asm volatile(REP8("v_add_u32_sdwa %0, %8, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n v_add_u32_sdwa %1, %8, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n"
                              "v_add_u32_sdwa %2, %8, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n v_add_u32_sdwa %3, %8, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n"
                              "v_add_u32_sdwa %4, %8, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n v_add_u32_sdwa %5, %8, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n"
                              "v_add_u32_sdwa %6, %8, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n v_add_u32_sdwa %7, %8, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));
        if (KIND == 3)
            asm volatile(REP8("v_mad_i32_i24 %0, %8, %8, %8\n v_mad_i32_i24 %1, %8, %8, %8\n v_mad_i32_i24 %2, %8, %8, %8\n v_mad_i32_i24 %3, %8, %8, %8\n"
                              "v_mad_i32_i24 %4, %8, %8, %8\n v_mad_i32_i24 %5, %8, %8, %8\n v_mad_i32_i24 %6, %8, %8, %8\n v_mad_i32_i24 %7, %8, %8, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));
        if (KIND == 4)
            asm volatile(REP8("v_add_u32 %0, %8, %8\n v_add_u32 %1, %8, %8\n v_add_u32 %2, %8, %8\n v_add_u32 %3, %8, %8\n"
                              "v_add_u32 %4, %8, %8\n v_add_u32 %5, %8, %8\n v_add_u32 %6, %8, %8\n v_add_u32 %7, %8, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));
        if (KIND == 5)
            asm volatile(REP8("v_pk_add_u16 %0, %8, %8\n v_pk_add_u16 %1, %8, %8\n v_pk_add_u16 %2, %8, %8\n v_pk_add_u16 %3, %8, %8\n"
                              "v_pk_add_u16 %4, %8, %8\n v_pk_add_u16 %5, %8, %8\n v_pk_add_u16 %6, %8, %8\n v_pk_add_u16 %7, %8, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    int *out; unsigned long long *cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8 * 4096);
    const int n = 2000;
    const char *names[6] = {"v_mul_i32_i24", "v_mul_i32_i24_dpp row_ror:1", "v_add_u32_sdwa (high words)", "v_mad_i32_i24", "v_add_u32", "v_pk_add_u16"};
    void (*ks[6])(int *, unsigned long long *, int, int) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>};
    for (int kind = 0; kind < 6; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(ks[kind], dim3(1), dim3(256), 0, 0, out, cyc, n, 3);     // 4 waves = one per SIMD of one CU
            hipDeviceSynchronize();
        }
        unsigned long long c;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("one wave per SIMD, independent %-32s %5.2f cycles per instruction\n", names[kind], (double)c / (n * 64.0));
    }
    return 0;
}
