// Micro-benchmark: cycles per dependent instruction for the ops on the Q15 IIR critical chain
// (one wave per SIMD).  hipcc --offload-arch=gfx950 chain_latency.hip -o chain_latency && ./chain_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ void k(int *out, unsigned long long *cyc, int n, int c)
{
    int v = threadIdx.x, w = c;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(v) : "v"(w));
            if (KIND == 1) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n" : "+v"(v));
            if (KIND == 2) asm volatile("v_mad_i32_i24 %0, %0, %1, %1\n v_lshrrev_b32 %0, 7, %0\n v_add3_u32 %0, %0, %1, %1\n v_bfe_i32 %0, %0, 0, 16\n s_nop 1\n v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v) : "v"(w));
            if (KIND == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(w));
            if (KIND == 4) asm volatile("v_lshrrev_b32 %0, 7, %0\n v_add3_u32 %0, %0, %1, %1" : "+v"(v) : "v"(w));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    int *out; unsigned long long *cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8 * 4096);
    const int n = 2000;
    const char *names[5] = {"v_mad_i32_i24 chain", "v_mov_dpp chain", "mad+lshr+add3+bfe+dpp (5 ops + s_nop)", "v_add_u32 chain", "lshr+add3 (2 ops)"};
    for (int waves = 1; waves <= 2; ++waves)
        for (int kind = 0; kind < 5; ++kind) {
            dim3 g(1), b(64 * 4 * waves);   // 4*waves waves in one workgroup -> `waves` per SIMD of one CU
            for (int rep = 0; rep < 2; ++rep) {
                if (kind == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, out, cyc, n, 3);
                if (kind == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, out, cyc, n, 3);
                if (kind == 2) hipLaunchKernelGGL(k<2>, g, b, 0, 0, out, cyc, n, 3);
                if (kind == 3) hipLaunchKernelGGL(k<3>, g, b, 0, 0, out, cyc, n, 3);
                if (kind == 4) hipLaunchKernelGGL(k<4>, g, b, 0, 0, out, cyc, n, 3);
                hipDeviceSynchronize();
            }
            unsigned long long c;
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const int ops = (kind == 2) ? 5 : (kind == 4 ? 2 : 1);
            printf("%d wave/SIMD  %-42s %6.2f s_memtime ticks per instruction\n", waves, names[kind], (double)c / (n * 16.0 * ops));
        }
    return 0;
}
