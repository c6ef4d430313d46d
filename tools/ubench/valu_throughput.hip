// Micro-benchmark: what a SIMD of gfx950 sustains per cycle for plain and packed fp32 VALU instructions as a
// function of the waves that share it -- the number the 8-waves-per-SIMD form of the float kernel is designed on.
// Every wave runs ILP independent accumulators (ILP = 1: one dependent chain, ILP = 8: issue-bound); the launch
// fills every CU with `waves` waves per SIMD; time = slowest wave (all waves stamp, the host takes the maximum)
// and wall time by events.  The clock is read in the kernel: s_memtime (shader cycles) over s_memrealtime (100 MHz).
// hipcc -O3 --offload-arch=gfx950 valu_throughput.hip -o valu_throughput && ./valu_throughput
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float v2f __attribute__((ext_vector_type(2)));

// KIND 0 v_fma_f32   1 v_pk_fma_f32   2 v_add_f32   3 v_pk_add_f32   4 v_mul_f32 (VOP2)   5 v_pk_mul_f32
//      6 v_fma_f32 with an SGPR operand   7 v_pk_fma_f32 with an SGPR pair operand (op_sel_hi broadcast)
template <int KIND, int ILP>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *cyc, unsigned long long *rt, int n, float c0, float c1)
{
    v2f a[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) a[i] = v2f{threadIdx.x * 1e-3f + i, 0.5f + i};
    const v2f b = {c0, c1};
    const float bs = c0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i].x) : "v"(b.x));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
                if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
                if (KIND == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 6) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i].x) : "s"(bs));
                if (KIND == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "s"(b));
                if (KIND == 8) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].x) : "s"(bs), "v"(a[i].y));       // VOP2, SGPR src0
                if (KIND == 9) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(a[i].y));      // VOP2, VGPRs
                if (KIND == 10) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "s"(bs), "v"(a[i].y));   // VOP3, SGPR src0, 3 distinct regs
                if (KIND == 11) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b.x), "v"(a[i].y));  // VOP3, VGPRs, 3 distinct regs
                if (KIND == 12) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i].x) : "s"(bs));                    // VOP2 mul by SGPR
                if (KIND == 13) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i].x) : "s"(bs));                    // VOP2 add SGPR
                if (KIND == 14) asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i].x) : "v"(b.x));
                if (KIND == 15) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i].x) : "v"(a[i].y));
                if (KIND == 16) asm volatile("v_fmac_f32 %0, 0x3f7fbe77, %1" : "+v"(a[i].x) : "v"(a[i].y));          // literal constant
                if (KIND == 17) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(a[i].y));  // VOP3 all VGPR distinct
                if (KIND == 18) { asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[(i + 1) % ILP].x) : "v"(b.x)); }
                if (KIND == 19) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i].x));
                if (KIND == 20) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "s"(bs));      // SGPR as addend
                if (KIND == 21) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].x) : "v"(a[i].y), "v"(a[i].y));  // VOP2 same src
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    v2f r = {0, 0};
#pragma unroll
    for (int i = 0; i < ILP; ++i) r += a[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r.x + r.y;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        cyc[w] = t1 - t0;
        rt[w] = r1 - r0;
    }
}

template <int KIND, int ILP>
void run(const char *name, int waves, float *out, unsigned long long *cyc, unsigned long long *rt)
{
    // `waves` waves per SIMD: 256-thread workgroups (one wave per SIMD each), `waves` of them per CU
    const int n = 4000, cus = 256, grid = cus * waves;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, ILP>), dim3(grid), dim3(256), 0, 0, out, cyc, rt, n, 0.999f, 1e-3f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> c(grid * 4), r(grid * 4);
    hipMemcpy(c.data(), cyc, 8 * c.size(), hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), rt, 8 * r.size(), hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    double clk = 0;
    for (size_t i = 0; i < r.size(); ++i) clk += 0;
    const double cmed = (double)c[c.size() / 2], cmax = (double)c.back();
    const double ghz = cmed / ((double)r[r.size() / 2] * 10.0);      // realtime tick = 10 ns
    const double instr = (double)n * 8 * ILP;                          // per wave
    // cycles of one SIMD per instruction = median wave time * (1 / (instr * waves sharing the SIMD))
    // the launch ends with its slowest wave: SIMD time per instruction from the LAST wave's end = max cycles / all instructions of the SIMD
    printf("%-30s %d waves/SIMD: %5.2f SIMD-cyc/instr (slowest wave %8.0f cyc, median %8.0f)  wall %7.1f us = %5.3f ns/instr/SIMD  clk ~%.2f GHz\n",
           name, waves, cmax / (instr * waves), cmax, cmed, ms * 1e3, ms * 1e6 / (instr * waves), ghz);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main()
{
    float *out;
    unsigned long long *cyc, *rt;
    hipMalloc(&out, (size_t)256 * 8 * 256 * 4);
    hipMalloc(&cyc, 8 * 256 * 8 * 4);
    hipMalloc(&rt, 8 * 256 * 8 * 4);
    for (int waves = 2; waves <= 8; waves *= 2) {
        run<0, 8>("v_fma_f32 v,v,v(dup)", waves, out, cyc, rt);
        run<17, 8>("v_fma_f32 vvv distinct", waves, out, cyc, rt);
        run<11, 8>("v_fma_f32 d+=v*v", waves, out, cyc, rt);
        run<10, 8>("v_fma_f32 d+=s*v", waves, out, cyc, rt);
        run<20, 8>("v_fma_f32 d=d*v+s", waves, out, cyc, rt);
        run<9, 8>("v_fmac_f32 d+=v*v", waves, out, cyc, rt);
        run<8, 8>("v_fmac_f32 d+=s*v", waves, out, cyc, rt);
        run<16, 8>("v_fmac_f32 d+=lit*v", waves, out, cyc, rt);
        run<21, 8>("v_fmac_f32 d+=v*v same", waves, out, cyc, rt);
        run<12, 8>("v_mul_f32 d=s*d", waves, out, cyc, rt);
        run<13, 8>("v_add_f32 d=s+d", waves, out, cyc, rt);
        run<14, 8>("v_add_f32_dpp row_shr", waves, out, cyc, rt);
        run<15, 8>("v_mov_b32_dpp row_shr", waves, out, cyc, rt);
        run<1, 8>("v_pk_fma_f32", waves, out, cyc, rt);
        run<7, 8>("v_pk_fma_f32 sgpr-pair", waves, out, cyc, rt);
        run<18, 8>("pk_fma + fma alternating(x2)", waves, out, cyc, rt);
        run<19, 8>("v_sqrt_f32", waves, out, cyc, rt);
    }
    return 0;
}
