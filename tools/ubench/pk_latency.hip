// Micro-benchmark: issue-to-dependent-issue latency of the packed-fp32 ops on the float IIR's critical
// chain (y = x + s1; s1 = na1*y + t), and what interleaving independent chains buys.
// hipcc -O3 --offload-arch=gfx950 pk_latency.hip -o pk_latency && ./pk_latency
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));

// KIND 0: dependent v_pk_fma_f32 chain (VGPR operands)      1: the same with an SGPR multiplier
//      2: dependent v_fma_f32 chain                          3: dependent v_pk_add_f32 chain
//      4: the unit-numerator recursion step, one chain       5: two independent recursion chains interleaved
//      6: recursion + 2 predictor FMAs (the fused loop)      7: two chains + predictors
//      8: four independent recursion chains
template <int KIND>
__global__ void k(float *out, unsigned long long *cyc, int n, float c0, float c1, float c2)
{
    v2f a = {threadIdx.x * 1e-3f, 0.5f}, b = {0.25f, 0.125f}, s1 = a, s2 = b, u1 = b, u2 = a, n1 = {0, 0}, n2 = {0, 0};
    v2f p1 = a, p2 = b, q1 = b, q2 = a;
    const v2f x = {0.01f, 0.02f};
    const float na1 = c0, na2 = c1, b1 = c2;
    const v2f cc = {c0, c1};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %2, %0, %1 op_sel_hi:[0,1,1]" : "+v"(a) : "v"(b), "s"(cc));
            if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a.x) : "v"(b.x));
            if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
            if (KIND == 4 || KIND == 5 || KIND == 6 || KIND == 7 || KIND == 8) {
                v2f y = x + s1;
                s1 = na1 * y + (b1 * x + s2);
                s2 = na2 * y + x;
                if (KIND == 6 || KIND == 7) { n1 += c0 * y; n2 += c1 * y; }
                if (KIND == 5 || KIND == 7 || KIND == 8) {
                    v2f y2 = x + u1;
                    u1 = na1 * y2 + (b1 * x + u2);
                    u2 = na2 * y2 + x;
                    if (KIND == 7) { n1 += c2 * y2; n2 += c1 * y2; }
                }
                if (KIND == 8) {
                    v2f y3 = x + p1;
                    p1 = na1 * y3 + (b1 * x + p2);
                    p2 = na2 * y3 + x;
                    v2f y4 = x + q1;
                    q1 = na1 * y4 + (b1 * x + q2);
                    q2 = na2 * y4 + x;
                }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const v2f r = a + b + s1 + s2 + u1 + u2 + n1 + n2 + p1 + p2 + q1 + q2;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r.x + r.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8 * 4096);
    const int n = 2000;
    const char *names[9] = {"v_pk_fma_f32 chain (VGPR)", "v_pk_fma_f32 chain (SGPR multiplier)", "v_fma_f32 chain", "v_pk_add_f32 chain",
                            "IIR step, 1 chain (4 pk ops)", "IIR step, 2 chains (8 pk ops)", "IIR step + predict, 1 chain (6 pk ops)",
                            "IIR step + predict, 2 chains (12 pk ops)", "IIR step, 4 chains (16 pk ops)"};
    const int ops[9] = {1, 1, 1, 1, 4, 8, 6, 12, 16};
    void (*ks[9])(float *, unsigned long long *, int, float, float, float) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>};
    for (int waves = 1; waves <= 4; waves *= 2)
        for (int kind = 0; kind < 9; ++kind) {
            dim3 g(1), b(64 * 4 * waves);   // 4*waves waves in one workgroup -> `waves` per SIMD of one CU
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(ks[kind], g, b, 0, 0, out, cyc, n, -0.3f, 0.2f, 1.5f);
                hipDeviceSynchronize();
            }
            unsigned long long c;
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%d wave/SIMD  %-44s %6.2f cycles per step, %5.2f per instruction\n", waves, names[kind],
                   (double)c / (n * 16.0), (double)c / (n * 16.0 * ops[kind]));
        }
    return 0;
}
