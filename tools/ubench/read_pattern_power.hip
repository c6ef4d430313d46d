// Does the ORDER in which a frame is fetched change what the memory system draws?  One 64 KiB frame in, one out per
// 256-thread workgroup (LDS-DMA in, 16-byte nontemporal stores out, no arithmetic), the frame fetched in two rounds:
//   pattern 0: round h = the h-th contiguous 32 KiB of the frame                     (the bypass kernel's stage-in)
//   pattern 1: round h = the h-th 128-byte half of every 256-byte row of the frame   (the IIR kernels' stage-in:
//              a thread owns 64 consecutive samples and takes them as two chunks of 32)
// Runs the chosen pattern back to back for `seconds`; the caller samples rocm-smi meanwhile (tools/read_pattern_power.sh).
// hipcc -O3 --offload-arch=gfx950 read_pattern_power.hip -o read_pattern_power && ./read_pattern_power PATTERN SECONDS
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

constexpr int N = 16384;
typedef float f4v __attribute__((ext_vector_type(4)));

template <int PATTERN>
__global__ __launch_bounds__(256, 4) void k(const float *__restrict__ in, float *__restrict__ out, int batch)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, f = blockIdx.x;
    if (f >= batch) return;
    const float *xin = in + (size_t)f * N;
    float *o = out + (size_t)f * N;
    const f4v *lds4 = reinterpret_cast<const f4v *>(smem);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h == 1) __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = wave * 8 + i;                                   // 1 KiB slab of the LDS image
            const float *src;
            if (PATTERN == 0) src = xin + h * 8192 + n * 256 + lane * 4;
            else src = xin + (8 * n + (lane >> 3)) * 64 + h * 32 + (lane & 7) * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(smem + n * 1024), 16, 0, 0);
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f4v v = lds4[g * 256 + t];
            __builtin_nontemporal_store(v, reinterpret_cast<f4v *>(o + h * 8192) + g * 256 + t);
        }
    }
}

int main(int argc, char **argv)
{
    const int pattern = argc > 1 ? atoi(argv[1]) : 0;
    const double seconds = argc > 2 ? atof(argv[2]) : 4.0;
    const int batch = 4096, ROT = 4;
    float *in[ROT], *out[ROT];
    for (int i = 0; i < ROT; ++i) {
        hipMalloc(&in[i], (size_t)batch * N * 4);
        hipMalloc(&out[i], (size_t)batch * N * 4);
        hipMemset(in[i], 0, (size_t)batch * N * 4);
    }
    auto kern = pattern == 0 ? k<0> : k<1>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 35616);
    const auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    double dt = 0;
    do {
        for (int i = 0; i < 100; ++i, ++n) hipLaunchKernelGGL(kern, dim3(batch), dim3(256), 35616, 0, in[n % ROT], out[n % ROT], batch);
        hipDeviceSynchronize();
        dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (dt < seconds);
    printf("pattern %d: %ld launches in %.2f s = %.1f us per launch = %.2f TB/s\n", pattern, n, dt, dt / n * 1e6,
           2.0 * batch * N * 4 / (dt / n) / 1e12);
    return 0;
}
