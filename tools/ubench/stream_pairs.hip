// Which pairs of HIP streams of one process run kernels concurrently?  The runtime maps streams onto a small number
// of hardware queues; two streams that share one execute in order.  Overlap mode (sa_set_overlap) needs its internal
// streams on different queues: this probe shows how the mapping falls for streams created one after the other.
// hipcc -O3 --offload-arch=gfx950 stream_pairs.hip -o stream_pairs && ./stream_pairs [nstreams]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void spin(unsigned long long ticks)          // s_memrealtime: 100 MHz, the same on every XCD
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 16;
    std::vector<hipStream_t> s(n);
    for (auto &x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    s.insert(s.begin(), (hipStream_t) nullptr);          // row / column 0: the null stream
    const int created = n;
    (void)created;
    hipEvent_t e0, ea, eb;
    hipEventCreate(&e0); hipEventCreate(&ea); hipEventCreate(&eb);
    const unsigned long long T = 10000;                 // 100 us
    for (auto &x : s) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, x, 100ull);
    hipDeviceSynchronize();
    printf("concurrency of stream i (rows) with stream j (columns): . = concurrent, S = serialised\n    ");
    for (int j = 0; j <= n; ++j) printf("%2d ", j - 1);
    printf("   (-1 = the null stream)\n");
    for (int i = 0; i <= n; ++i) {
        printf("%2d  ", i - 1);
        for (int j = 0; j <= n; ++j) {
            if (i == j) { printf(" - "); continue; }
            hipEventRecord(e0, s[i]);
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[i], T);
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[j], T);
            hipEventRecord(ea, s[i]);
            hipEventRecord(eb, s[j]);
            hipEventSynchronize(ea);
            hipEventSynchronize(eb);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, eb);
            printf(" %c ", ms < 0.15f ? '.' : 'S');
        }
        printf("\n");
    }
    return 0;
}
