// Semantics check on the device: v_dot2_i32_i16 and the SDWA shift-and-insert pair used by the Q15 FFT butterfly,
// against plain C on random operands.  hipcc -O2 --offload-arch=gfx950 sdwa_dot2_check.hip -o sdwa_dot2_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k(const int *a, const int *b, int *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = a[i], y = b[i];
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(x), "v"(y));
    unsigned p, q;
    const int xs = x >> 13, ys = y >> 13;       // 19-bit values: (v >> 2) fits 17 bits, low word taken
    asm("v_ashrrev_i32_sdwa %0, %4, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_ashrrev_i32_sdwa %1, %4, %3 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_ashrrev_i32_sdwa %0, %4, %3 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_ashrrev_i32_sdwa %1, %4, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "s_nop 0"
        : "=&v"(p), "=&v"(q) : "v"(xs), "v"(ys), "v"(2));
    out[4 * i + 0] = d;
    out[4 * i + 1] = (int)p;
    out[4 * i + 2] = (int)q;
    out[4 * i + 3] = 0;
}

int main()
{
    const int n = 1 << 16;
    std::vector<int> a(n), b(n), o(4 * n);
    srand(3);
    for (int i = 0; i < n; ++i) { a[i] = (rand() << 16) ^ rand(); b[i] = (rand() << 16) ^ rand() ^ (rand() << 1); }
    int *da, *db, *dout;
    (void)hipMalloc(&da, 4 * n); (void)hipMalloc(&db, 4 * n); (void)hipMalloc(&dout, 16 * n);
    (void)hipMemcpy(da, a.data(), 4 * n, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, b.data(), 4 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
    (void)hipMemcpy(o.data(), dout, 16 * n, hipMemcpyDeviceToHost);
    int bad_dot = 0, bad_p = 0, bad_q = 0;
    for (int i = 0; i < n; ++i) {
        const int x = a[i], y = b[i];
        const int d = (int)((unsigned)((short)x * (short)y) + (unsigned)((x >> 16) * (y >> 16)));
        const int xs = x >> 13, ys = y >> 13;
        const unsigned p = ((unsigned)(xs >> 2) & 0xFFFFu) | ((unsigned)(ys >> 2) << 16);
        const unsigned q = ((unsigned)(ys >> 2) & 0xFFFFu) | ((unsigned)(xs >> 2) << 16);
        if (o[4 * i] != d) { if (bad_dot < 3) printf("dot2 %08x . %08x = %d, expected %d\n", x, y, o[4 * i], d); ++bad_dot; }
        if ((unsigned)o[4 * i + 1] != p) { if (bad_p < 3) printf("pack p %08x expected %08x (xs %08x ys %08x)\n", o[4 * i + 1], p, xs, ys); ++bad_p; }
        if ((unsigned)o[4 * i + 2] != q) { if (bad_q < 3) printf("pack q %08x expected %08x\n", o[4 * i + 2], q); ++bad_q; }
    }
    printf("v_dot2_i32_i16: %d of %d differ; shift-insert pair: %d / %d of %d differ\n", bad_dot, n, bad_p, bad_q, n);
    return 0;
}
