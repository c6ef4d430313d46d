// fft_regs.hpp -- small power-of-two FFTs held entirely in VGPRs (gfx950).
//
// A thread owns R complex points in registers.  The transform is decimation-in-time, radix 2,
// with the input placed in bit-reversed slots by the caller (free: the caller chooses which
// slot each LDS/global load lands in), so the output comes out in natural order.  Every
// twiddle is a compile-time constant; a butterfly costs 6 FMAs (a' = a + w*b as two FMA
// chains, b' = 2a - a'), 4 adds for w = 1 and w = -i.  All indices are compile-time so the
// arrays never leave registers (cdna_hip_programming.md section 5.4 rule 20).
#pragma once
#include <hip/hip_runtime.h>
#include <utility>

namespace safft {

struct cf {
    float x, y;
};

__host__ __device__ constexpr int brev(int v, int bits)
{
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}

// cos(2*pi*m/32), m = 0..8
__host__ __device__ constexpr float cos32_q(int m)
{
    constexpr float t[9] = {1.0f,
                            0.98078528040323043f,
                            0.92387953251128674f,
                            0.83146961230254524f,
                            0.70710678118654757f,
                            0.55557023301960218f,
                            0.38268343236508978f,
                            0.19509032201612825f,
                            0.0f};
    return t[m];
}

__host__ __device__ constexpr float cos32(int m)
{
    m = ((m % 32) + 32) % 32;
    if (m <= 8) return cos32_q(m);
    if (m <= 16) return -cos32_q(16 - m);
    if (m <= 24) return -cos32_q(m - 16);
    return cos32_q(32 - m);
}

__host__ __device__ constexpr float sin32(int m) { return cos32(m - 8); }

// a' = a + W*b, b' = a - W*b with W = exp(-2*pi*i*M32/32)
template <int M32>
__device__ __forceinline__ void bfly(cf &a, cf &b)
{
    if constexpr (M32 == 0) {
        const cf t = b;
        b = {a.x - t.x, a.y - t.y};
        a = {a.x + t.x, a.y + t.y};
    } else if constexpr (M32 == 8) {          // W = -i : W*b = (b.y, -b.x)
        const cf t = {b.y, -b.x};
        b = {a.x - t.x, a.y - t.y};
        a = {a.x + t.x, a.y + t.y};
    } else {
        constexpr float wr = cos32(M32), wi = -sin32(M32);
        const float nr = __builtin_fmaf(wr, b.x, __builtin_fmaf(-wi, b.y, a.x));
        const float ni = __builtin_fmaf(wr, b.y, __builtin_fmaf(wi, b.x, a.y));
        b = {__builtin_fmaf(2.0f, a.x, -nr), __builtin_fmaf(2.0f, a.y, -ni)};
        a = {nr, ni};
    }
}

template <int HALF, int... Ks>
__device__ __forceinline__ void dit_group(cf *a, std::integer_sequence<int, Ks...>)
{
    (bfly<Ks * (16 / HALF)>(a[Ks], a[Ks + HALF]), ...);
}

template <int R, int HALF>
__device__ __forceinline__ void dit_stage(cf (&a)[R])
{
#pragma unroll
    for (int g = 0; g < R; g += 2 * HALF) dit_group<HALF>(&a[g], std::make_integer_sequence<int, HALF>{});
}

// in: a[brev(n)] = x[n];  out: a[k] = sum_n x[n] exp(-2*pi*i*n*k/R)
template <int R>
__device__ __forceinline__ void fft_dit(cf (&a)[R])
{
    static_assert(R == 4 || R == 8 || R == 16 || R == 32, "unsupported size");
    dit_stage<R, 1>(a);
    dit_stage<R, 2>(a);
    if constexpr (R >= 8) dit_stage<R, 4>(a);
    if constexpr (R >= 16) dit_stage<R, 8>(a);
    if constexpr (R >= 32) dit_stage<R, 16>(a);
}

__device__ __forceinline__ cf cmul(cf a, cf w)
{
    return {__builtin_fmaf(a.x, w.x, -a.y * w.y), __builtin_fmaf(a.x, w.y, a.y * w.x)};
}

}  // namespace safft
