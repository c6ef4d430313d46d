// fft_regs.hpp -- small power-of-two FFTs held entirely in VGPRs (gfx950).
//
// A thread owns R complex points in registers, each a packed float pair (re, im) so that the
// arithmetic maps onto the packed-fp32 VALU instructions (v_pk_add_f32 / v_pk_fma_f32): gfx950
// issues a wave64 VALU instruction in 4 cycles whether it carries one or two floats per lane.
// The transform is decimation-in-time, radix 2, with the input placed in bit-reversed slots by the
// caller (free: the caller chooses which slot each LDS/global load lands in), so the output comes
// out in natural order.  Every twiddle is a compile-time constant; a butterfly costs 3 packed FMAs
// (a' = a + w*b as two FMAs, b' = 2a - a'), 2 packed adds for w = 1 and w = -i.  All indices are
// compile-time so the arrays never leave registers (cdna_hip_programming.md section 5.4 rule 20).
//
// The functions are __host__ __device__ so tests/cpp/test_fft_regs.cpp can check them on the CPU.
#pragma once
#include <hip/hip_runtime.h>
#include <utility>

namespace safft {

typedef float cf __attribute__((ext_vector_type(2)));   // (re, im)

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

__host__ __device__ __forceinline__ cf swap_ri(cf v) { return __builtin_shufflevector(v, v, 1, 0); }
__host__ __device__ __forceinline__ cf pk_fma(cf a, cf b, cf c) { return __builtin_elementwise_fma(a, b, c); }

// a' = a + W*b, b' = a - W*b with W = exp(-2*pi*i*M32/32)
template <int M32>
__host__ __device__ __forceinline__ void bfly(cf &a, cf &b)
{
    if constexpr (M32 == 0) {
        const cf t = b;
        b = a - t;
        a = a + t;
    } else if constexpr (M32 == 8) {          // W = -i : W*b = (b.y, -b.x)
#if defined(__HIP_DEVICE_COMPILE__)
        // the swap and the negation are operand modifiers; left alone the compiler builds (b.y, -b.x)
        // with a v_xor and a v_mov first
        cf na, nb;
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(na) : "v"(a), "v"(b));
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(nb) : "v"(a), "v"(b));
        a = na;
        b = nb;
#else
        const cf t = {b.y, -b.x};
        b = a - t;
        a = a + t;
#endif
    } else {
        constexpr float wr = cos32(M32), wi = -sin32(M32);
        const cf wrr = {wr, wr};
        const cf wii = {-wi, wi};
        const cf n = pk_fma(wii, swap_ri(b), pk_fma(wrr, b, a));   // a + W*b
        const cf two = {2.0f, 2.0f};
        b = pk_fma(two, a, -n);
        a = n;
    }
}

template <int HALF, int... Ks>
__host__ __device__ __forceinline__ void dit_group(cf *a, std::integer_sequence<int, Ks...>)
{
    (bfly<Ks * (16 / HALF)>(a[Ks], a[Ks + HALF]), ...);
}

template <int R, int HALF>
__host__ __device__ __forceinline__ void dit_stage(cf (&a)[R])
{
#pragma unroll
    for (int g = 0; g < R; g += 2 * HALF) dit_group<HALF>(&a[g], std::make_integer_sequence<int, HALF>{});
}

// in: a[brev(n)] = x[n];  out: a[k] = sum_n x[n] exp(-2*pi*i*n*k/R)
template <int R>
__host__ __device__ __forceinline__ void fft_dit(cf (&a)[R])
{
    static_assert(R == 4 || R == 8 || R == 16 || R == 32, "unsupported size");
    dit_stage<R, 1>(a);
    dit_stage<R, 2>(a);
    if constexpr (R >= 8) dit_stage<R, 4>(a);
    if constexpr (R >= 16) dit_stage<R, 8>(a);
    if constexpr (R >= 32) dit_stage<R, 16>(a);
}

// complex product a * w as two packed ops
__host__ __device__ __forceinline__ cf cmul(cf a, cf w)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // one statement for the pair: between two asm statements the compiler pads a wait state (s_nop) whenever the
    // second reads the first's output, which the hardware does not need for an ordinary VALU dependency
    cf t, r;
    asm("v_pk_mul_f32 %0, %2, %3 op_sel:[0,0] op_sel_hi:[0,1]\n\t"                                        // a.x * (w.x, w.y)
        "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"                      // + a.y * (-w.y, w.x)
        : "=&v"(t), "=v"(r) : "v"(a), "v"(w));
    return r;
#else
    const cf t = {a.x * w.x, a.x * w.y};
    return {__builtin_fmaf(a.y, -w.y, t.x), __builtin_fmaf(a.y, w.x, t.y)};
#endif
}

}  // namespace safft
