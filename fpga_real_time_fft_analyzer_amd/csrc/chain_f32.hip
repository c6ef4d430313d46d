// chain_f32.hip -- fused float signal path for gfx950 (MI355X):
//     Hann window -> 6-section biquad cascade -> 16384-point real FFT -> magnitude
// replacing new/hann8192.vhd -> new/filter_iir12_cust.vhd -> ip/xfft_0 of the reference with one
// pass over HBM (read 64 KiB, write 64 KiB per frame).  One 256-thread workgroup per frame.
//
// Layout of the computation (DESIGN.md sections 3-5):
//   * thread t owns samples [64t, 64t+64) for the IIR (the recursion is serial in time), obtained
//     through a coalesced global read + XOR-swizzled LDS transpose;
//   * the IIR is evaluated per section as  predict (dot product giving each chunk's end state from
//     zero state)  ->  scan over the 256 chunks (2x2 transition powers; wave shuffle + one LDS hop)
//     ->  the exact DF2T recursion started from the scanned state;
//   * the real FFT is an 8192-point complex FFT of z[m] = x[2m] + i x[2m+1] factored 32 x 16 x 16,
//     each factor done in registers (fft_regs.hpp), with padded LDS exchanges between factors,
//     followed by the split step X[k] = Xe[k] + W_N^k Xo[k];
//   * magnitudes of all 16384 bins are written (upper half mirrored), dword-per-lane coalesced.
// The factor 1/2 of the split step is folded into the window table (exact in binary fp).
#include "sa_common.hpp"
#include "fft_regs.hpp"
#include "../../include/specan.h"

using safft::cf;

namespace {

constexpr int kThreads = 256;
constexpr int kLdsComplex = 32 * 272;                 // largest exchange layout (8704 complex)
constexpr int kLdsBytes = kLdsComplex * 8 + 256;      // + scan scratch (6 sections x 4 waves x 2 floats)

__device__ __forceinline__ float fast_sqrt(float v) { return __builtin_amdgcn_sqrtf(v); }

// ---------------------------------------------------------------------------------------------
// Stage-in for the IIR: coalesced 16-byte loads, window multiply, swizzled LDS transpose so that
// thread t ends with its 64 consecutive samples in v[].
// LDS image: row r = chunk (256 B), 16-byte column c stored at column c ^ (r & 15).
__device__ __forceinline__ void stage_in_chunks(const float *__restrict__ xin, const float *__restrict__ win,
                                                float4 *lds4, int t, float (&v)[64])
{
    const float4 *x4 = reinterpret_cast<const float4 *>(xin);
    const float4 *w4 = reinterpret_cast<const float4 *>(win);
    float4 xv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int u = i * 256 + t;
        const float4 a = x4[u];
        const float4 w = w4[u];
        xv[i] = make_float4(a.x * w.x, a.y * w.y, a.z * w.z, a.w * w.w);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = i * 16 + (t >> 4);
        const int pc = (t & 15) ^ (r & 15);
        lds4[r * 16 + pc] = xv[i];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float4 q = lds4[t * 16 + (j ^ (t & 15))];
        v[4 * j + 0] = q.x;
        v[4 * j + 1] = q.y;
        v[4 * j + 2] = q.z;
        v[4 * j + 3] = q.w;
    }
}

// ---------------------------------------------------------------------------------------------
// Cascade over the thread's chunk, all sections, in place.  scr: LDS scratch [6][4] float2.
__device__ __forceinline__ void iir_cascade(float (&v)[64], const SaIirPlan *__restrict__ plan, float2 *scr,
                                            int t)
{
    const int lane = t & 63;
    const int wave = t >> 6;
    const int nsec = plan->nsec;
    for (int s = 0; s < nsec; ++s) {
        const SaIirSecPlan &sp = plan->sec[s];
        // predict: end state of this chunk from zero initial state
        float z1a = 0.f, z1b = 0.f, z2a = 0.f, z2b = 0.f;
#pragma unroll
        for (int j = 0; j < 64; j += 2) {
            z1a = __builtin_fmaf(sp.m[0][j], v[j], z1a);
            z2a = __builtin_fmaf(sp.m[1][j], v[j], z2a);
            z1b = __builtin_fmaf(sp.m[0][j + 1], v[j + 1], z1b);
            z2b = __builtin_fmaf(sp.m[1][j + 1], v[j + 1], z2b);
        }
        float z1 = z1a + z1b, z2 = z2a + z2b;
        // inclusive scan over the 64 chunks of this wave: Z_l = z_l + P Z_{l-1}
#pragma unroll
        for (int lev = 0; lev < 6; ++lev) {
            const int d = 1 << lev;
            const float u1 = __shfl_up(z1, d, 64);
            const float u2 = __shfl_up(z2, d, 64);
            const float p00 = sp.plev[lev][0], p01 = sp.plev[lev][1], p10 = sp.plev[lev][2], p11 = sp.plev[lev][3];
            if (lane >= d) {
                z1 = __builtin_fmaf(p00, u1, __builtin_fmaf(p01, u2, z1));
                z2 = __builtin_fmaf(p10, u1, __builtin_fmaf(p11, u2, z2));
            }
        }
        if (lane == 63) scr[s * 4 + wave] = make_float2(z1, z2);
        __syncthreads();
        // state at the start of this wave's first chunk
        float c1 = 0.f, c2 = 0.f;
        for (int u = 0; u < wave; ++u) {
            const float2 tt = scr[s * 4 + u];
            const float n1 = __builtin_fmaf(sp.p64[0], c1, __builtin_fmaf(sp.p64[1], c2, tt.x));
            const float n2 = __builtin_fmaf(sp.p64[2], c1, __builtin_fmaf(sp.p64[3], c2, tt.y));
            c1 = n1;
            c2 = n2;
        }
        float e1 = __shfl_up(z1, 1, 64), e2 = __shfl_up(z2, 1, 64);
        if (lane == 0) {
            e1 = 0.f;
            e2 = 0.f;
        }
        const float4 pp = *reinterpret_cast<const float4 *>(&sp.ppow[lane][0]);
        float s1 = __builtin_fmaf(pp.x, c1, __builtin_fmaf(pp.y, c2, e1));
        float s2 = __builtin_fmaf(pp.z, c1, __builtin_fmaf(pp.w, c2, e2));
        // exact DF2T recursion from the scanned state
        const float b0 = sp.c[0], b1 = sp.c[1], b2 = sp.c[2], na1 = -sp.c[3], na2 = -sp.c[4];
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const float x = v[j];
            const float y = __builtin_fmaf(b0, x, s1);
            s1 = __builtin_fmaf(na1, y, __builtin_fmaf(b1, x, s2));
            s2 = __builtin_fmaf(na2, y, b2 * x);
            v[j] = y;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Split step of the packed real FFT for the bin pair (k, 8192-k) and the output write.
//   Xe = Z[k] + conj Z[M-k],  Xo = -i (Z[k] - conj Z[M-k]),  X[k] = Xe + W_N^k Xo,  X[M-k] = conj(Xe - W_N^k Xo)
// (the 1/2 of the textbook form is already in the window table).
template <int OUT>
__device__ __forceinline__ void split_pair(const cf *ldc, const float2 *__restrict__ twP, void *__restrict__ out,
                                           int f, int k)
{
    const int km = (SA_MC - k) & (SA_MC - 1);
    const cf zk = ldc[k + (k >> 5)];
    const cf zm = ldc[km + (km >> 5)];
    const float2 w = twP[k];
    const float xer = zk.x + zm.x, xei = zk.y - zm.y;
    const float xor_ = zk.y + zm.y, xoi = zm.x - zk.x;
    const float tr = __builtin_fmaf(w.x, xor_, -w.y * xoi);
    const float ti = __builtin_fmaf(w.x, xoi, w.y * xor_);
    const float pr = xer + tr, pi = xei + ti;               // X[k]
    const float qr = xer - tr, qi = xei - ti;               // conj X[M-k]
    if constexpr (OUT == SA_OUT_MAG_FULL) {
        float *o = reinterpret_cast<float *>(out) + (size_t)f * SA_NPTS;
        const float mp = fast_sqrt(__builtin_fmaf(pr, pr, pi * pi));
        const float mq = fast_sqrt(__builtin_fmaf(qr, qr, qi * qi));
        o[k] = mp;
        o[SA_MC - k] = mq;
        o[SA_MC + k] = mq;
        if (k != 0) o[SA_NPTS - k] = mp;
    } else if constexpr (OUT == SA_OUT_MAG_HALF) {
        float *o = reinterpret_cast<float *>(out) + (size_t)f * (SA_MC + 1);
        o[k] = fast_sqrt(__builtin_fmaf(pr, pr, pi * pi));
        o[SA_MC - k] = fast_sqrt(__builtin_fmaf(qr, qr, qi * qi));
    } else {
        float2 *o = reinterpret_cast<float2 *>(out) + (size_t)f * (SA_MC + 1);
        o[k] = make_float2(pr, pi);
        o[SA_MC - k] = make_float2(qr, -qi);
    }
}

// ---------------------------------------------------------------------------------------------
template <bool IIR, int OUT>
__global__ __launch_bounds__(kThreads, 2) void chain_f32_kernel(const float *__restrict__ in,
                                                                 void *__restrict__ out, int batch,
                                                                 const float *__restrict__ win,
                                                                 const float2 *__restrict__ twA,
                                                                 const float2 *__restrict__ twB,
                                                                 const float2 *__restrict__ twP,
                                                                 const SaIirPlan *__restrict__ plan)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf *ldc = reinterpret_cast<cf *>(smem);
    float4 *lds4 = reinterpret_cast<float4 *>(smem);
    float2 *scr = reinterpret_cast<float2 *>(smem + kLdsComplex * 8);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;

    {
        // one frame per workgroup: no loop => no loop-invariant address hoisting (which spilled)
        const int f = blockIdx.x;
        if (f >= batch) return;
        const float *xin = in + (size_t)f * SA_NPTS;
        cf a[32];

        if constexpr (IIR) {
            float v[64];
            stage_in_chunks(xin, win, lds4, t, v);
            iir_cascade(v, plan, scr, t);
            // exchange: chunk layout -> pass-A layout (index + index/32 padding)
#pragma unroll
            for (int j = 0; j < 32; ++j) ldc[33 * t + j] = {v[2 * j], v[2 * j + 1]};
            __syncthreads();
#pragma unroll
            for (int m1 = 0; m1 < 32; ++m1) a[safft::brev(m1, 5)] = ldc[264 * m1 + t + (t >> 5)];
        } else {
            const float2 *x2 = reinterpret_cast<const float2 *>(xin);
            const float2 *w2 = reinterpret_cast<const float2 *>(win);
#pragma unroll
            for (int m1 = 0; m1 < 32; ++m1) {
                const float2 xv = x2[256 * m1 + t];
                const float2 wv = w2[256 * m1 + t];
                a[safft::brev(m1, 5)] = {xv.x * wv.x, xv.y * wv.y};
            }
        }

        // ---- pass A: 32-point FFT over m1 (stride 256), then twiddle W_8192^(k1*m2), m2 = t
        safft::fft_dit<32>(a);
#pragma unroll
        for (int k1 = 1; k1 < 32; ++k1) {
            const float2 w = twA[k1 * 256 + t];
            a[k1] = safft::cmul(a[k1], {w.x, w.y});
        }
        __syncthreads();   // every read of the previous LDS image is done
#pragma unroll
        for (int k1 = 0; k1 < 32; ++k1) ldc[k1 * 272 + t] = a[k1];
        __syncthreads();

        // ---- pass B: for (k1, b): 16-point FFT over a (m2 = 16a + b), twiddle W_256^(b*c)
        const int lo = lane & 15;          // b in pass B, c in pass C
        const int kq = lane >> 4;
        cf p[2][16];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int k1 = 8 * wave + 4 * q + kq;
#pragma unroll
            for (int aa = 0; aa < 16; ++aa) p[q][safft::brev(aa, 4)] = ldc[k1 * 272 + 16 * aa + lo];
        }
        safft::fft_dit<16>(p[0]);
        safft::fft_dit<16>(p[1]);
#pragma unroll
        for (int c = 1; c < 16; ++c) {
            const float2 w = twB[c * 16 + lo];
            p[0][c] = safft::cmul(p[0][c], {w.x, w.y});
            p[1][c] = safft::cmul(p[1][c], {w.x, w.y});
        }
        __syncthreads();   // all pass-B reads done before the image is overwritten
        // exchange inside each 16-lane group: [k1][c][b] with row pitch 17
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int base = (8 * wave + 4 * q + kq) * 272;
#pragma unroll
            for (int c = 0; c < 16; ++c) ldc[base + c * 17 + lo] = p[q][c];
        }
        __syncthreads();
        // ---- pass C: for (k1, c): 16-point FFT over b -> d;  Z[k1 + 32c + 512d]
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int base = (8 * wave + 4 * q + kq) * 272;
#pragma unroll
            for (int b = 0; b < 16; ++b) p[q][safft::brev(b, 4)] = ldc[base + lo * 17 + b];
        }
        safft::fft_dit<16>(p[0]);
        safft::fft_dit<16>(p[1]);
        __syncthreads();
        // natural-order image of Z with index + index/32 padding
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int k1 = 8 * wave + 4 * q + kq;
#pragma unroll
            for (int d = 0; d < 16; ++d) ldc[k1 + 33 * lo + 528 * d] = p[q][d];
        }
        __syncthreads();

        // ---- split step + output.  Pair (k, 8192-k), k = t + 256j; thread 0 also takes k = 4096.
#pragma unroll 4
        for (int j = 0; j < 16; ++j) split_pair<OUT>(ldc, twP, out, f, t + 256 * j);
        if (t == 0) split_pair<OUT>(ldc, twP, out, f, 4096);
    }
}

// Window (+ IIR) only: the FFT input time series (debug / parity output, not a hot path).
template <bool IIR>
__global__ __launch_bounds__(kThreads, 2) void time_f32_kernel(const float *__restrict__ in,
                                                                float *__restrict__ out, int batch,
                                                                const float *__restrict__ win,
                                                                const SaIirPlan *__restrict__ plan)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *lds4 = reinterpret_cast<float4 *>(smem);
    float2 *scr = reinterpret_cast<float2 *>(smem + kLdsComplex * 8);
    const int t = threadIdx.x;
    {
        const int f = blockIdx.x;
        if (f >= batch) return;
        float v[64];
        stage_in_chunks(in + (size_t)f * SA_NPTS, win, lds4, t, v);
        if constexpr (IIR) iir_cascade(v, plan, scr, t);
        float4 *o4 = reinterpret_cast<float4 *>(out + (size_t)f * SA_NPTS + 64 * t);
#pragma unroll
        for (int j = 0; j < 16; ++j)   // undo the folded 1/2 (exact)
            o4[j] = make_float4(2.f * v[4 * j], 2.f * v[4 * j + 1], 2.f * v[4 * j + 2], 2.f * v[4 * j + 3]);
    }
}

template <typename K>
hipError_t set_lds(K kernel)
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               kLdsBytes);
}

}  // namespace

hipError_t sa_launch_chain_f32(const float *in, void *out, int batch, int out_kind, bool iir,
                               const SaF32Tables &tb, hipStream_t stream)
{
    if (batch <= 0) return hipSuccess;
    const dim3 grid(batch), block(kThreads);
    hipError_t e = hipSuccess;
#define SA_LAUNCH(IIRF, OUTK)                                                                          \
    do {                                                                                               \
        auto kern = chain_f32_kernel<IIRF, OUTK>;                                                      \
        e = set_lds(kern);                                                                             \
        if (e != hipSuccess) return e;                                                                 \
        hipLaunchKernelGGL(kern, grid, block, kLdsBytes, stream, in, out, batch, tb.win_half, tb.twA, \
                           tb.twB, tb.twP, tb.plan);                                                   \
    } while (0)
    if (out_kind == SA_OUT_TIME) {
        if (iir) {
            auto kern = time_f32_kernel<true>;
            e = set_lds(kern);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, grid, block, kLdsBytes, stream, in, reinterpret_cast<float *>(out), batch,
                               tb.win_half, tb.plan);
        } else {
            auto kern = time_f32_kernel<false>;
            e = set_lds(kern);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, grid, block, kLdsBytes, stream, in, reinterpret_cast<float *>(out), batch,
                               tb.win_half, tb.plan);
        }
    } else if (iir) {
        switch (out_kind) {
            case SA_OUT_MAG_FULL: SA_LAUNCH(true, SA_OUT_MAG_FULL); break;
            case SA_OUT_MAG_HALF: SA_LAUNCH(true, SA_OUT_MAG_HALF); break;
            case SA_OUT_SPEC_HALF: SA_LAUNCH(true, SA_OUT_SPEC_HALF); break;
            default: return hipErrorInvalidValue;
        }
    } else {
        switch (out_kind) {
            case SA_OUT_MAG_FULL: SA_LAUNCH(false, SA_OUT_MAG_FULL); break;
            case SA_OUT_MAG_HALF: SA_LAUNCH(false, SA_OUT_MAG_HALF); break;
            case SA_OUT_SPEC_HALF: SA_LAUNCH(false, SA_OUT_SPEC_HALF); break;
            default: return hipErrorInvalidValue;
        }
    }
#undef SA_LAUNCH
    return hipGetLastError();
}
