// chain_f32_small.hip -- the bypassed float chain (Hann window -> 16384-point real FFT -> magnitude, mode 0xB1: the board's
// power-on mode, new/command_control.vhd:31) for SMALL batches: one 512-thread workgroup per frame, i.e. two waves per
// SIMD on ONE frame, where chain_f32.hip runs one.  BASELINE config 2 is B = 256 = one frame per CU: with a single
// four-wave workgroup on the CU every phase of the frame runs on lone waves (one vector instruction per ~6 clocks, LDS
// round trips with nothing to cover them; profiles/r4_phase_stamps_noiir_lone.txt), and a frame's phases cannot overlap
// each other.  Half the arithmetic per wave and a partner wave on every SIMD shorten the middle of the frame.
//
// How eight waves share one transform without meeting at every barrier of a common exchange (round 3's 512-thread form
// did, and lost): the 8192-point complex FFT of z[m] = x[2m] + i x[2m+1] starts with ONE radix-2 decimation-in-frequency
// step,
//     e[m] = z[m] + z[m + 4096],      o[m] = (z[m] - z[m + 4096]) W_8192^m,      m < 4096,
// after which Z[2k'] = FFT_4096(e)[k'] and Z[2k'+1] = FFT_4096(o)[k'] are two INDEPENDENT 4096-point transforms.  Waves
// 0..3 take e, waves 4..7 take o; each half is chain_f32.hip's transform with a 16-point first pass instead of a
// 32-point one (4096 = 16 x 16 x 16), in its own LDS region, on its own data.  Both halves read the same 32 inputs per
// thread out of the frame's LDS image (the whole frame, 64 KiB: the CU is otherwise empty).  The halves meet again only in
// the natural-order image of the split step, which chain_f32.hip has anyway: half g writes the bins of parity g,
//     Z[2 (k1 + 16 c + 256 d) + g] = Z[(2 k1 + g) + 32 c + 512 d],
// exactly the image layout of the 256-thread kernel with its k1 replaced by 2 k1 + g -- so the split step, its twiddles
// and its stores are that kernel's, divided over twice the threads.  No new table: the twiddles W_4096^(k1 u) =
// W_8192^(2 k1 u) and W_8192^u come from the anchors the 256-thread kernel uses (SaF32Tables::twT).
// Full-spectrum magnitudes of float32 frames only; every other case takes chain_f32.hip (sa_launch_chain_f32).
#include "chain_f32_dev.hpp"

namespace {

constexpr int kThreadsS = 512;
constexpr int kHalfRegion = 16 * 272;                 // complex slots of one half's exchange region (34 KiB)
constexpr int kSideOffS = 2 * kHalfRegion * 8;        // two complex side slots behind the two regions
constexpr int kLdsSmall = kSideOffS + 16;             // 69 648 B >= the 64 KiB input image that lives there first

// W_32^m = exp(-2 pi i m / 32), m = 0..15 (the constant factor of the decimation-in-frequency twiddles)
__device__ constexpr float kW32[16][2] = {{1.000000000f, -0.000000000f}, {0.980785280f, -0.195090322f}, {0.923879533f, -0.382683432f}, {0.831469612f, -0.555570233f}, {0.707106781f, -0.707106781f}, {0.555570233f, -0.831469612f}, {0.382683432f, -0.923879533f}, {0.195090322f, -0.980785280f}, {0.000000000f, -1.000000000f}, {-0.195090322f, -0.980785280f}, {-0.382683432f, -0.923879533f}, {-0.555570233f, -0.831469612f}, {-0.707106781f, -0.707106781f}, {-0.831469612f, -0.555570233f}, {-0.923879533f, -0.382683432f}, {-0.980785280f, -0.195090322f}};

// Position of Z[k] inside the half image of the natural-order exchange (chain_f32.hip, zrow_pos)
__device__ __forceinline__ int zrow_pos_s(int within, int row)
{
    const int rest = within & 511;
    return rest + (rest >> 5) + 528 * row;
}
__device__ __forceinline__ int zpos_low_s(int q) { return zrow_pos_s(q, q >> 9); }
__device__ __forceinline__ int zpos_partner_s(int w) { return zrow_pos_s(w, (4 + (w >> 9)) & 7); }

__global__ __launch_bounds__(kThreadsS) void chain_f32_small_kernel(const float *__restrict__ in, float *__restrict__ out, int batch,
                                                                    const float4 *__restrict__ winb,
                                                                    const float4 *__restrict__ twT,
                                                                    const float4 *__restrict__ twB,
                                                                    const float2 *__restrict__ twC)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int f = blockIdx.x;
    if (f >= batch) return;
    cf *ldc = reinterpret_cast<cf *>(smem);
    cf *side = reinterpret_cast<cf *>(smem + kSideOffS);
    const int tt = threadIdx.x;                       // 0..511
    const int g = __builtin_amdgcn_readfirstlane(tt >> 8);     // half: 0 = even bins (e), 1 = odd bins (o); wave-uniform
    const int t = tt & 255;                           // column u of the half = thread index of the 256-thread kernel
    const int lane = tt & 63;
    const int wave = tt >> 6;                         // 0..7
    const int wh = wave & 3;                          // wave inside the half
    const int lo = lane & 15;
    const int kq = lane >> 4;
    const float *xin = in + (size_t)f * SA_NPTS;
    cf *reg = ldc + g * kHalfRegion;                  // this half's exchange region

    // ---- stage-in: the whole frame HBM -> LDS in natural order (64 requests of 1 KiB, eight per wave)
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n = wave * 8 + i;
        const float *src = xin + n * 256 + lane * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(smem + n * 1024), 16, 0, SA_DMA_AUX);
    }
    __builtin_amdgcn_s_setprio(0);
    // anchors (requested behind the DMA, consumed after the first butterflies): rows 0..4 of twT hold W^(1..8 t), W^(16 t),
    // W^(24 t) with W = W_8192; row 5 the split-step anchors
    float4 an[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) an[i] = twT[i * 256 + t];
    const float4 an5 = twT[5 * 256 + t];
    __syncthreads();

    // ---- the radix-2 DIF step on the windowed samples: a[m1] = z[256 m1 + t] (+/-) z[256 (m1 + 16) + t], m1 = 0..15
    cf a[16];
    {
        const cf wu = {an[0].x, an[0].y};                                        // W_8192^t
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            // window of the pass-A layout: winb[p][t] = window at samples 512 (2p) + 2t, +1, 512 (2p + 1) + 2t, +1
            const float4 wl = winb[pp * 256 + t], wh4 = winb[(pp + 8) * 256 + t];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int m1 = 2 * pp + s;
                const cf zl = ldc[256 * m1 + t], zh = ldc[256 * (m1 + 16) + t];
                const cf xl = s == 0 ? cf{zl.x * wl.x, zl.y * wl.y} : cf{zl.x * wl.z, zl.y * wl.w};
                const cf xh = s == 0 ? cf{zh.x * wh4.x, zh.y * wh4.y} : cf{zh.x * wh4.z, zh.y * wh4.w};
                cf v;
                if (g == 0) {
                    v = xl + xh;
                } else {
                    v = xl - xh;
                    // W_8192^(256 m1 + t) = W_32^m1 * W_8192^t
                    if (m1 != 0) v = safft::cmul(v, cf{kW32[m1][0], kW32[m1][1]});
                    v = safft::cmul(v, wu);
                }
                a[safft::brev(m1, 4)] = v;
            }
        }
    }
    // ---- pass A: 16-point FFT over m1, then twiddle W_4096^(k1 t) = W_8192^(2 k1 t): 2 k1 = 8 a + b, b in {0, 2, 4, 6}
    safft::fft_dit<16>(a);
    {
        const cf wb[4] = {{1.f, 0.f}, {an[0].z, an[0].w}, {an[1].z, an[1].w}, {an[2].z, an[2].w}};      // W^(2t), W^(4t), W^(6t)
        const cf wa[4] = {{1.f, 0.f}, {an[3].z, an[3].w}, {an[4].x, an[4].y}, {an[4].z, an[4].w}};      // W^(8t), W^(16t), W^(24t)
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) {
            if ((k1 & 3) != 0) a[k1] = safft::cmul(a[k1], wb[k1 & 3]);
            if ((k1 >> 2) != 0) a[k1] = safft::cmul(a[k1], wa[k1 >> 2]);
        }
    }
    // ---- exchange A -> B inside the half: row k1 (pitch 272), column t; thread (wh, kq, lo) reads row 4 wh + kq,
    //      columns 16 aa + lo.  The input image is dead once every thread holds its samples.
    cf p[16];
    lds_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) reg[r * 272 + t] = a[r];
    lds_barrier();
    const int row = 4 * wh + kq;                      // k1 of this thread from here on
#pragma unroll
    for (int aa = 0; aa < 16; ++aa) p[safft::brev(aa, 4)] = reg[row * 272 + 16 * aa + lo];
    // ---- pass B: 16-point FFT over a, twiddle W_256^(b c), b = lo
    safft::fft_dit<16>(p);
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) {                       // twB[pp][b] = (W_256^(2pp * b), W_256^((2pp+1) * b))
        const float4 w = twB[pp * 16 + lo];
        if (pp > 0) p[2 * pp] = safft::cmul(p[2 * pp], {w.x, w.y});
        p[2 * pp + 1] = safft::cmul(p[2 * pp + 1], {w.z, w.w});
    }
    // ---- exchange B -> C: a 16x16 transpose inside each 16-lane group, through the row this group just read (pitch 17);
    //      only these 16 lanes touch the row: no workgroup barrier, the LDS executes a wave's accesses in order
    {
        const int base = row * 272;
#pragma unroll
        for (int c = 0; c < 16; ++c) reg[base + c * 17 + lo] = p[c];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int b = 0; b < 16; ++b) p[safft::brev(b, 4)] = reg[base + lo * 17 + b];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // ---- pass C: 16-point FFT over b -> d;  this half's bin k' = row + 16 c + 256 d, c = lo, i.e. Z[(2 row + g) + 32 c + 512 d]
    safft::fft_dit<16>(p);
    const int k1z = 2 * row + g;                      // the k1 of the 256-thread kernel's image layout
    const cf wP = {an5.x, an5.y}, wPn = {an5.z, an5.w};
    // ---- natural-order image + split step, two rounds (chain_f32.hip): round 0 = d in {0..3, 12..15}, round 1 = d in {4..11};
    //      thread tt evaluates the bin group jj = g of the 256-thread kernel's thread t
    float *orow = out;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        lds_barrier();
#pragma unroll
        for (int dd = 0; dd < 8; ++dd) {
            const int dsel = (r == 0) ? (dd < 4 ? dd : dd + 8) : dd + 4;
            ldc[k1z + 33 * lo + 528 * dd] = p[dsel];
        }
        if (r == 0 && tt == 0) {                              // k1 = 0, c = 0 of the even half: d = 12 and d = 4
            side[0] = p[12];
            side[1] = p[4];
        }
        lds_barrier();
        const int jj = g;
        const int q0 = 4 * (t + 256 * jj);                 // k0 - 2048 r: bins q0 .. q0+4 of this round
        const int k0 = q0 + 2048 * r;
        cf w[5];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = cmul_s(wP, twC[(2 * r + jj) * 5 + e]);
        {
            const float2 c0 = twC[(2 * r + jj) * 5], c1 = twC[(2 * r + jj + 1) * 5];
            const cf csel = (t == 255) ? cf{c1.x, c1.y} : cf{c0.x, c0.y};
            w[4] = safft::cmul(wPn, csel);
        }
        const int pa = zpos_low_s(q0), pb = zpos_low_s(q0 + 4);
        const int pm0 = zpos_partner_s(2048 - q0), pm4 = zpos_partner_s(2044 - q0);
        cf zk[5] = {ldc[pa], ldc[pa + 1], ldc[pa + 2], ldc[pa + 3], ldc[pb]};
        cf zm[5] = {ldc[pm0], ldc[pm4 + 3], ldc[pm4 + 2], ldc[pm4 + 1], ldc[pm4]};
        if (r == 0 && q0 == 2044) {                           // the seam pair (2048, 6144)
            zk[4] = side[1];
            zm[4] = side[0];
        }
        if (r == 1 && q0 == 0) zm[0] = side[0];
        cf R[5], I[5];
#pragma unroll
        for (int e = 0; e < 5; ++e) split_eval(zk[e], zm[e], w[e], R[e], I[e]);
        split_store<SA_OUT_MAG_FULL>(R, I, orow, f, k0);
    }
}

}  // namespace

hipError_t sa_launch_chain_f32_small(const float *in, float *out, int batch, const SaF32Tables &tb, hipStream_t stream, SaLaunchEv ev)
{
    if (batch <= 0) return hipSuccess;
    const hipError_t e = sa_set_dyn_lds_once(reinterpret_cast<const void *>(chain_f32_small_kernel), kLdsSmall);
    if (e != hipSuccess) return e;
    hipExtLaunchKernelGGL(chain_f32_small_kernel, dim3(batch), dim3(kThreadsS), kLdsSmall, stream, ev.start, ev.stop, 0, in, out, batch,
                          tb.win_b, tb.twT, tb.twB, tb.twC);
    return hipGetLastError();
}
