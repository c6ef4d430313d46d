// chain_f32_dev.hpp -- device helpers of the float-path kernels (chain_f32.hip): LDS-only barrier, DPP row shifts,
// nontemporal stores, the split step of the packed real FFT and its output layouts.
#pragma once
#include "sa_common.hpp"
#include "fft_regs.hpp"
#include "../../include/specan.h"

using safft::cf;
typedef float v2f __attribute__((ext_vector_type(2)));

namespace {

#ifndef SA_DMA_AUX
#define SA_DMA_AUX 2          // cache policy bits of the input LDS-DMA: 2 = nontemporal (a frame is read once); 0 in A/B
                              // builds: 132.2 -> 130.3 us stream-ordered, no difference with two launches in flight
#endif

__device__ __forceinline__ float fast_sqrt(float v) { return __builtin_amdgcn_sqrtf(v); }

// Phase stamps: diagnostic build only (make stamps -> libspecan_hip_stamps.so, tools/phase_stamps.py).
// In the product build SA_STAMP expands to nothing.
#ifdef SA_STAMPS
__device__ unsigned long long *g_sa_stamps = nullptr;
#define SA_STAMP(i)                                                                        \
    do {                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        if (threadIdx.x == 0 && g_sa_stamps) {   /* never a store through a null table */  \
            unsigned long long c_;                                                         \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c_)::"memory");     \
            g_sa_stamps[(size_t)blockIdx.x * 16 + (i)] = c_;                               \
        }                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    } while (0)
#else
#define SA_STAMP(i) do {} while (0)
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
// counter (s_waitcnt vmcnt(0)), i.e. it waits for every outstanding global load AND store of the wave;
// the exchanges below hand data over through LDS alone, so in-flight twiddle loads and output stores
// may stay in flight across them.  (The stage-in barrier keeps __syncthreads(): the LDS-DMA completes
// on vmcnt.)
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---------------------------------------------------------------------------------------------
// DPP helpers: value of the lane `n` to the left inside the 16-lane row, 0 when there is none.
template <int N>
__device__ __forceinline__ float row_shr(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + N, 0xF, 0xF, true));
}

__device__ __forceinline__ float lane_get(float v, int src_lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v)));
}

// Output stores are streaming: every byte is written once and not read by this launch.  Marked
// nontemporal they do not displace the window / twiddle tables (and the other workgroups' input lines)
// from L2: measured -9 % on the fused kernel and -19 % on the no-IIR kernel at B = 4096.
typedef float f4nt __attribute__((ext_vector_type(4)));
typedef float f2nt __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_nt(float *p, float a, float b, float c, float d)
{
    __builtin_nontemporal_store(f4nt{a, b, c, d}, reinterpret_cast<f4nt *>(p));
}
__device__ __forceinline__ void store_nt(float2 *p, float a, float b)
{
    __builtin_nontemporal_store(f2nt{a, b}, reinterpret_cast<f2nt *>(p));
}
__device__ __forceinline__ void store_nt(float *p, float a) { __builtin_nontemporal_store(a, p); }

// ---------------------------------------------------------------------------------------------
// a * w with w wave-uniform (an SGPR pair): two packed ops, no copy of w into VGPRs
__device__ __forceinline__ cf cmul_s(const cf a, const float2 wu)
{
    const cf w = {wu.x, wu.y};
    cf t, r;
    asm("v_pk_mul_f32 %0, %2, %3 op_sel:[0,0] op_sel_hi:[0,1]\n\t"                                        // a.x * (w.x, w.y)
        "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"                      // + a.y * (-w.y, w.x)
        : "=&v"(t), "=v"(r) : "v"(a), "s"(w));
    return r;
}

// Split step of the packed real FFT for the bin pair (k, 8192-k):
//   Xe = Z[k] + conj Z[M-k],  Xo = -i (Z[k] - conj Z[M-k]),  X[k] = Xe + W_N^k Xo,  X[M-k] = conj(Xe - W_N^k Xo)
// (the 1/2 of the textbook form is already in the window table).  With s = Z[k] + Z[M-k], d = Z[k] - Z[M-k]:
//   Xe = (s.x, d.y), Xo = (s.y, -d.x), T = W Xo = s.y (w.x, w.y) + d.x (w.y, -w.x)
// and the results come out transposed, R = (Re P, Re Q) = s.x + (T.x, -T.x), I = (Im P, Im Q) = d.y + (T.y, -T.y)
// with P = X[k], Q = conj X[M-k]: every operand is a broadcast / swap / negation of a register pair
// (modifiers of the packed instructions), never a pair assembled from two registers, and both squared
// magnitudes are one packed multiply-add.
__device__ __forceinline__ void split_eval(const cf zk, const cf zm, const cf w, cf &R, cf &I)
{
    const cf s = zk + zm;
    const cf d = zk - zm;
    // written out: the compiler assembles (w.y, -w.x) and (T.x, -T.x) with v_xor/v_mov pairs otherwise
    // (one asm statement: no compiler pad between the dependent instructions)
    cf u, tw;
    asm("v_pk_mul_f32 %0, %4, %6 op_sel:[1,0] op_sel_hi:[1,1]\n\t"                                          // u = s.y * w
        "v_pk_fma_f32 %1, %5, %6, %0 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]\n\t"                   // T = u + d.x * (w.y, -w.x)
        "v_pk_add_f32 %2, %4, %1 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]\n\t"                             // R = s.x + (T.x, -T.x)
        "v_pk_add_f32 %3, %5, %1 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]"                                  // I = d.y + (T.y, -T.y)
        : "=&v"(u), "=&v"(tw), "=&v"(R), "=v"(I) : "v"(s), "v"(d), "v"(w));
}

// Output of one group of bins k0..k0+4 (k0 = 4g).  The group evaluates five pairs so that all four
// output streams (bins k, 8192-k and their mirrors 16384-k, 8192+k) leave as aligned 16-byte stores:
//   [k0 .. k0+3] = |P0..3|        [16384-k0-4 .. ] = |P4..1|
//   [8192+k0 ..] = |Q0..3|        [8192-k0-4 ..  ] = |Q4..1|
// Every bin of the frame is written exactly once over the 1024 groups.
template <int OUT>
__device__ __forceinline__ void split_store(const cf (&R)[5], const cf (&I)[5], void *__restrict__ out, int f, int k0)
{
    if constexpr (OUT == SA_OUT_MAG_FULL || OUT == SA_OUT_MAG_HALF) {
        float mp[5], mq[5];
#pragma unroll
        for (int e = 0; e < 5; ++e) {
            const cf m2 = safft::pk_fma(I[e], I[e], R[e] * R[e]);          // (|P|^2, |Q|^2)
            mp[e] = fast_sqrt(m2.x);
            mq[e] = fast_sqrt(m2.y);
        }
        if constexpr (OUT == SA_OUT_MAG_FULL) {
            float *o = reinterpret_cast<float *>(out) + (size_t)f * SA_NPTS;
            store_nt(o + k0, mp[0], mp[1], mp[2], mp[3]);
            store_nt(o + SA_NPTS - k0 - 4, mp[4], mp[3], mp[2], mp[1]);
            store_nt(o + SA_MC + k0, mq[0], mq[1], mq[2], mq[3]);
            store_nt(o + SA_MC - k0 - 4, mq[4], mq[3], mq[2], mq[1]);
        } else {
            float *o = reinterpret_cast<float *>(out) + (size_t)f * (SA_MC + 1);     // rows are not 16-byte aligned
#pragma unroll
            for (int e = 0; e < 4; ++e) store_nt(o + k0 + e, mp[e]);
#pragma unroll
            for (int e = 1; e < 5; ++e) store_nt(o + SA_MC - k0 - e, mq[e]);
            if (k0 == 0) store_nt(o + SA_MC, mq[0]);
        }
    } else {
        float2 *o = reinterpret_cast<float2 *>(out) + (size_t)f * (SA_MC + 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) store_nt(o + k0 + e, R[e].x, I[e].x);
#pragma unroll
        for (int e = 1; e < 5; ++e) store_nt(o + SA_MC - k0 - e, R[e].y, -I[e].y);
        if (k0 == 0) store_nt(o + SA_MC, R[0].y, -I[0].y);
    }
}

}  // namespace
