// chain_q15.hip -- bit-exact integer signal path for gfx950 (MI355X).
//
//   cascade kernels  : Q15 window (new/hann8192.vhd:36-39) + 6-stage integer biquad cascade, in the FPGA-exact Q7
//                      form (filter_q7_kernel: new/filter_iir_cust.vhd:96-117, new/filter_iir12_cust.vhd:68-240) and
//                      in the wide Q2.14 form (filter_w14_kernel: six independent sections, the build's own spec,
//                      oracle/specan_oracle.c:or_iir_sos_q14).  Both recursions are non-linear (per-product
//                      truncation and 16-bit wrap; rounding and saturation), so a frame cannot be cut in
//                      time; the parallelism is batch x section: one frame per 16-lane DPP row, the six sections a
//                      systolic pipeline along the lanes (lane s works on sample n-s).  One cascade wave per SIMD
//                      (a lone wave: one instruction per ~2.5 ns), one helper wave beside it for staging and flushing.
//   sa_fft kernel    : SA-FXFFT-1, the fixed-point FFT that stands where ip/xfft_0 stands
//                      (radix-4 DIF, >>2 per stage, Q15 twiddles, truncation).  One 1024-thread
//                      workgroup per frame, data in LDS as packed (re,im) int16 pairs, Stockham
//                      (autosort) addressing so the result is in natural order, stages paired in registers; the
//                      arithmetic per butterfly is exactly oracle/specan_oracle.c:or_fxfft16k.
#include "sa_common.hpp"
#include <cstdlib>
#include <type_traits>
#include "../../include/specan.h"

#if defined(SA_STAMPS)
// diagnostic build only: per wave {s_memrealtime at start, at end, HW_ID | XCC_ID << 32} (tools/q15_placement.py)
__device__ unsigned long long *g_q15_stamps = nullptr;
extern "C" int sa_debug_set_q15_stamps(void *p)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_q15_stamps), &p, sizeof(p));
}
#define SA_Q15_STAMP_BEGIN(widx)                                                                      \
    const unsigned long long sa_t0_ = __builtin_amdgcn_s_memrealtime();                               \
    const int sa_widx_ = (widx)
#define SA_Q15_STAMP_END()                                                                            \
    do {                                                                                              \
        if ((threadIdx.x & 63) == 0 && g_q15_stamps) {                                                \
            g_q15_stamps[3 * sa_widx_ + 0] = sa_t0_;                                                  \
            g_q15_stamps[3 * sa_widx_ + 1] = __builtin_amdgcn_s_memrealtime();                        \
            g_q15_stamps[3 * sa_widx_ + 2] =                                                          \
                (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |      \
                ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32); \
        }                                                                                             \
    } while (0)
#else
#define SA_Q15_STAMP_BEGIN(widx) do {} while (0)
#define SA_Q15_STAMP_END() do {} while (0)
#endif

namespace {

// ------------------------------------------------------------------------------------------ window
// new/hann8192.vhd:36-39: out = resize16(product(31..15) + product(14)).
// product(31..15) + product(14) = floor((p + 2^14) / 2^15); resize16 of the 17-bit sum keeps its sign bit (16) and its
// low 15 bits, which differs from plain truncation only for +32768 (x = c = -32768), mapped to 0.
__device__ __forceinline__ int win_rtl(int x, int c)
{
    const int r = (x * c + 16384) >> 15;
    return (int)(short)((r & 0x7FFF) | ((r >> 1) & 0x8000));
}

// SURVEY quirk Q2 alternative: ROM + 32768 as unsigned Q16 Hann, round half up.  |x (c + 32768) + 32768| < 2^31 for
// 16-bit x and c (largest 32767 * 65535 + 32768): 32-bit arithmetic is exact.
__device__ __forceinline__ int win_u16(int x, int c)
{
    return (int)(short)((x * (c + 32768) + 32768) >> 16);
}

__device__ __forceinline__ unsigned pack2(int lo, int hi) { return ((unsigned)lo & 0xFFFFu) | ((unsigned)hi << 16); }
// (sat16(lo), sat16(hi)) as one packed dword: v_cvt_pk_i16_i32 saturates and packs in ONE instruction
// (two v_med3_i32, an and and a shift-or otherwise -- a seventh of the kernel's vector instructions)
__device__ __forceinline__ unsigned sat_pack2(int lo, int hi)
{
    typedef short s2 __attribute__((ext_vector_type(2)));
    const s2 r = __builtin_amdgcn_cvt_pk_i16(lo, hi);
    return __builtin_bit_cast(unsigned, r);
}

__device__ __forceinline__ int lo16(unsigned v) { return (int)(short)(v & 0xFFFFu); }
__device__ __forceinline__ int hi16(unsigned v) { return (int)v >> 16; }

// ------------------------------------------------------------------------------------------ IIR
// Tile size: 256 samples.  (Round 3 compiled the cascade a second time with 128-sample tiles for overlapped launches -- half
// the LDS per workgroup, so that two cascades fit beside an FFT workgroup.  With the helper waves the large tiles win at every
// depth -- 10.4 vs 9.9-10.2 M frames/s at depth 2, profiles/r4_q15_helper_waves.txt -- and the second build is gone.)
constexpr int kTile = 256;             // samples per staging tile
constexpr int kRing = 2 * kTile;      // output ring per frame: the pipeline delivers sample T - 5 at step T
constexpr int kRingPitch = kRing + 8;
constexpr int kFramesPerWave = 4;      // one frame per 16-lane row (two frames per wave and two waves per SIMD: 1.47 x slower,
                                       // profiles/r4_int_step_rate.txt)
// moving a tile between memory and LDS: 16 bytes (8 samples) per lane, kTile / 8 lanes per frame row
constexpr int kTileLanes = kTile / 8;                       // lanes that cover one row of a tile
constexpr int kTileRows = 64 / kTileLanes;                  // rows a wave covers per pass
constexpr int kTilePasses = kFramesPerWave / kTileRows;     // passes over the wave's four frames
static_assert(kTile % 32 == 0 && kTileLanes <= 64 && kTileRows * kTilePasses == kFramesPerWave, "tile geometry");

// One biquad step, FPGA-exact Q7 form (new/filter_iir_cust.vhd:96-117):
//   y = B2*x[n] + B1*x[n-1] + B0*x[n-2] - A0*y[n-2] - A1*y[n-1], each product >> 7 (floor), the sum
//   taken modulo 2^16 (wrapping each term first gives the same residue).
// The two subtracted terms use -floor(v/128) = floor((-v + 127)/128), so all five terms add.
// The taps are held pre-shifted by 9: floor(c v / 128) mod 2^16 is then bits 16..31 of the 32-bit product
// v * (c << 9) (exact: only bits above 31 are lost), i.e. its high word, which the SDWA form of v_add_u32
// reads in place -- no shift instructions -- and whose last add sign-extends the 16-bit result on write.
// The products are v_mul_i32_i24 / v_mad_i32_i24 (samples are 16-bit, shifted taps 17-bit: both fit the 24-bit operands).

// one tile of the wave's four frames (and the matching ROM words) on its way from HBM to the input ring: 16 B per lane and pass
struct Q15TileRegs {
    uint4 x[kTilePasses];
    uint4 c[kTilePasses];
};

// issue the global loads of one tile (4 frames x 256 samples and the matching ROM words), 16 B per lane
__device__ __forceinline__ void q15_load_tile(const int16_t *__restrict__ in, const int16_t *__restrict__ rom, int f0,
                                              int batch, int n0, int lane, Q15TileRegs &r)
{
#pragma unroll
    for (int i = 0; i < kTilePasses; ++i) {
        const int row = kTileRows * i + lane / kTileLanes;
        const int col = (lane % kTileLanes) * 8;
        const int f = f0 + row;
        r.x[i] = make_uint4(0, 0, 0, 0);
        if (f < batch) r.x[i] = *reinterpret_cast<const uint4 *>(in + (size_t)f * SA_NPTS + n0 + col);
        r.c[i] = *reinterpret_cast<const uint4 *>(rom + n0 + col);
    }
}

template <int PITCH>
__device__ __forceinline__ void q15_flush_tile(int16_t *__restrict__ out, const int16_t (*src)[PITCH], int src_col,
                                               int f0, int batch, int n0, int lane, int col_mask = 0x7fffffff, int ncols = kTile)
{
#pragma unroll
    for (int i = 0; i < kTilePasses; ++i) {
        const int row = kTileRows * i + lane / kTileLanes;
        const int col = (lane % kTileLanes) * 8;
        const int f = f0 + row;
        const int sc = (src_col + col) & col_mask;      // 8-sample chunks: a ring wraps between chunks only
        uint4 ov;
        ov.x = *reinterpret_cast<const unsigned *>(&src[row][sc + 0]);
        ov.y = *reinterpret_cast<const unsigned *>(&src[row][sc + 2]);
        ov.z = *reinterpret_cast<const unsigned *>(&src[row][sc + 4]);
        ov.w = *reinterpret_cast<const unsigned *>(&src[row][sc + 6]);
        if (f < batch && n0 + col >= 0 && col < ncols) *reinterpret_cast<uint4 *>(out + (size_t)f * SA_NPTS + n0 + col) = ov;
    }
}

// Window only (filter mode 0xB1 through sa_filter_q15: the windowed time series, new/hann8192.vhd:36-39): element-wise,
// 16 bytes (eight samples) of a frame per thread, the ROM words from the L2.
constexpr int kWinThreads = 256;
__global__ __launch_bounds__(kWinThreads) void window_q15_kernel(const int16_t *__restrict__ in, int16_t *__restrict__ out, int batch,
                                                                  SaQ15Params prm, const int16_t *__restrict__ rom)
{
    const size_t chunk = (size_t)blockIdx.x * kWinThreads + threadIdx.x;       // 16-byte chunk of the batch
    if (chunk >= (size_t)batch * (SA_NPTS / 8)) return;
    const int col = (int)(chunk % (SA_NPTS / 8)) * 8;
    const uint4 xv = *reinterpret_cast<const uint4 *>(in + chunk * 8);
    const uint4 cv = *reinterpret_cast<const uint4 *>(rom + col);
    const unsigned xs[4] = {xv.x, xv.y, xv.z, xv.w}, cs[4] = {cv.x, cv.y, cv.z, cv.w};
    unsigned o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int a, b;
        if (prm.win_mode == SA_WIN_RTL_SIGNED) {
            a = win_rtl(lo16(xs[q]), lo16(cs[q]));
            b = win_rtl(hi16(xs[q]), hi16(cs[q]));
        } else {
            a = win_u16(lo16(xs[q]), lo16(cs[q]));
            b = win_u16(hi16(xs[q]), hi16(cs[q]));
        }
        o[q] = pack2(a, b);
    }
    *reinterpret_cast<uint4 *>(out + chunk * 8) = make_uint4(o[0], o[1], o[2], o[3]);
}

// ------------------------------------------------------------------------------------------ IIR, Q7 (FPGA-exact)
// The cascade is organised around what bounds it: the recursion is serial in time, there is one wave per SIMD at
// B = 4096, and a lone wave issues one vector instruction per ~2.5 ns whatever the instruction is
// (profiles/r4_int_step_rate.txt) -- so the time per frame is 16 392 steps x the vector instructions of a step.
//   * the three feed-forward products come straight from the LEFT NEIGHBOUR's output registers
//     (v_mul_i32_i24_dpp row_ror:1): no cross-lane move, no x[n-1] / x[n-2] history registers;
//   * the input travels through the lanes the cascade leaves idle: lanes 9..15 and 0 of the 16-lane row are identity
//     stages forming a shift register, refilled with eight samples by ONE 16-bit LDS read and one select per eight steps;
//   * every step is one asm block in a fixed order, so the DPP read of a register the neighbour has just written always
//     has the two wait states the hardware asks for (the compiler cannot see into an asm block and does not pad);
//   * four cascade waves per workgroup (16 frames): the dispatcher places the waves of one workgroup on the four SIMDs of one
//     CU, 256 workgroups = one per CU at B = 4096.  (1 024 one-wave workgroups land two to a SIMD on part of the chip whenever
//     another kernel ran before: 673 us back to back, 943 us after anything else -- profiles/r2_q15_placement.txt.)
//   * four HELPER waves per workgroup (waves 4..7, one beside each cascade wave) do the staging and the flushing
//     (q15_helper_wave): a lone wave's unused issue turns are the only place where that work costs nothing.
// Lanes of a row: 0 = input, 1..6 = sections 0..5, 7..8 = delay (lane 8 emits sample T - 8 at step T: every group of 8
// steps ends with 8 consecutive, 16-byte aligned outputs), 9..15 = input shift register.  9 vector instructions per step,
// 7 when tap B1 is zero in both coefficient sets.  What was tried on this step and did not pay: Appendix B of DESIGN.md.
constexpr int kV2Waves = 4;                       // cascade waves per workgroup: one per SIMD of the CU
constexpr int kWgWaves = 2 * kV2Waves;            // + one helper wave per cascade wave (staging and flushing, see q15_helper_wave)
constexpr unsigned long long kOutMask = 0x0100010001000100ull;   // lane 8 of every row
constexpr int kInRing = 2 * kTile;                // input ring per frame: the tile in use + the one before it
constexpr int kInPitch = kInRing + 8;

// window the loaded tile and put it into its half (col0 = 0 or kTile) of the wave's input ring: 16 bytes per lane
__device__ __forceinline__ void q15_window_into_ring(const Q15TileRegs &r, int16_t (*dst)[kInPitch], int col0, int lane, int win_mode)
{
#pragma unroll
    for (int i = 0; i < kTilePasses; ++i) {
        const int row = kTileRows * i + lane / kTileLanes;
        const int col = col0 + (lane % kTileLanes) * 8;
        const unsigned xs[4] = {r.x[i].x, r.x[i].y, r.x[i].z, r.x[i].w};
        const unsigned cs[4] = {r.c[i].x, r.c[i].y, r.c[i].z, r.c[i].w};
        unsigned o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int a, b;
            if (win_mode == SA_WIN_RTL_SIGNED) {
                a = win_rtl(lo16(xs[q]), lo16(cs[q]));
                b = win_rtl(hi16(xs[q]), hi16(cs[q]));
            } else {
                a = win_u16(lo16(xs[q]), lo16(cs[q]));
                b = win_u16(hi16(xs[q]), hi16(cs[q]));
            }
            o[q] = pack2(a, b);
        }
        *reinterpret_cast<uint4 *>(&dst[row][col]) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// One block = one time step, written in the cyclic order the issue logic likes best.  With x = the LEFT neighbour's
// output (read in place by the DPP forms), y = the lane's own, and per step
//     y[n] = s2 + t,    t = t(B2, x[n]) - t(A1, y[n-1]),    s2 = t(B1, x[n-1]) + t(B0, x[n-2]) - t(A0, y[n-2])
// the block FIRST finishes y[n] from the t and s2 the previous block prepared (H), then starts everything of
// y[n+1] and y[n+2] that hangs on it:
//     H  y   = (int16)(s2 + t)                 G  u  = hi(p1) + hi(p2)        (terms of the block before)
//     B  p2  = x[n-1] * B0   (dpp)             C  p0 = x[n] * B2   (dpp; the neighbour's H is 3 instructions old)
//     A  p4  = y * -A1 + k                     I  s2 = u + hi(p3)
//     D  p1  = x[n] * B1     (dpp)             F  t  = hi(p0) + hi(p4)
//     E  p3  = y * -A0 + k
// No instruction reads a register written by either of the two instructions in front of it except F -> H of the
// next block (one in between); the DPP read of the y just written has two instructions in between, which is what
// the hardware asks for (the compiler cannot see into an asm block and must not be relied on to pad).
// tools/ubench/q7_step_rate.hip times these nine instructions alone, one wave per SIMD: 19.5 ns per step in this
// order, 24.1 ns in the order A B C D E F G H I (y finished second to last, its first reader one instruction later).
#define SA_Q7_BLOCK(Y, H1)                                                                                             \
    "v_add_u32_sdwa %[" Y "], %[s2], %[t] dst_sel:WORD_0 dst_unused:UNUSED_SEXT src0_sel:DWORD src1_sel:DWORD\n\t"     \
    "v_add_u32_sdwa %[u], %[p1], %[p2] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"        \
    "v_mul_i32_i24_dpp %[p2], %[" H1 "], %[cB0] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"                              \
    "v_mul_i32_i24_dpp %[p0], %[" Y "], %[cB2] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"                               \
    "v_mad_i32_i24 %[p4], %[" Y "], %[nA1], %[k]\n\t"                                                                  \
    "v_add_u32_sdwa %[s2], %[u], %[p3] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"         \
    "v_mul_i32_i24_dpp %[p1], %[" Y "], %[cB1] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"                               \
    "v_add_u32_sdwa %[t], %[p0], %[p4] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"        \
    "v_mad_i32_i24 %[p3], %[" Y "], %[nA0], %[k]"

// what one block hands to the next (zero history = all zero: hi(k) = 0)
struct Q7Carry {
    int p0, p1, p2, p3, p4, t, u, s2;
};

__device__ __forceinline__ void q7_block(int &y, Q7Carry &c, int y7, int cB2, int cB1, int cB0, int nA0, int nA1, int k127)
{
    int y0;
    asm volatile(SA_Q7_BLOCK("y0", "y7")
                 : [y0] "=&v"(y0), [s2] "+v"(c.s2), [p0] "+v"(c.p0), [p1] "+v"(c.p1), [p2] "+v"(c.p2), [p3] "+v"(c.p3),
                   [p4] "+v"(c.p4), [t] "+v"(c.t), [u] "+v"(c.u)
                 : [y7] "v"(y7), [cB2] "v"(cB2), [cB1] "v"(cB1), [cB0] "v"(cB0), [nA0] "v"(nA0), [nA1] "v"(nA1), [k] "s"(k127));
    y = y0;
}

typedef unsigned q7_u4 __attribute__((ext_vector_type(4)));

// 16-byte LDS store by the lanes of `mask` only, without a branch around it (the compiler's form is a
// saveexec / skip-branch / restore triple laid out of line: two taken branches per eight steps)
__device__ __forceinline__ void lds_store16_masked(unsigned addr, q7_u4 v, unsigned long long mask)
{
    unsigned long long saved;
    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                 "ds_write_b128 %[a], %[d]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(saved) : [m] "s"(mask), [a] "v"(addr), [d] "v"(v) : "memory", "scc");
}

// The 32 groups of a tile as ONE asm statement.  A lone wave issues at most one instruction per turn of its SIMD
// and stalls whole turns; which turns are lost depends on where the 8-byte instructions lie relative to the
// instruction fetch (tools/ubench/q7_nop_sweep.py: the same nine instructions run 19.4, 21.7 or 24.1 ns per step
// depending on a 4-byte s_nop in front of them or between them), so the loop is pinned: 64-byte aligned, nothing of
// the compiler's inside it (fourteen placements of a 4-byte `s_nop 0` inside the block were timed on the real kernel,
// 424-460 us: none beats the unpadded block in the order H G B C A I D F E).  Per group: select the refill into t, request the next refill (16-bit LDS read, used
// one group later: lgkmcnt(2) = everything but the two stores behind it), eight blocks, lane 8's eight outputs
// stored as dwords under an exec mask (restored five instructions before the next DPP read, as the hardware asks).
#define SA_Q7_TBLOCK(Y, H1)                                                                                            \
    "v_add_u32_sdwa " Y ", %[s2], %[t] dst_sel:WORD_0 dst_unused:UNUSED_SEXT src0_sel:DWORD src1_sel:DWORD\n\t"   \
    "v_add_u32_sdwa %[u], %[p1], %[p2] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"  \
    "v_mul_i32_i24_dpp %[p2], " H1 ", %[cB0] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"                           \
    "v_mul_i32_i24_dpp %[p0], " Y ", %[cB2] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"                            \
    "v_mad_i32_i24 %[p4], " Y ", %[nA1], %[k]\n\t"                                                               \
    "v_add_u32_sdwa %[s2], %[u], %[p3] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"   \
    "v_mul_i32_i24_dpp %[p1], " Y ", %[cB1] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"                            \
    "v_add_u32_sdwa %[t], %[p0], %[p4] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"  \
    "v_mad_i32_i24 %[p3], " Y ", %[nA0], %[k]\n\t"
// The lane's last eight outputs live in v52..v59 inside the loop (named registers: the two 16-byte stores of lane 8
// need them consecutive, and an asm operand cannot be addressed by sub-register); they are stored as sign-extended
// dwords, the saturating pack to int16 happens once per sample in the flush, where all 64 lanes have work, instead
// of four times per group here, where lane 8 of each row is the only one with a use for it.
#define SA_Q7_TGROUP(RD_OFF, WR_OFF0, WR_OFF1)                                                                         \
    "s_waitcnt lgkmcnt(2)\n\t"                                                                                         \
    "v_cndmask_b32_e64 %[t], %[t], %[xin], %[inm]\n\t"                                                                 \
    "ds_read_u16 %[xin], %[xa] offset:" RD_OFF "\n\t"                                                                  \
    SA_Q7_TBLOCK("v52", "v59") SA_Q7_TBLOCK("v53", "v52") SA_Q7_TBLOCK("v54", "v53") SA_Q7_TBLOCK("v55", "v54")        \
    SA_Q7_TBLOCK("v56", "v55") SA_Q7_TBLOCK("v57", "v56") SA_Q7_TBLOCK("v58", "v57") SA_Q7_TBLOCK("v59", "v58")        \
    "s_and_saveexec_b64 %[sv], %[outm]\n\t"                                                                            \
    "ds_write_b128 %[ra], v[52:55] offset:" WR_OFF0 "\n\t"                                                             \
    "ds_write_b128 %[ra], v[56:59] offset:" WR_OFF1 "\n\t"                                                             \
    "s_mov_b64 exec, %[sv]\n\t"

// The same block when the port tap B1 is zero in BOTH coefficient sets (the fixed ALPHA / BETA cascade of mode 0x00,
// imp/filter_pkg.vhd:54-68; the identity stages have B1 = 0 anyway): t(0, v) = 0 exactly, so the product D and the add G
// drop out and s2 = hi(p2) + hi(p3) is one instruction, issued BEFORE this block's B and E overwrite the previous
// block's p2 and p3.  Seven instructions per step, bit-identical results (new/filter_iir_cust.vhd:96-117 truncates every
// product separately: a zero tap contributes a zero term).  The host picks this form per launch from the coefficient
// bytes (sa_launch_filter_q15); any other upload runs the nine-instruction block.
#define SA_Q7I_H(Y) "v_add_u32_sdwa " Y ", %[s2], %[t] dst_sel:WORD_0 dst_unused:UNUSED_SEXT src0_sel:DWORD src1_sel:DWORD\n\t"
#define SA_Q7I_I "v_add_u32_sdwa %[s2], %[p2], %[p3] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"
#define SA_Q7I_B(H1) "v_mul_i32_i24_dpp %[p2], " H1 ", %[cB0] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
#define SA_Q7I_C(Y) "v_mul_i32_i24_dpp %[p0], " Y ", %[cB2] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
#define SA_Q7I_A(Y) "v_mad_i32_i24 %[p4], " Y ", %[nA1], %[k]\n\t"
#define SA_Q7I_E(Y) "v_mad_i32_i24 %[p3], " Y ", %[nA0], %[k]\n\t"
#define SA_Q7I_F "v_add_u32_sdwa %[t], %[p0], %[p4] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"
// order H I B A C E F (six orders and every single s_nop position were timed on the real kernel: this one, unpadded, is
// the fastest; profiles/r3_fuzz_and_soak.txt, gpurun_out/ab_q7_nb1*.log)
#define SA_Q7_NB1BLOCK(Y, H1) SA_Q7I_H(Y) SA_Q7I_I SA_Q7I_B(H1) SA_Q7I_A(Y) SA_Q7I_C(Y) SA_Q7I_E(Y) SA_Q7I_F
#define SA_Q7_NB1GROUP(RD_OFF, WR_OFF0, WR_OFF1)                                                                       \
    "s_waitcnt lgkmcnt(2)\n\t"                                                                                         \
    "v_cndmask_b32_e64 %[t], %[t], %[xin], %[inm]\n\t"                                                                 \
    "ds_read_u16 %[xin], %[xa] offset:" RD_OFF "\n\t"                                                                  \
    SA_Q7_NB1BLOCK("v52", "v59") SA_Q7_NB1BLOCK("v53", "v52") SA_Q7_NB1BLOCK("v54", "v53") SA_Q7_NB1BLOCK("v55", "v54")  \
    SA_Q7_NB1BLOCK("v56", "v55") SA_Q7_NB1BLOCK("v57", "v56") SA_Q7_NB1BLOCK("v58", "v57") SA_Q7_NB1BLOCK("v59", "v58")  \
    "s_and_saveexec_b64 %[sv], %[outm]\n\t"                                                                            \
    "ds_write_b128 %[ra], v[52:55] offset:" WR_OFF0 "\n\t"                                                             \
    "ds_write_b128 %[ra], v[56:59] offset:" WR_OFF1 "\n\t"                                                             \
    "s_mov_b64 exec, %[sv]\n\t"

// xa: LDS byte address of the lane's refill slot of the tile's first group; ra: of the ring slot of its outputs
template <bool NOB1>
__device__ __forceinline__ void q7_tile(int (&y)[8], Q7Carry &c, unsigned xa, unsigned ra, unsigned long long in_mask,
                                        unsigned long long out_mask, int cB2, int cB1, int cB0, int nA0, int nA1, int k127)
{
    int xin, cnt = kTile / 32;
    unsigned long long saved;
    if constexpr (NOB1) {
        // the seven-instruction block: p1 and u are not touched (p1 stays the zero it is: the drain group's block reads it)
        asm volatile(
            "v_mov_b32 v52, %[y0]\n\tv_mov_b32 v53, %[y1]\n\tv_mov_b32 v54, %[y2]\n\tv_mov_b32 v55, %[y3]\n\t"
            "v_mov_b32 v56, %[y4]\n\tv_mov_b32 v57, %[y5]\n\tv_mov_b32 v58, %[y6]\n\tv_mov_b32 v59, %[y7]\n\t"
            "ds_read_u16 %[xin], %[xa]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            ".p2align 6\n"
            "1:\n\t"
            SA_Q7_NB1GROUP("16", "0", "16") SA_Q7_NB1GROUP("32", "32", "48") SA_Q7_NB1GROUP("48", "64", "80") SA_Q7_NB1GROUP("64", "96", "112")
            "v_add_u32 %[xa], 64, %[xa]\n\t"
            "v_add_u32 %[ra], 0x80, %[ra]\n\t"
            "s_add_i32 %[cnt], %[cnt], -1\n\t"
            "s_cmp_lg_u32 %[cnt], 0\n\t"
            "s_cbranch_scc1 1b\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mov_b32 %[y0], v52\n\tv_mov_b32 %[y1], v53\n\tv_mov_b32 %[y2], v54\n\tv_mov_b32 %[y3], v55\n\t"
            "v_mov_b32 %[y4], v56\n\tv_mov_b32 %[y5], v57\n\tv_mov_b32 %[y6], v58\n\tv_mov_b32 %[y7], v59"
            : [y0] "+v"(y[0]), [y1] "+v"(y[1]), [y2] "+v"(y[2]), [y3] "+v"(y[3]), [y4] "+v"(y[4]), [y5] "+v"(y[5]), [y6] "+v"(y[6]),
              [y7] "+v"(y[7]), [s2] "+v"(c.s2), [p0] "+v"(c.p0), [p2] "+v"(c.p2), [p3] "+v"(c.p3), [p4] "+v"(c.p4),
              [t] "+v"(c.t), [xa] "+v"(xa), [ra] "+v"(ra), [xin] "=&v"(xin), [cnt] "+s"(cnt), [sv] "=&s"(saved)
            : [cB2] "v"(cB2), [cB0] "v"(cB0), [nA0] "v"(nA0), [nA1] "v"(nA1), [k] "s"(k127), [inm] "s"(in_mask), [outm] "s"(out_mask)
            : "memory", "scc", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59");
        // hand-over to the nine-instruction block of the drain group: its G adds hi(p1) = 0 and hi(p2) of the last B
        // into u BEFORE its I -- consistent with what the last I of this loop left in s2 only if u is rebuilt the same way,
        // which G does from p1, p2 itself.  Nothing to fix up.
        return;
    }
    asm volatile(
        "v_mov_b32 v52, %[y0]\n\tv_mov_b32 v53, %[y1]\n\tv_mov_b32 v54, %[y2]\n\tv_mov_b32 v55, %[y3]\n\t"
        "v_mov_b32 v56, %[y4]\n\tv_mov_b32 v57, %[y5]\n\tv_mov_b32 v58, %[y6]\n\tv_mov_b32 v59, %[y7]\n\t"
        "ds_read_u16 %[xin], %[xa]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"                      // first group: nothing in flight, lgkmcnt(2) passes
        ".p2align 6\n"
        "1:\n\t"
        SA_Q7_TGROUP("16", "0", "16") SA_Q7_TGROUP("32", "32", "48") SA_Q7_TGROUP("48", "64", "80") SA_Q7_TGROUP("64", "96", "112")
        "v_add_u32 %[xa], 64, %[xa]\n\t"
        "v_add_u32 %[ra], 0x80, %[ra]\n\t"
        "s_add_i32 %[cnt], %[cnt], -1\n\t"
        "s_cmp_lg_u32 %[cnt], 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mov_b32 %[y0], v52\n\tv_mov_b32 %[y1], v53\n\tv_mov_b32 %[y2], v54\n\tv_mov_b32 %[y3], v55\n\t"
        "v_mov_b32 %[y4], v56\n\tv_mov_b32 %[y5], v57\n\tv_mov_b32 %[y6], v58\n\tv_mov_b32 %[y7], v59"
        : [y0] "+v"(y[0]), [y1] "+v"(y[1]), [y2] "+v"(y[2]), [y3] "+v"(y[3]), [y4] "+v"(y[4]), [y5] "+v"(y[5]), [y6] "+v"(y[6]),
          [y7] "+v"(y[7]), [s2] "+v"(c.s2), [p0] "+v"(c.p0), [p1] "+v"(c.p1), [p2] "+v"(c.p2), [p3] "+v"(c.p3), [p4] "+v"(c.p4),
          [t] "+v"(c.t), [u] "+v"(c.u), [xa] "+v"(xa), [ra] "+v"(ra), [xin] "=&v"(xin), [cnt] "+s"(cnt), [sv] "=&s"(saved)
        : [cB2] "v"(cB2), [cB1] "v"(cB1), [cB0] "v"(cB0), [nA0] "v"(nA0), [nA1] "v"(nA1), [k] "s"(k127), [inm] "s"(in_mask),
          [outm] "s"(out_mask)
        : "memory", "scc", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59");
}

// flush one tile of the dword ring: 8 samples per lane, packed to int16 with saturation (exact: the values are
// sign-extended 16-bit numbers) and stored as 16 bytes
__device__ __forceinline__ void q7_flush_tile(int16_t *__restrict__ out, const int (*src)[kRingPitch], int src_col, int f0,
                                              int batch, int n0, int lane, int ncols = kTile)
{
    // n0 may be -8 (the helper's spans start eight samples early: q7 helper waves below) and ncols may be 8 (the tail)
#pragma unroll
    for (int i = 0; i < kTilePasses; ++i) {
        const int row = kTileRows * i + lane / kTileLanes;
        const int col = (lane % kTileLanes) * 8;
        const int f = f0 + row;
        const int sc = (src_col + col) & (kRing - 1);       // 8-sample chunks: the ring wraps between chunks only
        const int4 a = *reinterpret_cast<const int4 *>(&src[row][sc]);
        const int4 b = *reinterpret_cast<const int4 *>(&src[row][sc + 4]);
        if (f < batch && n0 + col >= 0 && col < ncols)
            *reinterpret_cast<uint4 *>(out + (size_t)f * SA_NPTS + n0 + col) =
                make_uint4(sat_pack2(a.x, a.y), sat_pack2(a.z, a.w), sat_pack2(b.x, b.y), sat_pack2(b.z, b.w));
    }
}

__device__ __forceinline__ void wave_lds_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
// workgroup barrier that waits for the wave's LDS traffic only (__syncthreads would also wait for the global loads of the
// tile after next and for the stores of the tile before: exactly what is meant to stay in flight)
__device__ __forceinline__ void wg_lds_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The helper wave of a cascade wave (same frames, same SIMD): everything that is not the recursion.  A lone wave issues one
// vector instruction per ~2.5 ns and leaves the rest of its SIMD's turns unused; staging and flushing on the cascade wave
// itself cost 0.6 instructions per step = 7-9 % of the kernel (filter_q7_kernel<true> 349 -> 325-331 us with this wave,
// profiles/r4_q15_helper_waves.txt).  During the cascade's tile k the helper windows tile k + 1 into the other half of the
// input ring, requests tile k + 2 from HBM and flushes the outputs of the tile before -- shifted by the pipeline's eight
// samples of delay, so that a flush covers exactly one half of the output ring (samples [kTile j - 8, kTile (j+1) - 8) live in
// slots [kTile j, kTile (j+1)) mod kRing) while the cascade writes the other half.  One workgroup barrier per tile (both sides
// wait for their LDS traffic only); the cascade side runs nt tiles, the drain, and the same nt + 2 barriers.
//   flush(j, n0, ncols): store ncols samples starting at sample n0 (may be -8: skipped) from ring slots kTile j ...
template <typename Flush>
__device__ __forceinline__ void q15_helper_wave(Flush flush, const int16_t *__restrict__ in, const int16_t *__restrict__ rom,
                                                int16_t (*tin)[kInPitch], int f0, int batch, int lane, int win_mode, bool idle)
{
    constexpr int nt = SA_NPTS / kTile;
    Q15TileRegs pre;
    if (!idle) {
        q15_load_tile(in, rom, f0, batch, 0, lane, pre);
        q15_window_into_ring(pre, tin, 0, lane, win_mode);
        q15_load_tile(in, rom, f0, batch, kTile, lane, pre);
    }
    wg_lds_sync();
    for (int k = 0; k <= nt; ++k) {                        // k = nt: the cascade runs its drain
        if (!idle) {
            if (k + 1 < nt) q15_window_into_ring(pre, tin, ((k + 1) & 1) * kTile, lane, win_mode);
            if (k + 2 < nt) q15_load_tile(in, rom, f0, batch, (k + 2) * kTile, lane, pre);
            if (k >= 1) flush(k - 1, (k - 1) * kTile - 8, kTile);
        }
        wg_lds_sync();
    }
    if (!idle) flush(nt, nt * kTile - 8, 8);               // the drain's eight samples
}

template <bool NOB1>
__global__ __launch_bounds__(64 * kWgWaves) void filter_q7_kernel(const int16_t *__restrict__ in, int16_t *__restrict__ out,
                                                                   int batch, SaQ15Params prm, const int16_t *__restrict__ rom)
{
    __shared__ __attribute__((aligned(16))) int16_t tin_all[kV2Waves][kFramesPerWave][kInPitch];
    __shared__ __attribute__((aligned(16))) int ring_all[kV2Waves][kFramesPerWave][kRingPitch];     // outputs as dwords
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wid & (kV2Waves - 1);
    const bool helper = wid >= kV2Waves;                   // wave 4 + w stages and flushes for cascade wave w
    const int lane = threadIdx.x & 63;
    int16_t (*tin)[kInPitch] = tin_all[wave];
    int (*ring)[kRingPitch] = ring_all[wave];
    const int fr = lane >> 4;           // frame slot in this wave
    const int l16 = lane & 15;          // role inside the row
    const int f0 = (blockIdx.x * kV2Waves + wave) * kFramesPerWave;
    constexpr int nt = SA_NPTS / kTile;
    const bool idle = f0 >= batch;      // a pair without frames still takes part in the workgroup's barriers
    if (helper) {
        q15_helper_wave([&](int j, int n0, int ncols) { q7_flush_tile(out, ring, j * kTile, f0, batch, n0, lane, ncols); }, in, rom, tin,
                        f0, batch, lane, prm.win_mode, idle);
        return;
    }
    SA_Q15_STAMP_BEGIN(blockIdx.x * kV2Waves + wave);

    // taps, pre-shifted by 9; identity = (128 x) >> 7
    int cB2 = 128 << 9, cB1 = 0, cB0 = 0, nA0 = 0, nA1 = 0;
    if (l16 >= 1 && l16 <= 6) {
        const int sec = l16 - 1;
        const int8_t *c = &prm.c12[(sec & 1) ? 6 : 0];   // stages 1,3,5 = set 0; 2,4,6 = set 1
        cB0 = c[0] << 9; cB1 = c[1] << 9; cB2 = c[2] << 9; nA0 = -(c[3] << 9); nA1 = -(c[4] << 9);
    }
    const int k127 = 127 << 9;
    // input shift register: lanes 0, 15, 14, .., 9 take samples T0, T0+1, .., T0+7 as their y[T0] (through t: they
    // are identity stages, s2 = 0) at the start of a group
    const int kin = (16 - l16) & 15;                       // 0..7 for the input lanes
    const bool is_in = kin < 8;
    const unsigned long long out_mask = kOutMask;                  // lane 8 of every row that carries a frame
    const unsigned long long in_mask = 0xFE01FE01FE01FE01ull;    // lanes 0 and 9..15
    const uint16_t *xrow = reinterpret_cast<const uint16_t *>(&tin[fr][0]) + (is_in ? kin : 0);
    const unsigned xrow_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const uint16_t *)xrow;
    const unsigned ring_addr = (unsigned)(size_t)(__attribute__((address_space(3))) int *)(&ring[fr][0]);

    int y[8] = {0, 0, 0, 0, 0, 0, 0, 0};                   // the lane's last eight outputs
    Q7Carry c = {0, 0, 0, 0, 0, 0, 0, 0};

    // eight steps in the compiler's hands (the drain group only; the tiles run q7_tile).  Lane 8 then holds
    // samples T0 - 8 .. T0 - 1; sample m lives in ring slot (m + 8) mod kRing, so that the groups of one tile store
    // to consecutive slots (the first group of the frame stores eight zeros into slots nobody reads).  The values
    // are sign-extended 16-bit numbers: the saturating pack is exact and one instruction per pair.
    auto group = [&](int T0, int xin) {
        c.t = is_in ? xin : c.t;
#pragma unroll
        for (int e = 0; e < 8; ++e) q7_block(y[e], c, y[(e + 7) & 7], cB2, cB1, cB0, nA0, nA1, k127);
        q7_u4 va, vb;
        va.x = y[0]; va.y = y[1]; va.z = y[2]; va.w = y[3];
        vb.x = y[4]; vb.y = y[5]; vb.z = y[6]; vb.w = y[7];
        lds_store16_masked(ring_addr + 4 * (T0 & (kRing - 1)), va, out_mask);
        lds_store16_masked(ring_addr + 4 * (T0 & (kRing - 1)) + 16, vb, out_mask);
    };

    wg_lds_sync();
    for (int k = 0; k < nt; ++k) {
        const int i0 = (k & 1) * kTile;
        if (!idle) q7_tile<NOB1>(y, c, xrow_addr + 2 * i0, ring_addr + 4 * i0, in_mask, out_mask, cB2, cB1, cB0, nA0, nA1, k127);
        wg_lds_sync();
    }
    if (!idle) group(nt * kTile, xrow[0]);                  // drains the pipeline: the frame's last eight samples
    wg_lds_sync();
    SA_Q15_STAMP_END();
}

// ------------------------------------------------------------------------------------------ IIR, wide Q2.14 form
// Mode 0xA2 (the build's own spec, oracle/specan_oracle.c:or_iir_sos_q14; the six sections scripts/fft_analyzer_gui.py:108-157
// designs and :1186-1192 cuts down to two): per section, direct form I,
//     y[n] = sat16( (b0 x[n] + b1 x[n-1] + b2 x[n-2] - a1 y[n-1] - a2 y[n-2] + 8192) >> 14 ),
// int16 taps and samples.  The exact sum needs 34 bits.  It is kept in two 32-bit accumulators without a single 64-bit
// instruction: every tap (and every negated feedback tap, -a in [-32767, 32768]) splits as c = 2^14 ch + cl with
// cl in [-8192, 8191] and ch in {-2..2}, so
//     acc_l = 8192 + sum cl v   (|acc_l| <= 5 * 2^13 * 2^15 + 2^13 < 2^31, no wrap at any partial sum)
//     acc_h =        sum ch v   (|acc_h| <= 10 * 2^15)
//     (acc + 8192) >> 14 = acc_h + (acc_l >> 14)            exactly (acc_h is an integer: the floor passes it)
// and the two products a packed-int16 dot product forms per instruction halve the count: with the lane's last two outputs
// and the neighbour's last two outputs held as packed pairs P = (lo: y[n-1], hi: y[n]), one step is
//     acc  = dot2(P_own[n-1], (-a2, -a1), 8192 | 0)          v_dot2_i32_i16, low and high half: 2 instructions
//     X    = P_neighbour[n-1]                                v_mov_b32_dpp row_ror:1 = (x[n-1], x[n])
//     acc += dot2(X, (b1, b0));  acc += dot2(X_prev, (b2, 0))                                   4 instructions
//     w    = acc_h + (acc_l >> 14)                           v_ashrrev_i32, v_add_u32
//     P_own[n] = (sat16(w[n-1]), sat16(w[n]))                v_cvt_pk_i16_i32: saturation and packing in ONE instruction
// = 10 vector instructions per step against 9 of the Q7 form (filter_q7_kernel<false>) and about 40 of the round-1 form
// (five 64-bit multiply-adds, a 64-bit shift, two compares and selects, three cross-lane moves).  Everything else is
// the Q7 kernel's structure: four waves per workgroup and one workgroup per CU at B = 4096 (one wave per SIMD in every
// placement scenario, profiles/r2_q15_placement.txt), lanes 0 and 9..15 of the row an input shift register refilled by
// one 16-bit LDS read per eight steps (identity sections: b0 = 16384 gives w = x exactly), lanes 7..8 delay stages so
// that lane 8 emits sample T - 8 at step T, the tile loop ONE pinned asm statement.  The outputs leave as packed int16
// (the pairs of the odd steps, v[52:55]: one 16-byte LDS store per eight steps), so the output ring is half the Q7 kernel's.
struct W14Taps {
    unsigned c01l, c01h, c2l, c2h, cfbl, cfbh;      // packed (lo, hi) int16 pairs: (b1, b0), (b2, 0), (-a2, -a1); low / high split
};
__device__ __forceinline__ void w14_split(int c, int &cl, int &ch)
{
    cl = ((c + 8192) & 16383) - 8192;
    ch = (c - cl) >> 14;
}
__device__ __forceinline__ W14Taps w14_taps(int b0, int b1, int b2, int a1, int a2)
{
    int b0l, b0h, b1l, b1h, b2l, b2h, n1l, n1h, n2l, n2h;
    w14_split(b0, b0l, b0h); w14_split(b1, b1l, b1h); w14_split(b2, b2l, b2h);
    w14_split(-a1, n1l, n1h); w14_split(-a2, n2l, n2h);
    W14Taps t;
    t.c01l = pack2(b1l, b0l); t.c01h = pack2(b1h, b0h);
    t.c2l = pack2(b2l, 0);    t.c2h = pack2(b2h, 0);
    t.cfbl = pack2(n2l, n1l); t.cfbh = pack2(n2h, n1h);
    return t;
}

// what one tile hands to the next: the eight pair registers of the group, the two unsaturated outputs and the two
// neighbour pairs the next block reads
struct W14Carry {
    unsigned p[8];
    int w0, w1;
    unsigned x0, x1;
};

// block e of a group: PP = the pair of block e - 1, PC = this block's, (XC, XP) = the neighbour pair registers of this /
// the previous block, (WC, WP) = the unsaturated outputs likewise.  SEL: the refill select of the group's first block.
#define SA_W14_BLOCK(PP, PC, XC, XP, WC, WP, SEL)                                                                      \
    "v_dot2_i32_i16 %[al], " PP ", %[cfbl], %[k]\n\t"                                                                  \
    "v_dot2_i32_i16 %[ah], " PP ", %[cfbh], 0\n\t"                                                                     \
    "v_mov_b32_dpp " XC ", " PP " row_ror:1 row_mask:0xf bank_mask:0xf\n\t"                                            \
    "v_dot2_i32_i16 %[al], " XC ", %[c01l], %[al]\n\t"                                                                 \
    "v_dot2_i32_i16 %[ah], " XC ", %[c01h], %[ah]\n\t"                                                                 \
    "v_dot2_i32_i16 %[al], " XP ", %[c2l], %[al]\n\t"                                                                  \
    "v_dot2_i32_i16 %[ah], " XP ", %[c2h], %[ah]\n\t"                                                                  \
    "v_ashrrev_i32 %[al], 14, %[al]\n\t"                                                                               \
    "v_add_u32 " WC ", %[ah], %[al]\n\t" SEL                                                                           \
    "v_cvt_pk_i16_i32 " PC ", " WP ", " WC "\n\t"
// pair registers: odd blocks v52..v55 (what lane 8 stores), even blocks v56..v59
#define SA_W14_GROUP(RD_OFF, WR_OFF)                                                                                   \
    "s_waitcnt lgkmcnt(1)\n\t"                                                                                         \
    SA_W14_BLOCK("v55", "v56", "%[x0]", "%[x1]", "%[w0]", "%[w1]", "v_cndmask_b32_e64 %[w0], %[w0], %[xin], %[inm]\n\t") \
    "ds_read_i16 %[xin], %[xa] offset:" RD_OFF "\n\t"                                                                  \
    SA_W14_BLOCK("v56", "v52", "%[x1]", "%[x0]", "%[w1]", "%[w0]", "")                                                 \
    SA_W14_BLOCK("v52", "v57", "%[x0]", "%[x1]", "%[w0]", "%[w1]", "")                                                 \
    SA_W14_BLOCK("v57", "v53", "%[x1]", "%[x0]", "%[w1]", "%[w0]", "")                                                 \
    SA_W14_BLOCK("v53", "v58", "%[x0]", "%[x1]", "%[w0]", "%[w1]", "")                                                 \
    SA_W14_BLOCK("v58", "v54", "%[x1]", "%[x0]", "%[w1]", "%[w0]", "")                                                 \
    SA_W14_BLOCK("v54", "v59", "%[x0]", "%[x1]", "%[w0]", "%[w1]", "")                                                 \
    SA_W14_BLOCK("v59", "v55", "%[x1]", "%[x0]", "%[w1]", "%[w0]", "")                                                 \
    "s_and_saveexec_b64 %[sv], %[outm]\n\t"                                                                            \
    "ds_write_b128 %[ra], v[52:55] offset:" WR_OFF "\n\t"                                                              \
    "s_mov_b64 exec, %[sv]\n\t"

// `iters` x 4 groups x 8 steps.  xa: LDS byte address of the lane's refill slot of the first group; ra: of the ring slot
// of its outputs.  The refill is requested one group ahead (lgkmcnt(1): everything but the store behind it).
__device__ __forceinline__ void w14_tile(W14Carry &c, const W14Taps &t, unsigned xa, unsigned ra, unsigned long long in_mask,
                                         unsigned long long out_mask, int iters)
{
    int xin, al, ah;
    unsigned long long saved;
    asm volatile(
        "v_mov_b32 v56, %[p0]\n\tv_mov_b32 v52, %[p1]\n\tv_mov_b32 v57, %[p2]\n\tv_mov_b32 v53, %[p3]\n\t"
        "v_mov_b32 v58, %[p4]\n\tv_mov_b32 v54, %[p5]\n\tv_mov_b32 v59, %[p6]\n\tv_mov_b32 v55, %[p7]\n\t"
        "ds_read_i16 %[xin], %[xa]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        ".p2align 6\n"
        "1:\n\t"
        SA_W14_GROUP("16", "0") SA_W14_GROUP("32", "16") SA_W14_GROUP("48", "32") SA_W14_GROUP("64", "48")
        "v_add_u32 %[xa], 64, %[xa]\n\t"
        "v_add_u32 %[ra], 64, %[ra]\n\t"
        "s_add_i32 %[cnt], %[cnt], -1\n\t"
        "s_cmp_lg_u32 %[cnt], 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mov_b32 %[p0], v56\n\tv_mov_b32 %[p1], v52\n\tv_mov_b32 %[p2], v57\n\tv_mov_b32 %[p3], v53\n\t"
        "v_mov_b32 %[p4], v58\n\tv_mov_b32 %[p5], v54\n\tv_mov_b32 %[p6], v59\n\tv_mov_b32 %[p7], v55"
        : [p0] "+v"(c.p[0]), [p1] "+v"(c.p[1]), [p2] "+v"(c.p[2]), [p3] "+v"(c.p[3]), [p4] "+v"(c.p[4]), [p5] "+v"(c.p[5]),
          [p6] "+v"(c.p[6]), [p7] "+v"(c.p[7]), [w0] "+v"(c.w0), [w1] "+v"(c.w1), [x0] "+v"(c.x0), [x1] "+v"(c.x1),
          [xa] "+v"(xa), [ra] "+v"(ra), [xin] "=&v"(xin), [al] "=&v"(al), [ah] "=&v"(ah), [cnt] "+s"(iters), [sv] "=&s"(saved)
        : [c01l] "v"(t.c01l), [c01h] "v"(t.c01h), [c2l] "v"(t.c2l), [c2h] "v"(t.c2h), [cfbl] "v"(t.cfbl), [cfbh] "v"(t.cfbh),
          [k] "s"(8192), [inm] "s"(in_mask), [outm] "s"(out_mask)
        : "memory", "scc", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59");
}

__global__ __launch_bounds__(64 * kWgWaves) void filter_w14_kernel(const int16_t *__restrict__ in, int16_t *__restrict__ out,
                                                                    int batch, SaQ15Params prm, const int16_t *__restrict__ rom)
{
    __shared__ __attribute__((aligned(16))) int16_t tin_all[kV2Waves][kFramesPerWave][kInPitch];
    __shared__ __attribute__((aligned(16))) int16_t ring_all[kV2Waves][kFramesPerWave][kRingPitch];
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wid & (kV2Waves - 1);
    const bool helper = wid >= kV2Waves;
    const int lane = threadIdx.x & 63;
    int16_t (*tin)[kInPitch] = tin_all[wave];
    int16_t (*ring)[kRingPitch] = ring_all[wave];
    const int fr = lane >> 4;           // frame slot in this wave
    const int l16 = lane & 15;          // role inside the row: 0 = input, 1..6 = sections 0..5, 7..8 = delay, 9..15 = input
    const int f0 = (blockIdx.x * kV2Waves + wave) * kFramesPerWave;
    constexpr int nt = SA_NPTS / kTile;
    const bool idle = f0 >= batch;      // a pair without frames still takes part in the workgroup's barriers
    if (helper) {
        q15_helper_wave([&](int j, int n0, int ncols) {
            q15_flush_tile<kRingPitch>(out, ring, (j * kTile) & (kRing - 1), f0, batch, n0, lane, kRing - 1, ncols);
        }, in, rom, tin, f0, batch, lane, prm.win_mode, idle);
        return;
    }
    SA_Q15_STAMP_BEGIN(blockIdx.x * kV2Waves + wave);

    W14Taps taps = w14_taps(16384, 0, 0, 0, 0);              // identity: (16384 x + 8192) >> 14 = x exactly
    if (l16 >= 1 && l16 <= prm.nsec_wide) {
        const int16_t *c = &prm.sos_q14[(l16 - 1) * 6];      // scipy row order [b0, b1, b2, a0, a1, a2], a0 ignored (= 1.0)
        taps = w14_taps(c[0], c[1], c[2], c[4], c[5]);
    }
    const int kin = (16 - l16) & 15;                         // 0..7 for the input lanes: lanes 0, 15, .., 9 take samples T0 .. T0+7
    const bool is_in = kin < 8;
    const unsigned long long out_mask = kOutMask;                  // lane 8 of every row that carries a frame
    const unsigned long long in_mask = 0xFE01FE01FE01FE01ull;    // lanes 0 and 9..15
    const int16_t *xrow = &tin[fr][0] + (is_in ? kin : 0);
    const unsigned xrow_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const int16_t *)xrow;
    const unsigned ring_addr = (unsigned)(size_t)(__attribute__((address_space(3))) int16_t *)(&ring[fr][0]);

    // Sample m lives in ring slot (m + 8) mod kRing (lane 8 holds samples T0 - 8 .. T0 - 1 at the end of the group that
    // starts at step T0).  The drain pass runs one iteration = four groups: the first delivers the frame's last eight
    // samples, the other three filter whatever the input ring holds into slots that were flushed long ago.
    W14Carry c = {};
    wg_lds_sync();
    for (int k = 0; k <= nt; ++k) {
        const int i0 = (k & 1) * kTile;
        if (!idle) w14_tile(c, taps, xrow_addr + 2 * i0, ring_addr + 2 * i0, in_mask, out_mask, k < nt ? kTile / 32 : 1);
        wg_lds_sync();
    }
    SA_Q15_STAMP_END();
}


// ------------------------------------------------------------------------------------------ FFT

__device__ __forceinline__ int sat16(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }

// The four outputs of a butterfly leave the adder tree as 32-bit sums X (re), Y (im) that still want the >> 2 of
// the spec.  Shift and pack are one instruction per half: v_ashrrev_i32 in its SDWA form writes the low word of
// its result into the chosen half of the destination (the first write zeroes the other half, the second preserves
// it); (sum of four int16) >> 2 lies in [-32768, 32767], so taking the low word is exact.  gfx950 wants one
// instruction between a sub-dword write and a read of the same register (the preserving write reads it): the four
// first-half writes come first, then the four second-half writes, then one s_nop before the compiler's code.
//   LO* / HI*: which sum goes to the low / high word of output 0..3.
#define SA_FX_PACK4(P0, P1, P2, P3, LO0, LO1, LO2, LO3, HI0, HI1, HI2, HI3)                                            \
    asm("v_ashrrev_i32_sdwa %0, %12, %4 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"        \
        "v_ashrrev_i32_sdwa %1, %12, %5 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"        \
        "v_ashrrev_i32_sdwa %2, %12, %6 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"        \
        "v_ashrrev_i32_sdwa %3, %12, %7 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"        \
        "v_ashrrev_i32_sdwa %0, %12, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"   \
        "v_ashrrev_i32_sdwa %1, %12, %9 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"   \
        "v_ashrrev_i32_sdwa %2, %12, %10 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"  \
        "v_ashrrev_i32_sdwa %3, %12, %11 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"  \
        "s_nop 0"                                                                                                      \
        : "=&v"(P0), "=&v"(P1), "=&v"(P2), "=&v"(P3)                                                                   \
        : "v"(LO0), "v"(LO1), "v"(LO2), "v"(LO3), "v"(HI0), "v"(HI1), "v"(HI2), "v"(HI3), "s"(2))

// lo(a) lo(b) + hi(a) hi(b), exact in 32 bits.  Written out: the builtin is selected as the accumulating two-operand
// form v_dot2c_i32_i16, which costs a v_mov of zero into the accumulator per product.
__device__ __forceinline__ int fx_dot2(unsigned a, unsigned b)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// the same with a wave-uniform second operand taken straight from its scalar register (no v_mov per product)
__device__ __forceinline__ int fx_dot2_s(unsigned a, unsigned b)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "s"(b));
    return r;
}

// y = sat16((u * w) >> 15), truncation (SA-FXFFT-1), for u packed as p = (lo = u.im, hi = u.re):
//   y.im = u.im wr + u.re wi = p . (wr, wi);   y.re = u.re wr - u.im wi = p . (-wi, wr)
// wi = -32768 has no int16 negation; the table holds it for the exponents 4082..4110 (-32768 sin rounds to -32768
// that far around pi/2).  Which butterflies of a thread can meet them is known at compile time (the u loops are
// unrolled): output 1 in stages 0 and 1 at u = 3 (e1 = 4082..4095), output 3 there at u = 1 (3 e1 = 4083..4110),
// output 2 at u = 1 or 2 in stages 0 and 1 and at u = 2 from stage 2 on (2 e1 = 4082..4110).  Those form the real
// part from the halves with two 24-bit multiplies (`wide1..3`); everything else takes both words of the table.
template <bool UNIFORM>
__device__ __forceinline__ unsigned fx_twiddle13(unsigned p, uint2 w)
{
    if constexpr (UNIFORM) return sat_pack2(fx_dot2_s(p, w.y) >> 15, fx_dot2_s(p, w.x) >> 15);
    else return sat_pack2(fx_dot2(p, w.y) >> 15, fx_dot2(p, w.x) >> 15);
}
template <bool UNIFORM>
__device__ __forceinline__ unsigned fx_twiddle2(unsigned p, unsigned w)
{
    const int pr = (hi16(p) * lo16(w) - lo16(p) * hi16(w)) >> 15;
    return sat_pack2(pr, (UNIFORM ? fx_dot2_s(p, w) : fx_dot2(p, w)) >> 15);
}

// the three twiddles of one butterfly of a per-lane stage: one 32-byte record (SaQ15Tables::twrec), read as 16 + 8 bytes
struct SaTw3 {
    uint2 w1, w2, w3;
};
__device__ __forceinline__ SaTw3 fx_twrec(const uint4 *__restrict__ twrec, int r)
{
    const uint4 a = twrec[2 * r];
    const uint2 b = *reinterpret_cast<const uint2 *>(&twrec[2 * r + 1]);
    return {make_uint2(a.x, a.y), make_uint2(a.z, a.w), b};
}

// one radix-4 DIF butterfly of SA-FXFFT-1 on packed (re, im) int16 pairs: 32-bit sums, >> 2 (truncation),
// Q15 twiddles on outputs 1..3 (exact pass-through when the exponent is 0), saturation to int16
// UNIFORM: the twiddles are the same for the whole wave (scalar loads, or compile-time exponents)
template <bool UNIFORM = false>
__device__ __forceinline__ void fx_butterfly(unsigned a, unsigned b, unsigned c, unsigned d, uint2 w1, uint2 w2,
                                             uint2 w3, bool unity, unsigned (&o)[4], bool wide1, bool wide2, bool wide3)
{
    const int ar = lo16(a), ai = hi16(a), br = lo16(b), bi = hi16(b);
    const int cr = lo16(c), ci = hi16(c), dr = lo16(d), di = hi16(d);
    const int sr = ar + cr, si = ai + ci, tr = ar - cr, ti = ai - ci;      // a +/- c
    const int ur = br + dr, ui = bi + di, vr = br - dr, vi = bi - di;      // b +/- d
    const int x0 = sr + ur, y0 = si + ui;
    const int x1 = tr + vi, y1 = ti - vr;                                  // a - i b - c + i d
    const int x2 = sr - ur, y2 = si - ui;
    const int x3 = tr - vi, y3 = ti + vr;                                  // a + i b - c - i d
    unsigned p0, p1, p2, p3;
    if (unity) {
        // pass-through: the results are in range by construction, the pack is all that is left
        SA_FX_PACK4(p0, p1, p2, p3, x0, x1, x2, x3, y0, y1, y2, y3);
        o[0] = p0; o[1] = p1; o[2] = p2; o[3] = p3;
    } else {
        SA_FX_PACK4(p0, p1, p2, p3, x0, y1, y2, y3, y0, x1, x2, x3);       // outputs 1..3 as (im, re) for the products
        o[0] = p0;
        o[1] = wide1 ? fx_twiddle2<UNIFORM>(p1, w1.x) : fx_twiddle13<UNIFORM>(p1, w1);
        o[2] = wide2 ? fx_twiddle2<UNIFORM>(p2, w2.x) : fx_twiddle13<UNIFORM>(p2, w2);
        o[3] = wide3 ? fx_twiddle2<UNIFORM>(p3, w3.x) : fx_twiddle13<UNIFORM>(p3, w3);
    }
}

// The first stage's butterfly: the inputs are real (imag = 0, new/command_control.vhd:123), which leaves 7 of the 16
// additions and 6 of the 8 shift-and-insert instructions: with s = a + c, t = a - c, u = b + d, v = b - d
//   out0 = (s + u, 0)    out1 = (t, -v)    out2 = (s - u, 0)    out3 = (t, v)        (each >> 2)
// and output 2's twiddle product is two multiplies (its imaginary input is 0).  Same results as fx_butterfly on
// (a, 0) .. (d, 0) by construction; a, b, c, d are sign-extended 16-bit samples.
__device__ __forceinline__ void fx_butterfly_real(int a, int b, int c, int d, uint2 w1, unsigned w2, uint2 w3, bool unity,
                                                  unsigned (&o)[4], bool wide1, bool wide3)
{
    const int sr = a + c, tr = a - c, ur = b + d, vr = b - d, nv = d - b;
    const int x0 = sr + ur, x2 = sr - ur;
    unsigned p0, p1, p2, p3;
    if (unity) {
        // (re, im) pairs as stored: out0 = (x0, 0), out1 = (t, -v), out2 = (x2, 0), out3 = (t, v)
        asm("v_ashrrev_i32_sdwa %0, %9, %4 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
            "v_ashrrev_i32_sdwa %1, %9, %5 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
            "v_ashrrev_i32_sdwa %2, %9, %6 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
            "v_ashrrev_i32_sdwa %3, %9, %5 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
            "v_ashrrev_i32_sdwa %1, %9, %7 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
            "v_ashrrev_i32_sdwa %3, %9, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
            "s_nop 0"
            : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3)
            : "v"(x0), "v"(tr), "v"(x2), "v"(nv), "v"(vr), "s"(2));
        o[0] = p0; o[1] = p1; o[2] = p2; o[3] = p3;
    } else {
        // outputs 1 and 3 as (im, re) for the products, output 2's real input as a plain number
        asm("v_ashrrev_i32_sdwa %0, %7, %3 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
            "v_ashrrev_i32_sdwa %1, %7, %4 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
            "v_ashrrev_i32_sdwa %2, %7, %5 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
            "v_ashrrev_i32_sdwa %1, %7, %6 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
            "v_ashrrev_i32_sdwa %2, %7, %6 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
            "s_nop 0"
            : "=&v"(p0), "=&v"(p1), "=&v"(p3)
            : "v"(x0), "v"(nv), "v"(vr), "v"(tr), "s"(2));
        const int u2 = x2 >> 2;
        o[0] = p0;
        o[1] = wide1 ? fx_twiddle2<false>(p1, w1.x) : fx_twiddle13<false>(p1, w1);
        o[2] = sat_pack2((u2 * lo16(w2)) >> 15, (u2 * hi16(w2)) >> 15);
        o[3] = wide3 ? fx_twiddle2<false>(p3, w3.x) : fx_twiddle13<false>(p3, w3);
    }
}

// SA-FXFFT-1 with 1024 threads per frame: 16 positions per thread (t + 1024 m); the seven radix-4 stages run as four
// register passes -- stage 0 from global memory, then (1,2), (3,4), (5,6) -- with one LDS exchange between passes.
// (Round 1 and most of round 2 ran 256 threads x 64 positions, stages 4..6 in registers: 120 registers per thread, 2 waves
// per SIMD, a quarter of a wave's life in s_waitcnt behind a barrier with one other wave to cover: 192 us.  Rounds 2-3 ran
// 1024 threads with ONE stage per LDS exchange for stages 0..4: 155-158 us.  Pairing the stages (three exchanges instead
// of five, the second stage of a pair shares one twiddle triple among a thread's four butterflies) and reading a lane's
// three twiddles as one 32-byte record instead of three strided gathers: 126-137 us by box, profiles/r4_fft_q15_passes.txt.)
//   7 radix-4 DIF stages, Stockham addressing:
//   storage after s stages: pos = j * 4^s + kappa   (j: remaining time index, kappa: bins so far)
//   butterfly bf in [0,4096): j' = bf >> 2s, kappa = bf & (4^s - 1); inputs at bf + i*4096;
//   output i' at (j' << (2s+2)) | (i' << 2s) | kappa; twiddle exponent i' * j' * 4^s.
// 8 waves per SIMD (two frames per CU, 64 KiB of LDS each) need <= 64 registers: the second launch bound asks for that.
constexpr int kFftWide = 1024;

template <bool WINDOW>
__global__ __launch_bounds__(kFftWide, 8) void fft_q15_kernel(const int16_t *__restrict__ in,
                                                               int16_t *__restrict__ out_iq, int batch,
                                                               SaQ15Params prm, const int16_t *__restrict__ rom,
                                                               const uint2 *__restrict__ tw, const uint4 *__restrict__ twrec)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_q[];
    unsigned *buf = reinterpret_cast<unsigned *>(smem_q);     // [16384] packed (re, im)
    const int t = threadIdx.x;
    const int f = blockIdx.x;
    if (f >= batch) return;
    // ---- stage 0 straight from global memory: the thread's 16 positions t + 1024 m as 2-byte loads (128 contiguous
    // bytes per wave instruction), optional window, imag = 0 (new/command_control.vhd:123).  No staging pass
    // through LDS, no barrier in front of the first butterflies; outputs 4 bf + i' are one 16-byte LDS write.
    // Exponents with wi = -32768 (see fx_butterfly): stages 0 and 1, u = 3 for output 1, u = 1 for output 3.
    {
        const int16_t *xf = in + (size_t)f * SA_NPTS + t;
        int x[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) x[m] = xf[kFftWide * m];
        if constexpr (WINDOW) {
            int c[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) c[m] = rom[t + kFftWide * m];
#pragma unroll
            for (int m = 0; m < 16; ++m)
                x[m] = (prm.win_mode == SA_WIN_RTL_SIGNED) ? win_rtl(x[m], c[m]) : win_u16(x[m], c[m]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int bf = t + kFftWide * u;                   // j' = bf, kappa = 0, e1 = bf
            unsigned o[4];
            const SaTw3 w = fx_twrec(twrec, bf);
            fx_butterfly_real(x[u], x[u + 4], x[u + 8], x[u + 12], w.w1, w.w2.x, w.w3, bf == 0, o, u == 3, u == 1);
            *reinterpret_cast<uint4 *>(buf + 4 * bf) = make_uint4(o[0], o[1], o[2], o[3]);
        }
        __syncthreads();
    }

    // ---- stages 1..4 as two register passes of two stages each.  A thread that runs the stage-s butterflies
    // bf = t + 1024 u (u = 0..3) holds, in output i' of butterfly u, input u of the stage-(s+1) butterfly
    // ((j' mod 4^(5-s)) << (2s+2)) | (i' << 2s) | kappa -- its own four next butterflies, which all share ONE twiddle
    // exponent (j'' = (t >> 2s) mod 4^(5-s) does not depend on i').  One LDS exchange per two stages instead of one per
    // stage, a quarter of the twiddle loads in the second stage of a pass.
    //   outputs of the pass: pos = (j'' << (2s+4)) | (i'' << (2s+2)) | (i' << 2s) | kappa
    // Pass (1,2) writes with kappa = t & 3 in the bank bits: the words are stored at pos ^ ((j'' & 15) << 2), which spreads
    // the 16 values of j'' in a wave over the banks (conflict-free), and pass (3,4) reads t + 1024 m through the same
    // exchange of bits (there it permutes the lanes of a wave: conflict-free as well).
    unsigned v[16];
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    {
#pragma unroll
        for (int m = 0; m < 16; ++m) v[m] = buf[t + kFftWide * m];
        __syncthreads();
        unsigned x[16];                                        // x[4 i' + u]
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e1 = ((t + kFftWide * u) >> 2) << 2;
            unsigned o[4];
            // exponents with wi = -32768 (see fx_butterfly): output 1 at u = 3, output 3 at u = 1, output 2 at u = 1 or 2
            const SaTw3 w = fx_twrec(twrec, 4096 + (e1 >> 2));
            fx_butterfly(v[u], v[u + 4], v[u + 8], v[u + 12], w.w1, w.w2, w.w3, e1 == 0, o, u == 3, u == 1 || u == 2, u == 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) x[4 * i + u] = o[i];
        }
        // stage 2: j'' = (t >> 2) & 255, exponent 16 j'' (never in 4082..4095; 3 e never in 4083..4110; 2 e = 4096 for
        // j'' = 128, i.e. threads 512..515: wave 8 takes the two-multiply form for output 2)
        const int j2 = (t >> 2) & 255, e2 = j2 << 4;
        const SaTw3 w2 = fx_twrec(twrec, 5120 + j2);
        const uint2 a1 = w2.w1, a2 = w2.w2, a3 = w2.w3;
        const int ob = ((j2 << 6) | (t & 3)) ^ ((j2 & 15) << 2);
#pragma unroll
        for (int ip = 0; ip < 4; ++ip) {
            unsigned o[4];
            if (wave == 8) fx_butterfly(x[4 * ip], x[4 * ip + 1], x[4 * ip + 2], x[4 * ip + 3], a1, a2, a3, false, o, false, true, false);
            else fx_butterfly(x[4 * ip], x[4 * ip + 1], x[4 * ip + 2], x[4 * ip + 3], a1, a2, a3, e2 == 0, o, false, false, false);
#pragma unroll
            for (int i = 0; i < 4; ++i) buf[ob ^ ((4 * i + ip) << 2)] = o[i];
        }
        __syncthreads();
    }
    {
        // pass (3,4): scalar twiddles in both stages (j' = wave + 16 u, then j'' = wave)
#pragma unroll
        for (int m = 0; m < 16; ++m) v[m] = buf[(t + kFftWide * m) ^ (wave << 2)];
        __syncthreads();
        unsigned x[16];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e1 = (wave + 16 * u) << 6;
            unsigned o[4];
            fx_butterfly<true>(v[u], v[u + 4], v[u + 8], v[u + 12], tw[e1], tw[2 * e1], tw[3 * e1], e1 == 0, o, false, u == 2, false);
#pragma unroll
            for (int i = 0; i < 4; ++i) x[4 * i + u] = o[i];
        }
        const int e2 = wave << 8;                              // 2 e = 4096 for wave 8
        const uint2 a1 = tw[e2], a2 = tw[2 * e2], a3 = tw[3 * e2];
        const int ob = (wave << 10) | (t & 63);
#pragma unroll
        for (int ip = 0; ip < 4; ++ip) {
            unsigned o[4];
            if (wave == 8) fx_butterfly<true>(x[4 * ip], x[4 * ip + 1], x[4 * ip + 2], x[4 * ip + 3], a1, a2, a3, false, o, false, true, false);
            else fx_butterfly<true>(x[4 * ip], x[4 * ip + 1], x[4 * ip + 2], x[4 * ip + 3], a1, a2, a3, e2 == 0, o, false, false, false);
#pragma unroll
            for (int i = 0; i < 4; ++i) buf[ob | ((4 * i + ip) << 6)] = o[i];
        }
        __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) v[m] = buf[t + kFftWide * m];
    unsigned w[16];
    // stage 5 (4^s = 1024): j' = u, kappa = t; outputs land at m' = 4u + i'; exponents are compile-time
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        unsigned o[4];
        fx_butterfly<true>(v[u], v[u + 4], v[u + 8], v[u + 12], tw[u * 1024], tw[2 * u * 1024], tw[3 * u * 1024], u == 0, o, false,
                           u == 2, false);
#pragma unroll
        for (int i = 0; i < 4; ++i) w[4 * u + i] = o[i];
    }
    // stage 6 (4^s = 4096): no twiddles; outputs at m' = u + 4 i' = natural-order bin t + 1024 m'
    // frame layout: [16384] x (re, im) int16 = 65536 bytes (imp/sequ2.vhd:153); one dword per lane
    unsigned *o32 = reinterpret_cast<unsigned *>(out_iq + (size_t)f * SA_NPTS * 2);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        unsigned o[4];
        fx_butterfly(w[u], w[u + 4], w[u + 8], w[u + 12], make_uint2(0u, 0u), make_uint2(0u, 0u), make_uint2(0u, 0u), true, o, false,
                     false, false);
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_nontemporal_store(o[i], o32 + t + kFftWide * (u + 4 * i));   // streaming: written once
    }
}

}  // namespace

hipError_t sa_launch_filter_q15(const int16_t *in, int16_t *out_time, int batch, const SaQ15Params &p,
                                const SaQ15Tables &t, hipStream_t stream, SaLaunchEv ev)
{
    if (batch <= 0) return hipSuccess;
    const int per_wg = kFramesPerWave * kV2Waves;
    const dim3 grid_wg((batch + per_wg - 1) / per_wg), block_wg(64 * kWgWaves);
    if (p.filter == SA_FILTER_NONE) {
        const size_t chunks = (size_t)batch * (SA_NPTS / 8);
        hipExtLaunchKernelGGL(window_q15_kernel, dim3((unsigned)((chunks + kWinThreads - 1) / kWinThreads)), dim3(kWinThreads), 0, stream,
                              ev.start, ev.stop, 0, in, out_time, batch, p, t.rom);
    } else if (p.filter == SA_FILTER_WIDE) {
        hipExtLaunchKernelGGL(filter_w14_kernel, grid_wg, block_wg, 0, stream, ev.start, ev.stop, 0, in, out_time, batch, p, t.rom);
    } else {
        // B1 = 0 in both coefficient sets (wire order b0,b1,b2,a0,a1,a2 per set): the seven-instruction step
        const bool nob1 = p.c12[1] == 0 && p.c12[7] == 0;
        if (nob1)
            hipExtLaunchKernelGGL(filter_q7_kernel<true>, grid_wg, block_wg, 0, stream, ev.start, ev.stop, 0, in, out_time, batch, p, t.rom);
        else
            hipExtLaunchKernelGGL(filter_q7_kernel<false>, grid_wg, block_wg, 0, stream, ev.start, ev.stop, 0, in, out_time, batch, p, t.rom);
    }
    return hipGetLastError();
}

hipError_t sa_launch_fft_q15(const int16_t *in_time, int16_t *out_iq, int batch, bool apply_window,
                             const SaQ15Params &p, const SaQ15Tables &t, hipStream_t stream, SaLaunchEv ev)
{
    if (batch <= 0) return hipSuccess;
    const dim3 grid(batch), block(kFftWide);
    const int lds = SA_NPTS * 4;
    auto k = apply_window ? fft_q15_kernel<true> : fft_q15_kernel<false>;
    const hipError_t e = sa_set_dyn_lds_once(reinterpret_cast<const void *>(k), lds);
    if (e != hipSuccess) return e;
    hipExtLaunchKernelGGL(k, grid, block, lds, stream, ev.start, ev.stop, 0, in_time, out_iq, batch, p, t.rom, t.tw, t.twrec);
    return hipGetLastError();
}
