// specan_abi.cpp -- host side of the C ABI declared in include/specan.h.
//
// Owns: the opaque handle, device tables (window, twiddles, IIR plan), the Q15 workspace and the
// command-byte state machine that mirrors new/rx_filter_coeff.vhd + new/command_control.vhd.
// Never touches caller tensors except through the pointers given to the process calls, never
// falls back to CPU compute.
#include "../../include/specan.h"
#include "sa_common.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <set>
#include <string>
#include <utility>
#include <vector>

namespace {

// last sa_create() failure of the calling thread (sa_last_error(NULL)); per thread, so that concurrent creates on
// several host threads -- one per GPU, SURVEY 8(e) -- do not race on it
thread_local std::string g_create_error;

// imp/filter_pkg.vhd:54-68, wire order B0,B1,B2,A0,A1,A2 per set (ALPHA then BETA)
const int8_t kDefaultQ7[12] = {-14, 0, 14, 107, 21, 127, -15, 0, 15, 107, -21, 127};

struct Mat2 {
    double a, b, c, d;
};
inline Mat2 mul(const Mat2 &x, const Mat2 &y)
{
    return {x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d};
}

inline Mat2 mpow(Mat2 m, int e)
{
    Mat2 r = {1, 0, 0, 1};
    while (e > 0) {
        if (e & 1) r = mul(r, m);
        m = mul(m, m);
        e >>= 1;
    }
    return r;
}

inline void put(float *dst, const Mat2 &m)
{
    dst[0] = (float)m.a; dst[1] = (float)m.b; dst[2] = (float)m.c; dst[3] = (float)m.d;
}

// Pole coordinates of one section (sa_common.hpp): M = T^-1 has the eigen-directions (1, a1 + lambda) as unit columns
// -- real and imaginary part for a complex pair -- and A = T A0 M is what predictor and scan work with.
void pole_coordinates(double a1, double a2, Mat2 *Mo, Mat2 *To, Mat2 *Ao)
{
    const Mat2 A0 = {-a1, 1.0, -a2, 0.0};
    Mat2 M = {1, 0, 0, 1};
    const double disc = a1 * a1 - 4.0 * a2;
    if (a2 != 0.0 && disc != 0.0) {
        double c0[2], c1[2];
        if (disc < 0.0) {
            c0[0] = 1.0; c0[1] = 0.5 * a1;                 // Re (1, a1 + lambda), lambda = -a1/2 + i sqrt(-disc)/2
            c1[0] = 0.0; c1[1] = 0.5 * std::sqrt(-disc);   // Im
        } else {
            const double sq = std::sqrt(disc);
            c0[0] = 1.0; c0[1] = a1 + 0.5 * (-a1 + sq);
            c1[0] = 1.0; c1[1] = a1 + 0.5 * (-a1 - sq);
        }
        const double n0 = std::hypot(c0[0], c0[1]), n1 = std::hypot(c1[0], c1[1]);
        const Mat2 cand = {c0[0] / n0, c1[0] / n1, c0[1] / n0, c1[1] / n1};
        const double det = cand.a * cand.d - cand.b * cand.c;
        // unit columns: |det| = sine of the angle between them; below 1e-6 the pair is numerically defective
        if (std::isfinite(det) && std::fabs(det) > 1e-6) M = cand;
    }
    const double detM = M.a * M.d - M.b * M.c;
    const Mat2 T = {M.d / detM, -M.b / detM, -M.c / detM, M.a / detM};
    *Mo = M;
    *To = T;
    *Ao = mul(T, mul(A0, M));
}

// Padded section count, unit-numerator rewrite and folded gain shared by both plan layouts; returns the padded count.
int normalise_cascade(const double *sos_in, int nsec_in, double *sos /*[36]*/, bool *unit_out, double *gain_out)
{
    const int nsec = nsec_in == 0 ? 0 : (nsec_in <= 2 ? 2 : (nsec_in <= 4 ? 4 : 6));
    for (int s = 0; s < nsec; ++s)
        for (int i = 0; i < 6; ++i)
            sos[6 * s + i] = s < nsec_in ? sos_in[6 * s + i] : ((i == 0 || i == 3) ? 1.0 : 0.0);
    bool unit = nsec > 0 && nsec == nsec_in;
    double gain = 1.0;
    for (int s = 0; s < nsec && unit; ++s) {
        const double b0 = sos[6 * s], b2 = sos[6 * s + 2];
        if (b0 == 0.0 || b2 != b0 || !std::isfinite(1.0 / b0)) unit = false;
        gain *= b0;
    }
    if (unit && (!std::isfinite(gain) || std::fabs(gain) < 1e-30 || std::fabs(gain) > 1e30)) unit = false;
    if (unit) {
        for (int s = 0; s < nsec; ++s) {
            const double b0 = sos[6 * s];
            sos[6 * s + 1] /= b0;
            sos[6 * s] = 1.0;
            sos[6 * s + 2] = 1.0;
        }
    } else {
        gain = 1.0;
    }
    *unit_out = unit;
    *gain_out = gain;
    return nsec;
}

inline void put_cm(float *dst, const Mat2 &m)      // column-major
{
    dst[0] = (float)m.a; dst[1] = (float)m.c; dst[2] = (float)m.b; dst[3] = (float)m.d;
}


// Build the predict/scan/recurse plan for an a0-normalised SOS (rows b0,b1,b2,1,a1,a2), double in.
// The kernels are compiled for 2, 4 and 6 sections; shorter cascades are padded with identity
// sections (b0 = 1, rest 0: y = x exactly, all scan matrices and predictor taps come out zero).
//
// Unit-numerator form: when no padding is needed and every section has b2 == b0 != 0 (all-pole-pair
// zeros on the unit circle: Butterworth / Chebyshev / elliptic low-, high-pass and band-stop), the
// sections are rewritten as b = [1, b1/b0, 1] and the product of the b0's is folded into this plan's
// copy of the window: one multiply less per sample and section in the recursion.
// half_win: 0.5 * window in natural order (size SA_NPTS).
// cosw: {a0, a1} when the window is a0 - a1 cos(2 pi n / (N-1)) (then the IIR kernels evaluate it in place,
// see SaIirLaneTab::wgen), null for any other window.
void build_plan(const double *sos_in, int nsec_in, SaIirK *plan, SaIirLaneTab *lt, const float *half_win,
                const double *cosw = nullptr)
{
    std::memset(plan, 0, sizeof(*plan));
    std::memset(lt, 0, sizeof(*lt));
    double sos[36];
    bool unit;
    double gain;
    const int nsec = normalise_cascade(sos_in, nsec_in, sos, &unit, &gain);
    plan->nsec = nsec;
    plan->unit = unit ? 1 : 0;
    plan->gain = (float)gain;
    if (half_win)
        for (int t = 0; t < 256; ++t)
            for (int g = 0; g < 16; ++g)
                for (int e = 0; e < 4; ++e)
                    lt->win_t[(g * 256 + t) * 4 + e] = (float)((double)half_win[64 * t + 4 * g + e] * gain);
    plan->wingen = 0;
    if (half_win && cosw) {
        const double theta = 2.0 * M_PI / (double)(SA_NPTS - 1), S = 0.5 * gain;
        plan->wingen = 1;
        lt->wg0 = (float)(S * cosw[0]);
        for (int t = 0; t < SA_NTHREADS; ++t)
            for (int h = 0; h < 2; ++h) {
                const double a = theta * (double)(64 * t + 32 * h);
                lt->wgen[t][2 * h] = (float)(-S * cosw[1] * std::cos(a));
                lt->wgen[t][2 * h + 1] = (float)(S * cosw[1] * std::sin(a));
            }
        for (int j = 0; j < SA_CHUNK; ++j) {
            lt->wcs[j][0] = (float)std::cos(theta * j);
            lt->wcs[j][1] = (float)std::sin(theta * j);
        }
    }
    for (int s = 0; s < nsec; ++s) {
        const double *r = sos + 6 * s;
        const double b0 = r[0], b1 = r[1], b2 = r[2], a1 = r[4], a2 = r[5];
        SaIirSecK &sp = plan->sec[s];
        sp.c[0] = (float)b0; sp.c[1] = (float)b1; sp.c[2] = (float)b2; sp.c[3] = (float)a1; sp.c[4] = (float)a2;
        Mat2 M, T, A;
        pole_coordinates(a1, a2, &M, &T, &A);
        put_cm(sp.mback, M);
        double v0 = T.a * (b1 - a1 * b0) + T.b * (b2 - a2 * b0);      // T Bv
        double v1 = T.c * (b1 - a1 * b0) + T.d * (b2 - a2 * b0);
        float (*mdst)[2] = s == 0 ? plan->m0 : plan->sec[s - 1].mnext;       // taps of section s ride with section s-1
        for (int j = SA_PRED_TAPS - 1; j >= 0; --j) {     // m[j] = A^(15-j) Bv: the taps of a HALF chunk (block Horner)
            mdst[j][0] = (float)v0;
            mdst[j][1] = (float)v1;
            const double n0 = A.a * v0 + A.b * v1, n1 = A.c * v0 + A.d * v1;
            v0 = n0; v1 = n1;
        }
        put_cm(s == 0 ? plan->p16_0 : plan->sec[s - 1].p16next, mpow(A, SA_PRED_TAPS));
        const Mat2 Pc = mpow(A, SA_CHUNK);                  // one chunk
        const Mat2 P2 = mul(Pc, Pc);                        // one thread (two chunks)
        const Mat2 Prow = mpow(P2, 16);                     // one 16-lane row
        put_cm(sp.pc, Pc);
        Mat2 q = P2, qr = Prow;
        auto tiny = [](const Mat2 &m) {
            const double mx = std::fmax(std::fmax(std::fabs(m.a), std::fabs(m.b)), std::fmax(std::fabs(m.c), std::fabs(m.d)));
            return mx < 1e-10;
        };
        sp.flags = tiny(Prow) ? SA_IIR_SKIP_ROWSCAN : 0;
        for (int i = 0; i < 4; ++i) {                       // powers 1,2,4,8
            put_cm(sp.plev[i], q);
            put_cm(sp.prow[i], qr);
            if (tiny(q)) sp.flags |= 1 << i;
            q = mul(q, q);
            qr = mul(qr, qr);
        }
        Mat2 pw = {1, 0, 0, 1};
        for (int i = 0; i < 16; ++i) {                      // lanetab[s][i] = P2^i
            put_cm(lt->p[s][i], pw);
            pw = mul(pw, P2);
        }
    }
}

// The RTL taps as real numbers: y = (B2 x + B1 x1 + B0 x2 - A0 y2 - A1 y1)/128
// => scipy row [B2,B1,B0, 128, A1, A0] / 128; stages alternate set 0 / set 1 (filter_iir12_cust.vhd:68-240).
void sos_from_q7(const int8_t *c12, double *sos /*[6][6]*/)
{
    for (int k = 0; k < 6; ++k) {
        const int8_t *c = c12 + ((k & 1) ? 6 : 0);
        double *r = sos + 6 * k;
        r[0] = c[2] / 128.0; r[1] = c[1] / 128.0; r[2] = c[0] / 128.0;
        r[3] = 1.0; r[4] = c[4] / 128.0; r[5] = c[3] / 128.0;
    }
}

}  // namespace

hipError_t sa_set_dyn_lds_once(const void *kernel, int bytes)
{
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;        // (kernel, device)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({kernel, dev})) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.insert({kernel, dev});
    return e;
}

struct sa_handle {
    int device = 0;
    std::string err;
    uint8_t filter_mode = SA_FILTER_NONE;
    int8_t c12_custom[12] = {0};
    int win_mode_q15 = SA_WIN_RTL_SIGNED;
    int16_t sos_q14[SA_MAXSEC * 6] = {0};
    int nsec_q14 = 0;
    // UART byte-stream state (new/rx_filter_coeff.vhd:41-66)
    int rx_count = -1;            // -1 = IDLE, 0..11 = ACQUIRE
    int8_t rx_buf[12] = {0};
    // host tables
    std::vector<int16_t> rom;
    SaIirK plan_default{}, plan_custom{};
    SaIirLaneTab lt_default{}, lt_custom{};
    std::vector<float> half_win;           // 0.5 * float window, natural order
    bool win_is_cos = true;                // the float window is a0 - a1 cos(2 pi n / (N-1)) (default: Hann)
    double win_cos[2] = {0.5, 0.5};
    double sos_custom[36] = {0};          // a0-normalised custom cascade (kept to rebuild on window change)
    int nsec_custom = 0;
    // device tables
    float4 *d_win_b = nullptr;
    float4 *d_win_t = nullptr;
    float4 *d_twT = nullptr, *d_twB = nullptr;
    float2 *d_twC = nullptr;
    SaIirLaneTab *d_lt_default = nullptr, *d_lt_custom = nullptr;
    int16_t *d_rom = nullptr;
    uint2 *d_twq = nullptr;          // SA-FXFFT-1 twiddles, {(wr, wi), (-wi, wr)} packed int16 pairs
    uint4 *d_twrec = nullptr;        // the same words regrouped per butterfly for the per-lane stages (SaQ15Tables::twrec)
    // Q15 IIR workspace, one per launch slot (slot 0 = ordered mode; overlap mode uses slots 0..depth-1).  A
    // workspace that is outgrown is retired, not freed (hipFree synchronises the whole device; launches in flight
    // may still use it): freed in sa_destroy.  Growth is geometric so that the retired total stays below the live one.
    static constexpr int kMaxOverlap = 4;
    int16_t *d_work[kMaxOverlap] = {nullptr, nullptr, nullptr, nullptr};
    int work_frames[kMaxOverlap] = {0, 0, 0, 0};
    std::vector<void *> retired;
    // ---- stream-ordered control plane (no device-wide synchronisation anywhere after sa_create)
    // Table uploads run on the handle's own control stream: it first waits for everything the handle has
    // launched so far, copies from a pinned staging slot, and records `uploaded`; the next process call makes its
    // stream wait for that event.  Other handles and other streams of the device are never stalled.
    // Ordering behind the handle's own launches: the event `launched` is bound to the completion of the last kernel
    // of every ordered-mode process call (hipExtLaunchKernel's stop event: it rides on the dispatch packet, where a
    // hipEventRecord after the launch puts a marker packet between two launches and measured 1.3-2.7 % of the step,
    // gpurun_out/ab_ov.log).  Uploads, stream switches and sa_destroy wait for that event; the caller's stream is
    // never touched after the call that passed it has returned, so the caller may destroy it at any time (touching
    // a destroyed stream crashes inside the runtime: gpurun_out/gpu_tests_b.log).  `last_stream` is compared, never
    // dereferenced; the capture query of control_allowed() touches `capture_stream` only, a stream whose capture the
    // handle has not yet seen closed (the caller ends a capture before destroying its stream: include/specan.h).
    hipStream_t ctl = nullptr;
    hipEvent_t launched = nullptr, uploaded = nullptr;
    bool launched_valid = false;           // `launched` has been bound to a launch at least once
    unsigned upload_gen = 0;               // number of uploads issued so far
    unsigned seen_gen = 0;                 // ordered mode: uploads the data stream has waited for
    hipStream_t last_stream = nullptr;     // stream of the most recent ordered-mode process call (compared, never used)
    bool have_last_stream = false;
    // a process call was captured into a graph on `capture_stream` and that capture has not been seen closed yet
    // (control_allowed): sticky across calls on OTHER streams; cleared by the query on that stream reporting "none"
    bool capture_open = false;
    hipStream_t capture_stream = nullptr;
    // ---- launch timing (opt-in, sa_set_profiling): a ring of timing-enabled event pairs; ordered-mode call k binds
    // pair k mod n to the begin of its first and the end of its last kernel (hipExtLaunchKernel: the events ride on the
    // dispatch packets, no marker packets), and `launched` aliases the pair's stop event meanwhile
    std::vector<hipEvent_t> prof_start, prof_stop;
    hipEvent_t launched_own = nullptr;     // the handle's own (timing-disabled) completion event
    unsigned long long prof_calls = 0;
    // ---- overlapped launches (opt-in, sa_set_overlap): consecutive process calls alternate over `overlap` internal
    // streams so that the tail of one launch runs under the head of the next; see include/specan.h
    int overlap = 1;
    hipStream_t ov_stream[kMaxOverlap] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t ov_fit_stream = nullptr;      // the caller stream the internal streams were last fitted to (compared, never used)
    bool ov_fit_valid = false;
    hipEvent_t ov_fork[kMaxOverlap] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ov_done[kMaxOverlap] = {nullptr, nullptr, nullptr, nullptr};
    bool ov_used[kMaxOverlap] = {false, false, false, false};      // ov_done[i] has been recorded
    bool ov_unjoined[kMaxOverlap] = {false, false, false, false};  // ... and no caller stream waits for it yet
    unsigned ov_seen_gen[kMaxOverlap] = {0, 0, 0, 0};
    unsigned long long ov_calls = 0;
    static constexpr int kStage = 4;       // pinned staging slots (a slot is reused after kStage uploads)
    void *stage[kStage] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t stage_done[kStage] = {nullptr, nullptr, nullptr, nullptr};
    bool stage_used[kStage] = {false, false, false, false};
    int stage_next = 0;
    // transport / sequencing state of imp/sequ2.vhd as far as the command bytes define it
    uint8_t transport = SA_CMD_ETHERNET_MODE;     // ether_en <= '1' on reset (imp/sequ2.vhd:85-86)
};

namespace {

int fail(sa_handle *h, int code, const char *what, hipError_t e = hipSuccess)
{
    char buf[256];
    if (e != hipSuccess)
        std::snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else
        std::snprintf(buf, sizeof buf, "%s", what);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

#define SA_HIP(h, call)                                          \
    do {                                                         \
        hipError_t e_ = (call);                                  \
        if (e_ != hipSuccess) return fail((h), SA_EHIP, #call, e_); \
    } while (0)

// ---- overlap mode: which streams run beside each other ------------------------------------------------------
// The runtime maps a process's streams onto a few hardware queues (four here) and two streams that share a queue
// execute in order: a handle whose two internal streams fall on one queue gets no overlap and pays for the fork /
// join events on top (measured, tools/ubench/stream_pairs.hip and profiles/r3_overlap_streams.txt: streams 3 and 4
// created back to back share a queue; such a handle ran 144 us per batch against 135 us stream-ordered and 127 us
// with two queues).  The mapping is not exposed, so sa_set_overlap() asks the hardware: a one-wave kernel that
// waits 100 us on the constant 100 MHz counter is put on both streams; if the second finishes within 150 us of the
// first one's start they ran side by side.  The loop ends on the counter or on its iteration cap, whichever first.
__global__ void sa_spin_kernel(unsigned ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < 200000 && __builtin_amdgcn_s_memrealtime() - t0 < ticks; ++i) __builtin_amdgcn_s_sleep(8);
}

// 1 = kernels on a and b overlap, 0 = they run one after the other, negative = HIP error (text in *err)
int streams_run_side_by_side(hipStream_t a, hipStream_t b, hipError_t *err)
{
    constexpr unsigned kTicks = 10000;                   // 100 us
    hipEvent_t e0 = nullptr, ea = nullptr, eb = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&ea);
    if (e == hipSuccess) e = hipEventCreate(&eb);
    float ms = 1e9f;
    for (int pass = 0; pass < 2 && e == hipSuccess; ++pass) {        // pass 0 warms the launch path up (code object load)
        const unsigned ticks = pass == 0 ? 10u : kTicks;
        e = hipEventRecord(e0, a);
        if (e == hipSuccess) hipLaunchKernelGGL(sa_spin_kernel, dim3(1), dim3(64), 0, a, ticks);
        if (e == hipSuccess) hipLaunchKernelGGL(sa_spin_kernel, dim3(1), dim3(64), 0, b, ticks);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(ea, a);
        if (e == hipSuccess) e = hipEventRecord(eb, b);
        if (e == hipSuccess) e = hipEventSynchronize(ea);
        if (e == hipSuccess) e = hipEventSynchronize(eb);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, eb);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (ea) (void)hipEventDestroy(ea);
    if (eb) (void)hipEventDestroy(eb);
    if (e != hipSuccess) { *err = e; return -1; }
    return ms < 0.15f ? 1 : 0;
}

// A new stream that runs beside every stream in `avoid`.  Best effort: after six candidates the last one is kept
// whatever the probe said (a GPU busy with other work can make side-by-side kernels look serial, and with four
// hardware queues five streams cannot all be apart).
int pick_stream(sa_handle *h, const hipStream_t *avoid, int navoid, hipStream_t *out)
{
    hipStream_t rejected[6];
    int nrej = 0, rc = SA_OK;
    *out = nullptr;
    for (int tries = 0; tries < 6 && !*out && rc == SA_OK; ++tries) {
        hipStream_t c = nullptr;
        hipError_t e = hipStreamCreateWithFlags(&c, hipStreamNonBlocking);
        if (e != hipSuccess) { rc = fail(h, SA_EHIP, "overlap: hipStreamCreateWithFlags", e); break; }
        bool ok = true;
        for (int j = 0; j < navoid && ok; ++j) {
            hipError_t pe = hipSuccess;
            const int r = streams_run_side_by_side(avoid[j], c, &pe);
            if (r < 0) { rc = fail(h, SA_EHIP, "overlap: stream probe", pe); ok = false; }
            else ok = r == 1;
        }
        if (rc == SA_OK && (ok || tries == 5)) *out = c;
        else rejected[nrej++] = c;
    }
    for (int q = 0; q < nrej; ++q) (void)hipStreamDestroy(rejected[q]);
    return rc;
}

constexpr size_t kStageBytes = sizeof(SaIirLaneTab);      // the largest table a handle uploads
static_assert(kStageBytes >= sizeof(float) * SA_NPTS, "a staging slot holds any table of the handle");

// Control-plane calls change host state and device tables; a process call that is being captured into a hipGraph
// has frozen the host part (kernel arguments) but not the tables, so such calls are refused while a capture that took
// one of the handle's process calls is still open.  Checked at the top of every control-plane entry point, before
// anything is changed.  The record is sticky: a later, uncaptured call on ANOTHER stream does not clear it; only the
// query on the capturing stream does (here, or in begin_call when that stream is used again), and once it has reported
// "none" that stream is never queried again on the record's behalf.
int control_allowed(sa_handle *h)
{
    if (!h->capture_open) return SA_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(h->capture_stream, &cs) != hipSuccess) {
        (void)hipGetLastError();
        cs = hipStreamCaptureStatusNone;
    }
    if (cs != hipStreamCaptureStatusNone)
        return fail(h, SA_ESTATE, "control-plane call while a stream that captured one of the handle's calls is still capturing");
    h->capture_open = false;
    h->capture_stream = nullptr;
    return SA_OK;
}

// Stream-ordered table update (see sa_handle): after everything launched so far, before everything launched
// later; asynchronous for the host except when all staging slots are still waiting for their copies.
int upload(sa_handle *h, void *dst, const void *src, size_t bytes)
{
    if (bytes > kStageBytes) return fail(h, SA_EINVAL, "upload: table larger than the staging slot");
    SA_HIP(h, hipSetDevice(h->device));
    const int slot = h->stage_next;
    h->stage_next = (slot + 1) % sa_handle::kStage;
    if (h->stage_used[slot]) SA_HIP(h, hipEventSynchronize(h->stage_done[slot]));   // that slot's old copy has run
    std::memcpy(h->stage[slot], src, bytes);
    if (h->launched_valid) SA_HIP(h, hipStreamWaitEvent(h->ctl, h->launched, 0));
    for (int i = 0; i < sa_handle::kMaxOverlap; ++i)
        if (h->ov_used[i]) SA_HIP(h, hipStreamWaitEvent(h->ctl, h->ov_done[i], 0));
    SA_HIP(h, hipMemcpyAsync(dst, h->stage[slot], bytes, hipMemcpyHostToDevice, h->ctl));
    SA_HIP(h, hipEventRecord(h->stage_done[slot], h->ctl));
    h->stage_used[slot] = true;
    SA_HIP(h, hipEventRecord(h->uploaded, h->ctl));
    ++h->upload_gen;
    return SA_OK;
}

// One process call = begin_call, launches on c.stream with workspace slot c.slot, end_call.
//   ordered mode: c.stream is the caller's stream; the call is ordered after pending table uploads and, if the
//     caller switched streams, after the handle's earlier launches; end_call records `launched` on it.
//   overlap mode (sa_set_overlap(h, d), d > 1): call k runs on internal stream k % d behind a fork event taken from
//     the caller's stream BEFORE that stream is made to wait for call k-d+1 (the join): kernel k depends on
//     what the caller enqueued before call k, not on kernels k-1 .. k-d+1, and may run beside them.
struct CallCtx {
    hipStream_t stream;
    hipEvent_t start;         // bound to the call's first kernel while sa_set_profiling is on, else null
    hipEvent_t stop;          // bound to the call's last kernel by the launcher (null inside a stream capture)
    int slot;
    bool overlapped, captured;
};

// First overlapped call from a caller stream: an internal stream that shares a hardware queue with the CALLER's
// stream is as bad as two internal streams on one queue (the join waits of the caller's stream sit in front of
// the internal stream's next kernel: the six-handle run of profiles/r3_overlap_streams.txt), and the caller's
// stream is only known here.  Every internal stream is probed against it and replaced if they run in order.
// Costs a host wait for the caller stream's earlier work plus ~0.3 ms per internal stream, once per (handle,
// caller stream).
int fit_overlap_streams(sa_handle *h, hipStream_t user)
{
    for (int i = 0; i < h->overlap; ++i) {
        hipError_t pe = hipSuccess;
        const int r = streams_run_side_by_side(user, h->ov_stream[i], &pe);
        if (r < 0) return fail(h, SA_EHIP, "overlap: stream probe", pe);
        if (r == 1) continue;
        hipStream_t avoid[sa_handle::kMaxOverlap + 1] = {user};
        int n = 1;
        for (int j = 0; j < h->overlap; ++j)
            if (j != i) avoid[n++] = h->ov_stream[j];
        hipStream_t repl = nullptr;
        const int rc = pick_stream(h, avoid, n, &repl);
        if (rc != SA_OK) return rc;
        if (h->ov_used[i]) SA_HIP(h, hipEventSynchronize(h->ov_done[i]));      // the old stream's work is over
        (void)hipStreamDestroy(h->ov_stream[i]);
        h->ov_stream[i] = repl;
        h->ov_seen_gen[i] = h->upload_gen - 1;                                   // the new stream has seen no upload
    }
    h->ov_fit_stream = user;
    h->ov_fit_valid = true;
    return SA_OK;
}

int begin_call(sa_handle *h, hipStream_t user, CallCtx *c)
{
    c->stream = user;
    c->start = nullptr;
    c->slot = 0;
    c->overlapped = false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    SA_HIP(h, hipStreamIsCapturing(user, &cs));
    c->captured = cs != hipStreamCaptureStatusNone;
    if (c->captured) {
        h->capture_open = true;
        h->capture_stream = user;
    } else if (h->capture_open && h->capture_stream == user) {
        h->capture_open = false;             // that stream's capture has ended
        h->capture_stream = nullptr;
    }
    if (h->overlap > 1) {
        if (c->captured)
            return fail(h, SA_ESTATE, "overlapped launches (sa_set_overlap > 1) cannot be captured into a graph");
        if (!h->ov_fit_valid || h->ov_fit_stream != user) {
            const int rc = fit_overlap_streams(h, user);
            if (rc != SA_OK) return rc;
        }
        const int d = h->overlap, slot = (int)(h->ov_calls % (unsigned)d);
        SA_HIP(h, hipEventRecord(h->ov_fork[slot], user));
        // join: the call issued d-1 calls ago (the next user of the oldest slot is the call after this one)
        const int join = (slot + 1) % d;
        if (h->ov_unjoined[join]) {
            SA_HIP(h, hipStreamWaitEvent(user, h->ov_done[join], 0));
            h->ov_unjoined[join] = false;
        }
        SA_HIP(h, hipStreamWaitEvent(h->ov_stream[slot], h->ov_fork[slot], 0));
        // (ordered-mode launches made before the switch to overlap mode have completed: sa_set_overlap waited)
        if (h->ov_seen_gen[slot] != h->upload_gen) {
            SA_HIP(h, hipStreamWaitEvent(h->ov_stream[slot], h->uploaded, 0));
            h->ov_seen_gen[slot] = h->upload_gen;
        }
        c->stream = h->ov_stream[slot];
        c->stop = h->ov_done[slot];
        c->slot = slot;
        c->overlapped = true;
        return SA_OK;
    }
    if (h->have_last_stream && h->last_stream != user) {
        if (h->launched_valid) SA_HIP(h, hipStreamWaitEvent(user, h->launched, 0));
        h->seen_gen = h->upload_gen - 1;     // the new stream has not seen the last upload either
    }
    if (h->seen_gen != h->upload_gen) {
        if (h->upload_gen) SA_HIP(h, hipStreamWaitEvent(user, h->uploaded, 0));
        h->seen_gen = h->upload_gen;
    }
    h->last_stream = user;
    h->have_last_stream = true;
    if (!c->captured && !h->prof_stop.empty()) {          // timed call: the ring's next pair; `launched` follows it
        const size_t i = (size_t)(h->prof_calls % h->prof_stop.size());
        c->start = h->prof_start[i];
        h->launched = h->prof_stop[i];
    }
    // a captured record would tie the event to the graph; replays are ordered by the caller (include/specan.h)
    c->stop = c->captured ? nullptr : h->launched;
    return SA_OK;
}

int end_call(sa_handle *h, const CallCtx &c)
{
    if (c.overlapped) {
        h->ov_used[c.slot] = true;
        h->ov_unjoined[c.slot] = true;
        ++h->ov_calls;
        return SA_OK;
    }
    if (!c.captured) {
        h->launched_valid = true;
        if (!h->prof_stop.empty()) ++h->prof_calls;
    }
    return SA_OK;
}

// Workspace of a launch slot, grown without touching launches in flight (see sa_handle::retired).
// `geometric`: grow by at least half (process calls with creeping batch sizes); exact sizing where the size is copied
// from another slot -- sa_set_overlap gave every slot max(the others, 1.5 x its own), and two slots leap-frogged each
// other by a factor 1.5 per mode change until hipMalloc failed (found by a 10-minute soak, seed 77).
int ensure_work(sa_handle *h, int slot, int frames, bool captured, bool geometric = true)
{
    if (frames <= h->work_frames[slot]) return SA_OK;
    if (captured) return fail(h, SA_ESTATE, "workspace growth inside a stream capture: call sa_reserve() first");
    long want = frames, geo = (long)h->work_frames[slot] + h->work_frames[slot] / 2;
    if (geometric && geo > want) want = geo;
    void *p = nullptr;
    SA_HIP(h, hipMalloc(&p, (size_t)want * SA_NPTS * sizeof(int16_t)));
    if (h->d_work[slot]) h->retired.push_back(h->d_work[slot]);
    h->d_work[slot] = (int16_t *)p;
    h->work_frames[slot] = (int)want;
    return SA_OK;
}

void default_window_f64(std::vector<double> &w)
{
    w.resize(SA_NPTS);
    for (int i = 0; i < SA_NPTS; ++i)   // scripts/hann_coeff.py:3-4
        w[i] = 0.5 * (1.0 - std::cos(2.0 * M_PI * (double)i / (double)(SA_NPTS - 1)));
}

void default_rom(std::vector<int16_t> &rom)
{
    std::vector<double> w;
    default_window_f64(w);
    rom.resize(SA_NPTS);
    for (int i = 0; i < SA_NPTS; ++i) {   // scripts/hann_coeff.py:5 (rint, int16 wrap: quirk Q1)
        const double r = std::rint((w[i] - 0.5) * 65536.0);
        rom[i] = (int16_t)(uint16_t)((int32_t)r & 0xFFFF);
    }
}

// half = 0.5 * window (exact scaling, undone by the split step).  Two device copies, each arranged so
// that the kernel's loads are coalesced 16-byte accesses in the layout it computes in:
//   tr (IIR kernels, chunk layout):   tr[g][t] = half[64t + 4g .. +3]
//   pa (no-IIR kernel, pass-A layout): pa[p][t] = half[512(2p)+2t], [..+1], half[512(2p+1)+2t], [..+1]
void transpose_window(const std::vector<float> &half, std::vector<float> &tr)
{
    tr.resize(SA_NPTS);
    for (int t = 0; t < 256; ++t)
        for (int g = 0; g < 16; ++g)
            for (int e = 0; e < 4; ++e) tr[(g * 256 + t) * 4 + e] = half[64 * t + 4 * g + e];
}

void pass_a_window(const std::vector<float> &half, std::vector<float> &pa)
{
    pa.resize(SA_NPTS);
    for (int p = 0; p < 16; ++p)
        for (int t = 0; t < 256; ++t) {
            float *o = &pa[(p * 256 + t) * 4];
            o[0] = half[512 * (2 * p) + 2 * t];
            o[1] = half[512 * (2 * p) + 2 * t + 1];
            o[2] = half[512 * (2 * p + 1) + 2 * t];
            o[3] = half[512 * (2 * p + 1) + 2 * t + 1];
        }
}

int rebuild_plans(sa_handle *h);

int upload_window_half(sa_handle *h, const std::vector<float> &half)
{
    std::vector<float> tr, pa;
    transpose_window(half, tr);
    pass_a_window(half, pa);
    h->half_win = half;
    int rc = upload(h, h->d_win_b, pa.data(), sizeof(float) * SA_NPTS);
    if (rc != SA_OK) return rc;
    rc = upload(h, h->d_win_t, tr.data(), sizeof(float) * SA_NPTS);
    if (rc != SA_OK) return rc;
    return rebuild_plans(h);          // each plan carries its own (gain-scaled) copy of the window
}

// Is w[n] = a0 - a1 cos(2 pi n / (N-1)) to within float rounding?  Least-squares fit of (a0, a1) in double, then
// the residual against 1.5e-7 of the window's peak: Hann, Hamming and every other two-term cosine window pass,
// anything else (Blackman, Kaiser, rectangular with a taper, ...) keeps the table.
bool fit_cosine_window(const float *w, double out[2])
{
    const double theta = 2.0 * M_PI / (double)(SA_NPTS - 1);
    double s1 = 0, sc = 0, scc = 0, sw = 0, swc = 0, peak = 0;
    for (int n = 0; n < SA_NPTS; ++n) {
        const double c = -std::cos(theta * n), v = (double)w[n];
        s1 += 1.0; sc += c; scc += c * c; sw += v; swc += v * c;
        peak = std::fmax(peak, std::fabs(v));
    }
    const double det = s1 * scc - sc * sc;
    if (!(det > 0.0) || !(peak > 0.0) || !std::isfinite(peak)) return false;
    const double a0 = (sw * scc - swc * sc) / det, a1 = (s1 * swc - sc * sw) / det;
    double worst = 0;
    for (int n = 0; n < SA_NPTS; ++n) worst = std::fmax(worst, std::fabs((double)w[n] - (a0 - a1 * std::cos(theta * n))));
    if (!(worst <= 1.5e-7 * peak)) return false;
    out[0] = a0;
    out[1] = a1;
    return true;
}

int set_window_f32_from(sa_handle *h, const float *w)
{
    std::vector<float> half(SA_NPTS);
    for (int i = 0; i < SA_NPTS; ++i) half[i] = 0.5f * w[i];
    h->win_is_cos = fit_cosine_window(w, h->win_cos);
    return upload_window_half(h, half);
}

// flat float view for tests (layout documented in include/specan.h, sa_iir_plan_from_sos): header, the six
// sections' constants, then the predictor taps m[6][16][2], their half-chunk matrices p16[6][4] and the per-lane
// matrices p[6][16][4] (every matrix row-major here)
int export_plan(const SaIirK &p, const SaIirLaneTab &lt, float *out, int cap)
{
    std::vector<float> v;
    auto put_i = [&](int x) { float f; std::memcpy(&f, &x, 4); v.push_back(f); };
    put_i(p.nsec); put_i(p.unit); v.push_back(p.gain); put_i(p.wingen);
    for (int s = 0; s < SA_MAXSEC; ++s) {
        const SaIirSecK &k = p.sec[s];
        for (int i = 0; i < 5; ++i) v.push_back(k.c[i]);
        put_i(k.flags); v.push_back(k.pad[0]); v.push_back(k.pad[1]);
        auto rm = [&](const float *m) { v.push_back(m[0]); v.push_back(m[2]); v.push_back(m[1]); v.push_back(m[3]); };   // stored column-major
        rm(k.pc);
        rm(k.mback);
        for (int i = 0; i < 4; ++i) rm(k.plev[i]);
        for (int i = 0; i < 4; ++i) rm(k.prow[i]);
    }
    for (int s = 0; s < SA_MAXSEC; ++s) {
        const float (*m)[2] = s == 0 ? p.m0 : p.sec[s - 1].mnext;
        v.insert(v.end(), &m[0][0], &m[0][0] + 2 * SA_PRED_TAPS);
    }
    for (int s = 0; s < SA_MAXSEC; ++s) {
        const float *m = s == 0 ? p.p16_0 : p.sec[s - 1].p16next;
        v.push_back(m[0]); v.push_back(m[2]); v.push_back(m[1]); v.push_back(m[3]);
    }
    for (int sct = 0; sct < SA_MAXSEC; ++sct)
        for (int i = 0; i < 16; ++i) {
            const float *m = lt.p[sct][i];
            v.push_back(m[0]); v.push_back(m[2]); v.push_back(m[1]); v.push_back(m[3]);
        }
    const int n = (int)v.size();
    if (out && cap > 0) std::memcpy(out, v.data(), sizeof(float) * (size_t)(cap < n ? cap : n));
    return n;
}

int set_custom_plan(sa_handle *h, const double *sos_norm, int nsec)
{
    std::memset(h->sos_custom, 0, sizeof h->sos_custom);
    std::memcpy(h->sos_custom, sos_norm, sizeof(double) * 6 * (size_t)nsec);
    h->nsec_custom = nsec;
    const double *cw = h->win_is_cos ? h->win_cos : nullptr;
    build_plan(h->sos_custom, nsec, &h->plan_custom, &h->lt_custom, h->half_win.data(), cw);
    return upload(h, h->d_lt_custom, &h->lt_custom, sizeof(SaIirLaneTab));
}

int rebuild_plans(sa_handle *h)
{
    double sos[36];
    sos_from_q7(kDefaultQ7, sos);
    const double *cw = h->win_is_cos ? h->win_cos : nullptr;
    build_plan(sos, 6, &h->plan_default, &h->lt_default, h->half_win.data(), cw);
    int rc = upload(h, h->d_lt_default, &h->lt_default, sizeof(SaIirLaneTab));
    if (rc != SA_OK) return rc;
    build_plan(h->sos_custom, h->nsec_custom, &h->plan_custom, &h->lt_custom, h->half_win.data(), cw);
    return upload(h, h->d_lt_custom, &h->lt_custom, sizeof(SaIirLaneTab));
}

}  // namespace

extern "C" {

int sa_abi_version(void) { return SA_ABI_VERSION; }

const char *sa_last_error(const sa_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int sa_create(int device, sa_handle **out)
{
    if (!out) return fail(nullptr, SA_EINVAL, "sa_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, SA_EHIP, "sa_create: no usable HIP device (this library has no CPU fallback)", e);
    if (device < 0 || device >= ndev) return fail(nullptr, SA_EINVAL, "sa_create: device index out of range");
    sa_handle *h = new (std::nothrow) sa_handle();
    if (!h) return fail(nullptr, SA_ENOMEM, "sa_create: out of host memory");
    h->device = device;
#define SA_HIPC(call)                                                      \
    do {                                                                   \
        hipError_t e_ = (call);                                            \
        if (e_ != hipSuccess) {                                            \
            fail(nullptr, SA_EHIP, #call, e_);                             \
            sa_destroy(h);                                                 \
            return SA_EHIP;                                                \
        }                                                                  \
    } while (0)
    SA_HIPC(hipSetDevice(device));
    SA_HIPC(hipStreamCreateWithFlags(&h->ctl, hipStreamNonBlocking));
    SA_HIPC(hipEventCreateWithFlags(&h->launched_own, hipEventDisableTiming));
    h->launched = h->launched_own;
    SA_HIPC(hipEventCreateWithFlags(&h->uploaded, hipEventDisableTiming));
    for (int i = 0; i < sa_handle::kStage; ++i) {
        SA_HIPC(hipHostMalloc(&h->stage[i], kStageBytes, hipHostMallocDefault));
        SA_HIPC(hipEventCreateWithFlags(&h->stage_done[i], hipEventDisableTiming));
    }
    SA_HIPC(hipMalloc(&h->d_win_b, sizeof(float) * SA_NPTS));
    SA_HIPC(hipMalloc(&h->d_win_t, sizeof(float) * SA_NPTS));
    SA_HIPC(hipMalloc(&h->d_twT, sizeof(float4) * 6 * 256));
    SA_HIPC(hipMalloc(&h->d_twB, sizeof(float4) * 8 * 16));
    SA_HIPC(hipMalloc(&h->d_twC, sizeof(float2) * 25));
    SA_HIPC(hipMalloc(&h->d_lt_default, sizeof(SaIirLaneTab)));
    SA_HIPC(hipMalloc(&h->d_lt_custom, sizeof(SaIirLaneTab)));
    SA_HIPC(hipMalloc(&h->d_rom, sizeof(int16_t) * SA_NPTS));
    SA_HIPC(hipMalloc(&h->d_twq, sizeof(uint2) * SA_NPTS));
    SA_HIPC(hipMalloc(&h->d_twrec, sizeof(uint4) * 2 * kSaTwRecs));

    // float tables
    {
        std::vector<double> w;
        default_window_f64(w);
        std::vector<float> half(SA_NPTS);
        for (int i = 0; i < SA_NPTS; ++i) half[i] = (float)(0.5 * w[i]);
        std::vector<float> tr, pa;
        transpose_window(half, tr);
        pass_a_window(half, pa);
        h->half_win = half;
        SA_HIPC(hipMemcpy(h->d_win_b, pa.data(), sizeof(float) * SA_NPTS, hipMemcpyHostToDevice));
        SA_HIPC(hipMemcpy(h->d_win_t, tr.data(), sizeof(float) * SA_NPTS, hipMemcpyHostToDevice));
        std::vector<float4> ta(6 * 256), tb(8 * 16);
        std::vector<float2> tc(25);
        auto w8192 = [](long e) {                          // exp(-2 pi i e / 8192), e reduced first (exact)
            const double a = -2.0 * M_PI * (double)(e % 8192) / 8192.0;
            return make_float2((float)std::cos(a), (float)std::sin(a));
        };
        for (int t = 0; t < 256; ++t) {                    // per-thread anchors (SaF32Tables::twT)
            const int k[10] = {1, 2, 3, 4, 5, 6, 7, 8, 16, 24};
            for (int i = 0; i < 5; ++i) {
                const float2 u = w8192((long)k[2 * i] * t), v = w8192((long)k[2 * i + 1] * t);
                ta[i * 256 + t] = make_float4(u.x, u.y, v.x, v.y);
            }
            const double ap = -2.0 * M_PI * (double)(4 * t) / 16384.0;
            const double an = -2.0 * M_PI * (double)(4 * ((t + 1) & 255)) / 16384.0;      // (1, 0) for t = 255
            ta[5 * 256 + t] = make_float4((float)std::cos(ap), (float)std::sin(ap), (float)std::cos(an), (float)std::sin(an));
        }
        for (int pp = 0; pp < 8; ++pp)
            for (int b = 0; b < 16; ++b) {
                const double a0 = -2.0 * M_PI * (double)(2 * pp * b) / 256.0;
                const double a1 = -2.0 * M_PI * (double)((2 * pp + 1) * b) / 256.0;
                tb[pp * 16 + b] = make_float4((float)std::cos(a0), (float)std::sin(a0), (float)std::cos(a1), (float)std::sin(a1));
            }
        for (int blk = 0; blk < 5; ++blk)                   // block 4 = bin 4096 only (the seam of the last group)
            for (int e = 0; e < 5; ++e) {
                const double ang = -2.0 * M_PI * (double)(1024 * blk + e) / 16384.0;
                tc[blk * 5 + e] = make_float2((float)std::cos(ang), (float)std::sin(ang));
            }
        SA_HIPC(hipMemcpy(h->d_twT, ta.data(), sizeof(float4) * ta.size(), hipMemcpyHostToDevice));
        SA_HIPC(hipMemcpy(h->d_twB, tb.data(), sizeof(float4) * tb.size(), hipMemcpyHostToDevice));
        SA_HIPC(hipMemcpy(h->d_twC, tc.data(), sizeof(float2) * tc.size(), hipMemcpyHostToDevice));
    }
    // IIR plans: default = the fixed ALPHA/BETA cascade as real taps; custom = cleared coefficients
    {
        double sos[36];
        sos_from_q7(kDefaultQ7, sos);
        build_plan(sos, 6, &h->plan_default, &h->lt_default, h->half_win.data(), h->win_cos);
        SA_HIPC(hipMemcpy(h->d_lt_default, &h->lt_default, sizeof(SaIirLaneTab), hipMemcpyHostToDevice));
        sos_from_q7(h->c12_custom, sos);
        std::memcpy(h->sos_custom, sos, sizeof sos);
        h->nsec_custom = 6;
        build_plan(sos, 6, &h->plan_custom, &h->lt_custom, h->half_win.data(), h->win_cos);
        SA_HIPC(hipMemcpy(h->d_lt_custom, &h->lt_custom, sizeof(SaIirLaneTab), hipMemcpyHostToDevice));
    }
    // integer tables
    {
        default_rom(h->rom);
        SA_HIPC(hipMemcpy(h->d_rom, h->rom.data(), sizeof(int16_t) * SA_NPTS, hipMemcpyHostToDevice));
        std::vector<uint2> tq(SA_NPTS);
        for (int m = 0; m < SA_NPTS; ++m) {   // SA-FXFFT-1 twiddles: clamp16(rint(32768 cos)), clamp16(rint(-32768 sin))
            const double a = 2.0 * M_PI * (double)m / (double)SA_NPTS;
            long wr = std::lrint(32768.0 * std::cos(a)), wi = std::lrint(-32768.0 * std::sin(a));
            wr = wr > 32767 ? 32767 : (wr < -32768 ? -32768 : wr);
            wi = wi > 32767 ? 32767 : (wi < -32768 ? -32768 : wi);
            // second word (-wi, wr): the operand of the two-term dot product for the real part.  -wi does not fit
            // for wi = -32768 (exponents 4082..4110); the kernel never takes the second word of those entries
            // (fx_butterfly: `wide1`, `wide2`, `wide3`)
            const long nwi = -wi > 32767 ? 32767 : -wi;
            // the kernel's compile-time choice of the butterflies that avoid the second word rests on this range
            // (margins: 32768 sin = 32767.528 at 4082 and 4110, 32767.458 at 4081 and 4111; the threshold is .5)
            if (wi == -32768 && (m < 4082 || m > 4110)) {
                g_create_error = "sa_create: twiddle table: wi = -32768 outside exponents 4082..4110";
                sa_destroy(h);
                return SA_ESTATE;
            }
            tq[m].x = ((uint32_t)wr & 0xFFFFu) | ((uint32_t)wi << 16);
            tq[m].y = ((uint32_t)nwi & 0xFFFFu) | ((uint32_t)wr << 16);
        }
        SA_HIPC(hipMemcpy(h->d_twq, tq.data(), sizeof(uint2) * SA_NPTS, hipMemcpyHostToDevice));
        // One 32-byte record {w(e), w(2e), w(3e), pad} per butterfly of the stages whose exponents differ from lane to
        // lane: a lane's three twiddles are one contiguous read instead of three gathers at strides 8, 16 and 24 bytes.
        std::vector<uint4> rec(2 * kSaTwRecs);
        for (int r = 0; r < kSaTwRecs; ++r) {
            const int e = r < 4096 ? r : (r < 5120 ? 4 * (r - 4096) : 16 * (r - 5120));
            rec[2 * r] = make_uint4(tq[e].x, tq[e].y, tq[2 * e].x, tq[2 * e].y);
            rec[2 * r + 1] = make_uint4(tq[3 * e].x, tq[3 * e].y, 0u, 0u);
        }
        SA_HIPC(hipMemcpy(h->d_twrec, rec.data(), sizeof(uint4) * 2 * kSaTwRecs, hipMemcpyHostToDevice));
    }
    SA_HIPC(hipDeviceSynchronize());          // creation only: the blocking copies above are complete
#undef SA_HIPC
    *out = h;
    return SA_OK;
}

int sa_destroy(sa_handle *h)
{
    if (!h) return SA_OK;
    (void)hipSetDevice(h->device);
    // this handle's work only, through handle-owned objects (the caller's streams may be gone already)
    if (h->launched_valid) (void)hipEventSynchronize(h->launched);
    for (int i = 0; i < sa_handle::kMaxOverlap; ++i) {
        if (h->ov_stream[i]) (void)hipStreamSynchronize(h->ov_stream[i]);
        if (h->ov_fork[i]) (void)hipEventDestroy(h->ov_fork[i]);
        if (h->ov_done[i]) (void)hipEventDestroy(h->ov_done[i]);
        if (h->ov_stream[i]) (void)hipStreamDestroy(h->ov_stream[i]);
    }
    if (h->ctl) (void)hipStreamSynchronize(h->ctl);
    for (int i = 0; i < sa_handle::kStage; ++i) {
        if (h->stage[i]) (void)hipHostFree(h->stage[i]);
        if (h->stage_done[i]) (void)hipEventDestroy(h->stage_done[i]);
    }
    for (hipEvent_t e : h->prof_start) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->prof_stop) (void)hipEventDestroy(e);
    if (h->launched_own) (void)hipEventDestroy(h->launched_own);
    if (h->uploaded) (void)hipEventDestroy(h->uploaded);
    if (h->ctl) (void)hipStreamDestroy(h->ctl);
    (void)hipFree(h->d_win_b);
    (void)hipFree(h->d_win_t);
    (void)hipFree(h->d_twT);
    (void)hipFree(h->d_twB);
    (void)hipFree(h->d_twC);
    (void)hipFree(h->d_lt_default);
    (void)hipFree(h->d_lt_custom);
    (void)hipFree(h->d_rom);
    (void)hipFree(h->d_twq);
    (void)hipFree(h->d_twrec);
    for (int i = 0; i < sa_handle::kMaxOverlap; ++i) (void)hipFree(h->d_work[i]);
    for (void *p : h->retired) (void)hipFree(p);
    delete h;
    return SA_OK;
}

int sa_reserve(sa_handle *h, int max_batch)
{
    if (!h) return SA_EINVAL;
    if (max_batch < 0) return fail(h, SA_ESHAPE, "sa_reserve: negative batch");
    SA_HIP(h, hipSetDevice(h->device));
    for (int i = 0; i < h->overlap; ++i) {
        const int rc = ensure_work(h, i, max_batch, false);
        if (rc != SA_OK) return rc;
    }
    return SA_OK;
}

int sa_set_overlap(sa_handle *h, int depth)
{
    if (!h) return SA_EINVAL;
    if (depth < 1 || depth > sa_handle::kMaxOverlap) return fail(h, SA_EINVAL, "sa_set_overlap: depth must be 1..4");
    { const int rc = control_allowed(h); if (rc != SA_OK) return rc; }
    if (depth == h->overlap) return SA_OK;
    if (depth > 1 && !h->prof_stop.empty())
        return fail(h, SA_ESTATE, "sa_set_overlap: launch timing (sa_set_profiling) is for stream-ordered launches; turn it off first");
    SA_HIP(h, hipSetDevice(h->device));
    // leave the old mode with nothing of the handle's in flight (host wait on the handle's own work only)
    if (h->launched_valid) SA_HIP(h, hipEventSynchronize(h->launched));
    for (int i = 0; i < sa_handle::kMaxOverlap; ++i) {
        if (h->ov_used[i]) SA_HIP(h, hipEventSynchronize(h->ov_done[i]));
        h->ov_unjoined[i] = false;
    }
    for (int i = 0; i < depth; ++i) {
        if (!h->ov_stream[i]) {
            const int rc = pick_stream(h, h->ov_stream, i, &h->ov_stream[i]);     // beside the streams the handle already has
            if (rc != SA_OK) return rc;
        }
        if (!h->ov_fork[i]) SA_HIP(h, hipEventCreateWithFlags(&h->ov_fork[i], hipEventDisableTiming));
        if (!h->ov_done[i]) SA_HIP(h, hipEventCreateWithFlags(&h->ov_done[i], hipEventDisableTiming));
        // every slot starts with the workspace the handle already has somewhere
        int most = 0;
        for (int j = 0; j < sa_handle::kMaxOverlap; ++j) most = h->work_frames[j] > most ? h->work_frames[j] : most;
        const int rc = ensure_work(h, i, most, false, /*geometric=*/false);
        if (rc != SA_OK) return rc;
    }
    h->overlap = depth;
    h->ov_calls = 0;
    h->ov_fit_valid = false;
    return SA_OK;
}

int sa_get_overlap(const sa_handle *h, int *depth)
{
    if (!h || !depth) return SA_EINVAL;
    *depth = h->overlap;
    return SA_OK;
}

int sa_debug_overlap_streams(sa_handle *h, void *stream, int *side_by_side)
{
    if (!h || !side_by_side) return SA_EINVAL;
    { const int rc = control_allowed(h); if (rc != SA_OK) return rc; }
    SA_HIP(h, hipSetDevice(h->device));
    *side_by_side = 1;
    if (h->overlap < 2) return SA_OK;
    for (int i = 0; i < h->overlap; ++i)
        for (int j = -1; j < i; ++j) {                       // j = -1: the caller's stream
            hipError_t pe = hipSuccess;
            const int r = streams_run_side_by_side(j < 0 ? (hipStream_t)stream : h->ov_stream[j], h->ov_stream[i], &pe);
            if (r < 0) return fail(h, SA_EHIP, "sa_debug_overlap_streams", pe);
            if (r == 0) *side_by_side = 0;
        }
    return SA_OK;
}

int sa_set_profiling(sa_handle *h, int ring)
{
    if (!h) return SA_EINVAL;
    if (ring < 0 || ring > 65536) return fail(h, SA_EINVAL, "sa_set_profiling: ring must be 0..65536");
    { const int rc = control_allowed(h); if (rc != SA_OK) return rc; }
    if (ring > 0 && h->overlap > 1)
        return fail(h, SA_ESTATE, "sa_set_profiling: launch timing is for stream-ordered launches (sa_set_overlap(h, 1) first)");
    SA_HIP(h, hipSetDevice(h->device));
    // nothing of the handle's in flight while the completion event changes hands
    if (h->launched_valid) SA_HIP(h, hipEventSynchronize(h->launched));
    h->launched_valid = false;
    h->launched = h->launched_own;
    for (hipEvent_t e : h->prof_start) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->prof_stop) (void)hipEventDestroy(e);
    h->prof_start.clear();
    h->prof_stop.clear();
    h->prof_calls = 0;
    for (int i = 0; i < ring; ++i) {
        hipEvent_t a = nullptr, b = nullptr;
        hipError_t e = hipEventCreate(&a);
        if (e == hipSuccess) e = hipEventCreate(&b);
        if (e != hipSuccess) {
            if (a) (void)hipEventDestroy(a);
            return fail(h, SA_EHIP, "sa_set_profiling: hipEventCreate", e);
        }
        h->prof_start.push_back(a);
        h->prof_stop.push_back(b);
    }
    return SA_OK;
}

int sa_profile_read(sa_handle *h, float *ms, int cap)
{
    if (!h) return SA_EINVAL;
    if (cap < 0 || (cap > 0 && !ms)) return fail(h, SA_EINVAL, "sa_profile_read: bad buffer");
    if (h->prof_stop.empty()) return fail(h, SA_ESTATE, "sa_profile_read: sa_set_profiling is off");
    SA_HIP(h, hipSetDevice(h->device));
    const unsigned long long n = h->prof_stop.size();
    unsigned long long have = h->prof_calls < n ? h->prof_calls : n;
    if (have > (unsigned long long)cap) have = (unsigned long long)cap;
    for (unsigned long long j = 0; j < have; ++j) {
        const size_t i = (size_t)((h->prof_calls - have + j) % n);
        SA_HIP(h, hipEventSynchronize(h->prof_stop[i]));
        SA_HIP(h, hipEventElapsedTime(&ms[j], h->prof_start[i], h->prof_stop[i]));
    }
    return (int)have;
}

int sa_flush(sa_handle *h, void *stream)
{
    if (!h) return SA_EINVAL;
    SA_HIP(h, hipSetDevice(h->device));
    for (int i = 0; i < sa_handle::kMaxOverlap; ++i)
        if (h->ov_unjoined[i]) {
            SA_HIP(h, hipStreamWaitEvent((hipStream_t)stream, h->ov_done[i], 0));
            h->ov_unjoined[i] = false;
        }
    return SA_OK;
}

int sa_set_filter_mode(sa_handle *h, uint8_t cmd)
{
    if (!h) return SA_EINVAL;
    if (cmd != SA_FILTER_DEFAULT && cmd != SA_FILTER_CUSTOM && cmd != SA_FILTER_NONE && cmd != SA_FILTER_WIDE)
        return fail(h, SA_EINVAL, "sa_set_filter_mode: not a filter-select byte (0x00, 0xA1, 0xB1, 0xA2)");
    { const int rc = control_allowed(h); if (rc != SA_OK) return rc; }
    h->filter_mode = cmd;
    return SA_OK;
}

int sa_get_filter_mode(const sa_handle *h, uint8_t *cmd)
{
    if (!h || !cmd) return SA_EINVAL;
    *cmd = h->filter_mode;
    return SA_OK;
}

int sa_load_coeffs_q7(sa_handle *h, const int8_t c[12])
{
    if (!h) return SA_EINVAL;
    if (!c) return fail(h, SA_EINVAL, "sa_load_coeffs_q7: NULL coefficients");
    { const int rc = control_allowed(h); if (rc != SA_OK) return rc; }
    std::memcpy(h->c12_custom, c, 12);
    double sos[36];
    sos_from_q7(h->c12_custom, sos);
    return set_custom_plan(h, sos, 6);
}

int sa_get_coeffs_q7(const sa_handle *h, int8_t c[12])
{
    if (!h || !c) return SA_EINVAL;
    std::memcpy(c, h->c12_custom, 12);
    return SA_OK;
}

int sa_feed_command_bytes_ex(sa_handle *h, const uint8_t *bytes, size_t n, sa_cmd_events *ev)
{
    if (!h) return SA_EINVAL;
    if (!bytes && n) return fail(h, SA_EINVAL, "sa_feed_command_bytes: NULL bytes");
    { const int rc = control_allowed(h); if (rc != SA_OK) return rc; }
    for (size_t i = 0; i < n; ++i) {
        const uint8_t b = bytes[i];
        if (h->rx_count >= 0) {                      // ACQUIRE: busy, byte is a coefficient; neither command_control
            h->rx_buf[h->rx_count++] = (int8_t)b;    // nor sequ_2 sees it (uart_rx_valid and not busy,
            if (h->rx_count == 12) {                 // imp/dsp_system_top.vhd:644, new/command_control.vhd:51)
                h->rx_count = -1;
                const int rc = sa_load_coeffs_q7(h, h->rx_buf);
                if (rc != SA_OK) return rc;
                if (ev) { ++ev->n_uploads; ev->control_changed = 1; }
            }
            continue;
        }
        switch (b) {                                 // IDLE: command decode (command_control.vhd:53-62, sequ2.vhd:82-96)
            case SA_CMD_FILTER_UPDATE: h->rx_count = 0; break;
            case SA_FILTER_DEFAULT:
            case SA_FILTER_CUSTOM:
            case SA_FILTER_NONE:
                if (ev && h->filter_mode != b) ev->control_changed = 1;
                h->filter_mode = b;
                break;
            case SA_CMD_RESET: {                     // rst: mode B1 (:50), coefficients cleared (filter_iir12_cust.vhd:51-52),
                h->filter_mode = SA_FILTER_NONE;     // Ethernet transport (sequ2.vhd:85-86)
                h->transport = SA_CMD_ETHERNET_MODE;
                const int8_t z[12] = {0};
                const int rc = sa_load_coeffs_q7(h, z);
                if (rc != SA_OK) return rc;
                if (ev) { ++ev->n_reset; ev->control_changed = 1; }
                break;
            }
            case SA_CMD_ETHERNET_MODE:
            case SA_CMD_UART_MODE: h->transport = b; break;
            case SA_CMD_START: if (ev) ++ev->n_start; break;
            case SA_CMD_UART_REQUEST: if (ev) ++ev->n_uart_request; break;
            default: break;                          // unknown bytes: no effect, like the RTL
        }
    }
    if (ev) ev->transport = h->transport;
    return SA_OK;
}

int sa_feed_command_bytes(sa_handle *h, const uint8_t *bytes, size_t n, int *n_frames_requested)
{
    sa_cmd_events ev;
    std::memset(&ev, 0, sizeof ev);
    const int rc = sa_feed_command_bytes_ex(h, bytes, n, &ev);
    if (n_frames_requested) *n_frames_requested += ev.n_uart_request;
    return rc;
}

int sa_get_transport(const sa_handle *h, uint8_t *cmd)
{
    if (!h || !cmd) return SA_EINVAL;
    *cmd = h->transport;
    return SA_OK;
}

int sa_load_sos_f64(sa_handle *h, const double *sos, int n_sections)
{
    if (!h) return SA_EINVAL;
    if (!sos) return fail(h, SA_EINVAL, "sa_load_sos: NULL sos");
    if (n_sections < 0 || n_sections > SA_MAXSEC) return fail(h, SA_EINVAL, "sa_load_sos: 0..6 sections");
    { const int rc = control_allowed(h); if (rc != SA_OK) return rc; }
    double norm[36];
    for (int s = 0; s < n_sections; ++s) {
        const double a0 = sos[6 * s + 3];
        if (a0 == 0.0 || !std::isfinite(a0)) return fail(h, SA_EINVAL, "sa_load_sos: a0 must be finite and non-zero");
        for (int i = 0; i < 6; ++i) norm[6 * s + i] = sos[6 * s + i] / a0;
    }
    return set_custom_plan(h, norm, n_sections);
}

int sa_load_sos_f32(sa_handle *h, const float *sos, int n_sections)
{
    if (!h) return SA_EINVAL;
    if (!sos) return fail(h, SA_EINVAL, "sa_load_sos: NULL sos");
    if (n_sections < 0 || n_sections > SA_MAXSEC) return fail(h, SA_EINVAL, "sa_load_sos: 0..6 sections");
    double d[36];
    for (int i = 0; i < 6 * n_sections; ++i) d[i] = (double)sos[i];
    return sa_load_sos_f64(h, d, n_sections);
}

int sa_load_sos_q14(sa_handle *h, const int16_t *sos, int n_sections)
{
    if (!h) return SA_EINVAL;
    if (!sos) return fail(h, SA_EINVAL, "sa_load_sos_q14: NULL sos");
    if (n_sections < 0 || n_sections > SA_MAXSEC) return fail(h, SA_EINVAL, "sa_load_sos_q14: 0..6 sections");
    { const int rc = control_allowed(h); if (rc != SA_OK) return rc; }
    std::memset(h->sos_q14, 0, sizeof h->sos_q14);
    std::memcpy(h->sos_q14, sos, sizeof(int16_t) * 6 * n_sections);
    h->nsec_q14 = n_sections;
    return SA_OK;
}

int sa_set_window_q15(sa_handle *h, const int16_t *w)
{
    if (!h) return SA_EINVAL;
    { const int rc = control_allowed(h); if (rc != SA_OK) return rc; }
    if (w) h->rom.assign(w, w + SA_NPTS); else default_rom(h->rom);
    return upload(h, h->d_rom, h->rom.data(), sizeof(int16_t) * SA_NPTS);
}

int sa_get_window_q15(const sa_handle *h, int16_t *w)
{
    if (!h || !w) return SA_EINVAL;
    std::memcpy(w, h->rom.data(), sizeof(int16_t) * SA_NPTS);
    return SA_OK;
}

int sa_set_window_f32(sa_handle *h, const float *w)
{
    if (!h) return SA_EINVAL;
    { const int rc = control_allowed(h); if (rc != SA_OK) return rc; }
    if (w) return set_window_f32_from(h, w);
    std::vector<double> d;
    default_window_f64(d);
    std::vector<float> half(SA_NPTS);
    for (int i = 0; i < SA_NPTS; ++i) half[i] = (float)(0.5 * d[i]);
    h->win_is_cos = true;
    h->win_cos[0] = h->win_cos[1] = 0.5;                  // scripts/hann_coeff.py:3-4
    return upload_window_half(h, half);
}

int sa_set_window_mode_q15(sa_handle *h, int mode)
{
    if (!h) return SA_EINVAL;
    if (mode != SA_WIN_RTL_SIGNED && mode != SA_WIN_HANN_U16) return fail(h, SA_EINVAL, "sa_set_window_mode_q15: bad mode");
    { const int rc = control_allowed(h); if (rc != SA_OK) return rc; }
    h->win_mode_q15 = mode;
    return SA_OK;
}

static int q15_params(sa_handle *h, SaQ15Params *p)
{
    std::memset(p, 0, sizeof(*p));
    p->win_mode = h->win_mode_q15;
    p->filter = h->filter_mode;
    p->nsec_wide = h->nsec_q14;
    if (h->filter_mode == SA_FILTER_DEFAULT) std::memcpy(p->c12, kDefaultQ7, 12);
    else std::memcpy(p->c12, h->c12_custom, 12);
    std::memcpy(p->sos_q14, h->sos_q14, sizeof p->sos_q14);
    if (h->filter_mode == SA_FILTER_WIDE && h->nsec_q14 == 0) p->filter = SA_FILTER_NONE;   // no sections = wire
    return SA_OK;
}

int sa_filter_q15(sa_handle *h, const int16_t *in, int16_t *out_time, int batch, void *stream)
{
    if (!h) return SA_EINVAL;
    if (batch < 0) return fail(h, SA_ESHAPE, "sa_filter_q15: negative batch");
    if (batch == 0) return SA_OK;
    if (!in || !out_time) return fail(h, SA_EINVAL, "sa_filter_q15: NULL tensor");
    SA_HIP(h, hipSetDevice(h->device));
    CallCtx c;
    { const int rc = begin_call(h, (hipStream_t)stream, &c); if (rc != SA_OK) return rc; }
    SaQ15Params p;
    q15_params(h, &p);
    const SaQ15Tables t = {h->d_rom, h->d_twq, h->d_twrec};
    SA_HIP(h, sa_launch_filter_q15(in, out_time, batch, p, t, c.stream, {c.start, c.stop}));
    return end_call(h, c);
}

int sa_process_q15(sa_handle *h, const int16_t *in, int16_t *out_iq, int batch, void *stream)
{
    if (!h) return SA_EINVAL;
    if (batch < 0) return fail(h, SA_ESHAPE, "sa_process_q15: negative batch");
    if (batch == 0) return SA_OK;
    if (!in || !out_iq) return fail(h, SA_EINVAL, "sa_process_q15: NULL tensor");
    SA_HIP(h, hipSetDevice(h->device));
    CallCtx c;
    { const int rc = begin_call(h, (hipStream_t)stream, &c); if (rc != SA_OK) return rc; }
    SaQ15Params p;
    q15_params(h, &p);
    const SaQ15Tables t = {h->d_rom, h->d_twq, h->d_twrec};
    if (p.filter == SA_FILTER_NONE) {
        SA_HIP(h, sa_launch_fft_q15(in, out_iq, batch, true, p, t, c.stream, {c.start, c.stop}));
        return end_call(h, c);
    }
    { const int rc = ensure_work(h, c.slot, batch, c.captured); if (rc != SA_OK) return rc; }
    // The WIDE cascade does not gain from overlapped launches (tools/q15_overlap_modes.py, profiles/r4_q15_helper_waves.txt):
    // its step is made of packed dot products, the integer FFT's twiddle products are too, and side by side the two starve
    // each other -- 7.5-7.9 M frames/s at depth 2 when left free against 8.3 M stream-ordered.  At depth 2 its cascade
    // therefore waits for the previous call of the handle (8.0 M; at depth 3 it runs free: 7.9-8.2 M).
    const bool wide = p.filter == SA_FILTER_WIDE;
    if (c.overlapped && wide && h->overlap == 2) {
        const int prev = (c.slot + h->overlap - 1) % h->overlap;
        if (h->ov_used[prev]) SA_HIP(h, hipStreamWaitEvent(c.stream, h->ov_done[prev], 0));
    }
    SA_HIP(h, sa_launch_filter_q15(in, h->d_work[c.slot], batch, p, t, c.stream, {c.start, nullptr}));
    SA_HIP(h, sa_launch_fft_q15(h->d_work[c.slot], out_iq, batch, false, p, t, c.stream, {nullptr, c.stop}));
    return end_call(h, c);
}

int sa_process_f32(sa_handle *h, const float *in, void *out, int batch, int out_kind, void *stream)
{
    if (!h) return SA_EINVAL;
    if (batch < 0) return fail(h, SA_ESHAPE, "sa_process_f32: negative batch");
    if (out_kind < SA_OUT_MAG_FULL || out_kind > SA_OUT_TIME) return fail(h, SA_EINVAL, "sa_process_f32: bad out_kind");
    if (batch == 0) return SA_OK;
    if (!in || !out) return fail(h, SA_EINVAL, "sa_process_f32: NULL tensor");
    if (h->filter_mode == SA_FILTER_WIDE)
        return fail(h, SA_ESTATE, "sa_process_f32: filter mode 0xA2 (Q2.14) belongs to the Q15 path; use 0xA1 with sa_load_sos_f32");
    SA_HIP(h, hipSetDevice(h->device));
    CallCtx c;
    { const int rc = begin_call(h, (hipStream_t)stream, &c); if (rc != SA_OK) return rc; }
    // The section coefficients and predictor taps travel by value in the kernel arguments (stream-ordered
    // by construction); the per-lane matrices and the window live in device memory (stream-ordered uploads).
    SaF32Tables t = {h->d_win_b, h->d_win_t, h->d_twT, h->d_twB, h->d_twC, h->d_lt_custom, nullptr};
    if (h->filter_mode == SA_FILTER_DEFAULT) {
        t.lanetab = h->d_lt_default;
        t.iir = &h->plan_default;
    } else if (h->filter_mode == SA_FILTER_CUSTOM) {
        t.lanetab = h->d_lt_custom;
        t.iir = &h->plan_custom;
    }
    SA_HIP(h, sa_launch_chain_f32(in, out, batch, out_kind, t, c.stream, {c.start, c.stop}));
    return end_call(h, c);
}

int sa_process_f32_i16(sa_handle *h, const int16_t *in, float scale, void *out, int batch, int out_kind, void *stream)
{
    if (!h) return SA_EINVAL;
    if (batch < 0) return fail(h, SA_ESHAPE, "sa_process_f32_i16: negative batch");
    if (out_kind < SA_OUT_MAG_FULL || out_kind > SA_OUT_TIME) return fail(h, SA_EINVAL, "sa_process_f32_i16: bad out_kind");
    if (!(scale == scale) || scale - scale != 0.f) return fail(h, SA_EINVAL, "sa_process_f32_i16: scale is not finite");
    if (batch == 0) return SA_OK;
    if (!in || !out) return fail(h, SA_EINVAL, "sa_process_f32_i16: NULL tensor");
    if (h->filter_mode == SA_FILTER_WIDE)
        return fail(h, SA_ESTATE, "sa_process_f32_i16: filter mode 0xA2 (Q2.14) belongs to the Q15 path; use 0xA1 with sa_load_sos_f32");
    SA_HIP(h, hipSetDevice(h->device));
    CallCtx c;
    { const int rc = begin_call(h, (hipStream_t)stream, &c); if (rc != SA_OK) return rc; }
    SaF32Tables t = {h->d_win_b, h->d_win_t, h->d_twT, h->d_twB, h->d_twC, h->d_lt_custom, nullptr};
    if (h->filter_mode == SA_FILTER_DEFAULT) {
        t.lanetab = h->d_lt_default;
        t.iir = &h->plan_default;
    } else if (h->filter_mode == SA_FILTER_CUSTOM) {
        t.lanetab = h->d_lt_custom;
        t.iir = &h->plan_custom;
    }
    SA_HIP(h, sa_launch_chain_f32_i16(in, scale, out, batch, out_kind, t, c.stream, {c.start, c.stop}));
    return end_call(h, c);
}

int sa_pack_frame(const int16_t *iq_host, uint8_t *frame_bytes)
{
    if (!iq_host || !frame_bytes) return SA_EINVAL;
    for (int i = 0; i < SA_NPTS * 2; ++i) {          // explicit little-endian, independent of the host
        const uint16_t v = (uint16_t)iq_host[i];
        frame_bytes[2 * i] = (uint8_t)(v & 0xFF);
        frame_bytes[2 * i + 1] = (uint8_t)(v >> 8);
    }
    return SA_OK;
}

int sa_debug_iir_plan_f32(const sa_handle *h, float *out, int cap)
{
    if (!h) return SA_EINVAL;
    const bool def = h->filter_mode == SA_FILTER_DEFAULT;
    return export_plan(def ? h->plan_default : h->plan_custom, def ? h->lt_default : h->lt_custom, out, cap);
}

int sa_iir_plan_from_sos(const double *sos, int n_sections, float *out, int cap)
{
    if (!sos || n_sections < 0 || n_sections > SA_MAXSEC) return SA_EINVAL;
    double norm[36];
    for (int s = 0; s < n_sections; ++s) {
        const double a0 = sos[6 * s + 3];
        if (a0 == 0.0 || !std::isfinite(a0)) return SA_EINVAL;
        for (int i = 0; i < 6; ++i) norm[6 * s + i] = sos[6 * s + i] / a0;
    }
    SaIirK p;
    std::vector<SaIirLaneTab> lt(1);
    build_plan(norm, n_sections, &p, &lt[0], nullptr);
    return export_plan(p, lt[0], out, cap);
}

}  // extern "C"
