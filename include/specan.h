/*
 * specan.h -- C ABI of the MI355X spectrum-analyser signal path (libspecan_hip.so).
 *
 * Drop-in boundary for the hot path of mfkiwl/fpga-real-time-fft-analyzer:
 *     hann_window  ->  filter_iir12 | filter_iir12_cust  ->  xfft_0 (16384-pt)  ->  frame bytes
 * The reference exposes no FFI; its boundary is the byte contract between the FPGA and
 * scripts/fft_analyzer_gui.py.  Every entry point below cites the reference interface it
 * replaces (paths relative to the reference root; new/ = SDR_v2.srcs/sources_1/new,
 * imp/ = SDR_v2.srcs/sources_1/imports/new, gui.py = scripts/fft_analyzer_gui.py).
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures: device pointers are `void*`-compatible raw
 *     pointers (tensor.data_ptr()), the stream is a `void*` holding a hipStream_t
 *     (torch.cuda.current_stream().cuda_stream); NULL = the null stream.
 *   - every call returns 0 (SA_OK) or a negative SA_E* code; sa_last_error() gives the text.
 *     No exception or abort crosses the ABI.  There is NO CPU fallback: without a usable GPU
 *     sa_create() fails with SA_EHIP.
 *   - the library never allocates, frees or retains caller tensors.  It owns the opaque handle,
 *     its device-side tables and (for the Q15 IIR modes) a workspace sized by sa_reserve().
 *   - a handle is not thread-safe; use one handle per (GPU, stream) -- different handles may be
 *     used from different host threads at the same time (one thread per GPU).
 *     Process calls are asynchronous on the given stream; mode / coefficient / window changes are
 *     stream-ordered: they apply to every process call issued after them and to none issued
 *     before.  No call after sa_create() synchronises the device: table uploads run on the
 *     handle's own control stream behind an event, so other handles and streams are not stalled
 *     (a window or coefficient upload may block the HOST briefly when more than four uploads are
 *     still waiting for their copies; a Q15 workspace that is outgrown is kept until sa_destroy()
 *     rather than freed, because hipFree synchronises the device).
 *   - device: a handle belongs to the GPU given to sa_create().  Every call that touches the GPU makes that
 *     device the calling thread's current HIP device (hipSetDevice) and leaves it so; the stream and the
 *     pointers passed to a process call must belong to it (not checked: a check costs more than a launch).
 *     One process per GPU (torch.distributed ranks) or one host thread per GPU both work.
 *   - stream lifetime: the library uses the stream passed to a process call only inside that call.
 *     The caller may destroy it afterwards, in any order with later calls and sa_destroy(): uploads,
 *     stream switches and sa_destroy() order themselves behind an event the handle owns, bound to
 *     the completion of the call's last kernel.
 *   - hipGraph capture: process calls are capturable in ordered mode once sa_reserve() has sized
 *     the workspace.  A captured call freezes the control state of capture time in its kernel
 *     arguments; control-plane calls are refused (SA_ESTATE, nothing changed) while that capture
 *     is open -- also when other, uncaptured calls have been made on other streams meanwhile -- and
 *     after any control-plane call graphs captured earlier must be captured again.  End a capture
 *     before destroying the capturing stream (the handle asks that stream whether it still captures).
 *     Work replayed from a graph is not tracked by the handle: order it yourself (e.g. synchronise
 *     the replay stream) before a control-plane call.
 *   - frame length is fixed: SA_N = 16384 samples (gui.py:43-44, imp/dsp_system_top.vhd:440,
 *     ip/xfft_0/xfft_0.xci:12).
 */
#ifndef SPECAN_H_
#define SPECAN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SA_N 16384
#define SA_FRAME_BYTES 65536          /* gui.py:42  FRAME_SIZE_BYTES */
#define SA_ABI_VERSION 4

/* error codes */
#define SA_OK       0
#define SA_EINVAL  -1   /* bad argument (NULL pointer, bad enum, bad section count) */
#define SA_ESHAPE  -2   /* bad batch / shape */
#define SA_EHIP    -3   /* HIP runtime error; text in sa_last_error() */
#define SA_ESTATE  -4   /* call not valid in the current state (e.g. mode needs coefficients) */
#define SA_ENOMEM  -5

/* filter-select command bytes: gui.py:35-37, decoded in new/command_control.vhd:53-58.
 * Power-on / reset value is SA_FILTER_NONE (new/command_control.vhd:31). */
#define SA_FILTER_DEFAULT 0x00   /* fixed ALPHA/BETA cascade, imp/filter_iir12.vhd + imp/filter_pkg.vhd:54-68 */
#define SA_FILTER_CUSTOM  0xA1   /* uploaded coefficients, new/filter_iir12_cust.vhd */
#define SA_FILTER_NONE    0xB1   /* window -> FFT directly */
#define SA_FILTER_WIDE    0xA2   /* build extension (not a reference byte): 6 independent Q2.14 sections */

/* other command bytes understood by sa_feed_command_bytes(): gui.py:28-33 */
#define SA_CMD_START          0x55
#define SA_CMD_UART_REQUEST   0xA5
#define SA_CMD_RESET          0xFF
#define SA_CMD_ETHERNET_MODE  0xEF
#define SA_CMD_UART_MODE      0xFE
#define SA_CMD_FILTER_UPDATE  0xF1   /* followed by 12 coefficient bytes, new/rx_filter_coeff.vhd:45-56 */

/* Q15 window modes */
#define SA_WIN_RTL_SIGNED 0   /* new/hann8192.vhd:36-39: ROM word used as signed Q15 (FPGA-exact) */
#define SA_WIN_HANN_U16   1   /* ROM + 32768 as unsigned Q16 Hann (the evident intent) */

/* float output layouts for sa_process_f32() */
#define SA_OUT_MAG_FULL   0   /* float  [B,16384]   |X[k]|, all N bins (upper half mirrored; gui.py:294-305 plots all N) */
#define SA_OUT_MAG_HALF   1   /* float  [B,8193]    |X[k]|, k = 0..N/2 */
#define SA_OUT_SPEC_HALF  2   /* float2 [B,8193]    X[k] = (re,im), k = 0..N/2 (numpy.fft.rfft layout) */
#define SA_OUT_TIME       3   /* float  [B,16384]   FFT input: window (+ IIR) output time series */

typedef struct sa_handle sa_handle;

/* ---- lifetime ------------------------------------------------------------------------- */
/* Stands for "power on the board": state = filter NONE, zero custom coefficients
 * (new/filter_iir12_cust.vhd:51-52), default Hann ROM (new/hann.vhd, scripts/hann_coeff.py:3-5). */
int sa_create(int device, sa_handle **out);
int sa_destroy(sa_handle *h);
int sa_abi_version(void);
const char *sa_last_error(const sa_handle *h);   /* h may be NULL: last sa_create() failure */

/* Pre-size the internal workspace for batches up to max_batch frames (Q15 IIR modes need
 * B*32 KiB, per launch slot in overlap mode).  Optional: process calls grow it on demand (not
 * capturable into a hipGraph then). */
int sa_reserve(sa_handle *h, int max_batch);

/* Overlapped launches (opt-in; build extension).  Frames are independent -- every frame starts from a zero
 * filter state: the RTL clears the biquad history whenever i_valid = '0' (new/filter_iir_cust.vhd:142-146,
 * SURVEY quirk Q5), which this build applies once per frame -- so consecutive batches need not
 * run one after the other.  With depth d > 1, process call k runs on an internal stream of the
 * handle (k mod d), ordered after everything the caller's stream held when the call was made but
 * NOT after calls k-1 .. k-d+1: the tail of one launch runs under the head of the next (on the
 * Q15 path the FFT of batch k-1 under the filter of batch k).  Contract in this mode:
 *   - the results of call k are visible to work enqueued on the caller's stream after call
 *     k+d-1 has been made, or after sa_flush(); until then the call's input AND output tensors
 *     must not be written, freed or read by the caller;
 *   - process calls cannot be captured into a hipGraph (SA_ESTATE);
 *   - control-plane calls stay stream-ordered: they apply to all later calls and to no earlier one.
 * depth = 1 (the default) is the strictly stream-ordered mode described above.  sa_set_overlap()
 * waits on the host for the handle's own outstanding work when the depth changes.
 * The internal streams are chosen so that they execute side by side with each other AND with the caller's stream:
 * the runtime maps streams onto a few hardware queues and two streams on one queue run in order (a handle with
 * such a pair was slower in overlap mode than without it).  The mapping is not exposed, so the library probes
 * candidate streams with a 100 us one-wave kernel: in sa_set_overlap() among themselves, and in the FIRST
 * overlapped process call made from a given caller stream against that stream -- that one call waits on the host
 * for the stream's earlier work and takes about a millisecond longer.  Best effort on a GPU busy with other work.
 * The handle remembers ONE fitted caller stream (it is compared, never dereferenced: the caller may have destroyed it):
 * overlapped calls that alternate between caller streams re-run the probe at every change -- keep a handle on one
 * caller stream, as the one-handle-per-(GPU, stream) rule above says. */
int sa_set_overlap(sa_handle *h, int depth /* 1..4 */);
int sa_get_overlap(const sa_handle *h, int *depth);
/* Introspection for tests: re-runs that probe on the handle's internal streams and `stream`;
 * *side_by_side = 1 if every pair overlaps. */
int sa_debug_overlap_streams(sa_handle *h, void *stream, int *side_by_side);
/* Make `stream` wait for every outstanding overlapped call of the handle (no host wait). */
int sa_flush(sa_handle *h, void *stream);

/* Launch timing (opt-in; measurement aid, no counterpart in the reference).  With ring = n > 0 every stream-ordered
 * process call binds a pair of timing events to the BEGIN of its first kernel and the END of its last one -- the
 * events ride on the dispatch packets themselves (hipExtLaunchKernel), nothing is put between two launches, so a train
 * of calls runs as it does untimed -- and the handle keeps the pairs of the last n calls.  sa_profile_read() waits on
 * the host for those calls and writes their device times in milliseconds, oldest first, into ms[0 .. return value)
 * (at most cap).  A call with one kernel (the float chain, the bypassed integer chain) reports that kernel's duration;
 * the integer chain with a cascade reports cascade + FFT including the gap between them.  Captured calls are not
 * timed.  ring = 0 turns it off.  Refused (SA_ESTATE) while sa_set_overlap is above 1, and sa_set_overlap(d > 1) is
 * refused while it is on: kernels that run beside each other have no per-call time.  Changing the ring waits on the
 * host for the handle's own outstanding work. */
int sa_set_profiling(sa_handle *h, int ring /* 0..65536 */);
int sa_profile_read(sa_handle *h, float *ms, int cap);

/* ---- control plane of the path (what the UART bytes do) -------------------------------- */
/* new/command_control.vhd:53-58: accepts SA_FILTER_DEFAULT / CUSTOM / NONE (and SA_FILTER_WIDE). */
int sa_set_filter_mode(sa_handle *h, uint8_t cmd);
int sa_get_filter_mode(const sa_handle *h, uint8_t *cmd);

/* 12 int8 coefficients in wire order [b0,b1,b2,a0,a1,a2] x 2 (gui.py:598-605), stored into
 * COEFF_IIR_CF(0..11) (new/filter_iir12_cust.vhd:54, ports :83-94).  Semantics per the RTL:
 * stage taps B2*x[n]+B1*x[n-1]+B0*x[n-2]-A0*y[n-2]-A1*y[n-1], each product >>7, 16-bit wrap;
 * set 0 drives stages 1,3,5 and set 1 stages 2,4,6; A2 is unused (new/filter_iir_cust.vhd:96-117).
 * The float path (sa_process_f32) uses the same taps as real numbers c/128. */
int sa_load_coeffs_q7(sa_handle *h, const int8_t c[12]);
int sa_get_coeffs_q7(const sa_handle *h, int8_t c[12]);

/* Byte-stream front door: the UART RX path (imp/uart_rx.vhd -> new/rx_filter_coeff.vhd:41-66
 * + new/command_control.vhd:51-62).  0xF1 starts a 12-byte coefficient upload during which no
 * byte is interpreted as a command; 0x00/0xA1/0xB1 select the filter; 0xFF resets (filter NONE,
 * coefficients cleared); 0x55/0xA5/0xEF/0xFE are accepted and counted but have no effect on the
 * signal path.  Unknown bytes are ignored, like the RTL.  *n_frames_requested (optional) is
 * incremented once per 0xA5 seen outside a coefficient upload: the UART read request of
 * imp/sequ2.vhd:216 (0x55 only starts the acquisition, new/command_control.vhd:58-60, and is
 * reported by sa_feed_command_bytes_ex). */
int sa_feed_command_bytes(sa_handle *h, const uint8_t *bytes, size_t n, int *n_frames_requested);

/* The same front door with everything a transport shim needs to stand where imp/sequ2.vhd stands.
 * Counters are ADDED to (zero the struct first); `transport` and `control_changed` are set. */
typedef struct sa_cmd_events {
    int n_start;          /* 0x55: start_aq pulse (new/command_control.vhd:58-60, :75) */
    int n_uart_request;   /* 0xA5: UART read command (imp/sequ2.vhd:216) */
    int n_reset;          /* 0xFF: reset_n pulse (new/command_control.vhd:56-57) */
    int n_uploads;        /* completed 0xF1 + 12-byte coefficient uploads (new/rx_filter_coeff.vhd:45-56) */
    int control_changed;  /* non-zero: filter select, coefficients or a reset changed what the path computes */
    uint8_t transport;    /* SA_CMD_ETHERNET_MODE or SA_CMD_UART_MODE after the last byte (imp/sequ2.vhd:82-96);
                             reset selects Ethernet (imp/sequ2.vhd:85-86) */
} sa_cmd_events;
int sa_feed_command_bytes_ex(sa_handle *h, const uint8_t *bytes, size_t n, sa_cmd_events *ev);
int sa_get_transport(const sa_handle *h, uint8_t *cmd);

/* North-star wide formats (not in the reference): up to 6 independent sections, scipy row order
 * [b0,b1,b2,a0,a1,a2], normalised by a0 on load.  f32/f64 feed sa_process_f32 in CUSTOM mode;
 * q14 (int16 Q2.14, a0 ignored) feeds sa_process_q15 in WIDE mode. */
int sa_load_sos_f32(sa_handle *h, const float *sos, int n_sections);
int sa_load_sos_f64(sa_handle *h, const double *sos, int n_sections);
int sa_load_sos_q14(sa_handle *h, const int16_t *sos, int n_sections);

/* Window tables (host pointers, copied).  NULL restores the default generated with the formula of
 * scripts/hann_coeff.py:3-5 (the Q15 table includes the int16 wrap of entries 8178..8205). */
int sa_set_window_q15(sa_handle *h, const int16_t *w /* [16384] or NULL */);
int sa_set_window_f32(sa_handle *h, const float *w /* [16384] or NULL */);
int sa_set_window_mode_q15(sa_handle *h, int mode /* SA_WIN_* */);
int sa_get_window_q15(const sa_handle *h, int16_t *w /* [16384] */);

/* ---- data plane ----------------------------------------------------------------------- */
/* Q15 path, bit-exact integer pipeline: in [B,16384] int16 device (samples as the XADC delivers
 * them, imp/dsp_system_top.vhd:435) -> out_iq [B,16384,2] int16 device = the 65536-byte frames of
 * imp/sequ2.vhd:153 / gui.py:250-260 (re lo,hi, im lo,hi).  FFT = SA-FXFFT-1:
 * stands where ip/xfft_0 stands; 1/N scaling, truncation. */
int sa_process_q15(sa_handle *h, const int16_t *in, int16_t *out_iq, int batch, void *stream);

/* Window (+ integer IIR) only: the FFT input stream, [B,16384] int16 (fft_in16 of
 * new/command_control.vhd:90-123). */
int sa_filter_q15(sa_handle *h, const int16_t *in, int16_t *out_time, int batch, void *stream);

/* float path: in [B,16384] float32 device -> out per out_kind (SA_OUT_*), device.
 * Accuracy against the float64 oracle |rfft(sosfilt(sos, x*window))| (scipy / numpy), in ONE norm throughout: the
 * max-norm error of the magnitude spectrum of a frame relative to that spectrum's peak.  Within 1e-5 wherever
 * float32 arithmetic itself allows it: every fixture, the headline 12th-order Butterworth (worst of 4096 frames
 * 1.9e-6), 4412 of 4500 random designs (tests/fuzz_parity.py, seeds 7 / 11 / 23 x 1500, round 4).  For filters whose
 * poles sit next to the unit circle, or whose output is stop-band leakage far below the input, float32 runs out: a
 * SEQUENTIAL float32 evaluation of sosfilt's own recurrence is above 1e-5 in this norm for 237 of the same 4500
 * designs, this path for 88.  Of those 88, 86 are within 4x the sequential figure (83 within 2x); the two others
 * measure 4.0x (8.2e-5) and 15.6x (2.4e-4: a 12th-order Butterworth band-pass with double zeros at z = 1 and poles of
 * radius 0.984 eight degrees away from them -- a float32 recursion restarted from EXACT chunk start states gives the same
 * 2.4e-4, so the cost is that of evaluating in chunks at all, not of the predictor; tools/accuracy_study.py).  The GPU
 * tests enforce max(1e-5, 4x sequential float32) on the first 160 designs of seed 7 and pin three of the outliers by
 * name with the bounds they meet (tests/test_gpu_f32.py::test_random_designs).  Callers that need 1e-5 on such
 * designs need float64, which this path does not offer. */
int sa_process_f32(sa_handle *h, const float *in, void *out, int batch, int out_kind, void *stream);

/* The float path fed with the ADC's samples (build extension): in [B,16384] int16 device -- the board delivers
 * 12-bit samples sign-extended to int16 (imp/dsp_system_top.vhd:435) and that is what the ingest front-end moves over
 * PCIe.  x = (float)sample * scale is rounded once in the stage-in (scale = 1/2048 maps the ADC range to [-1, 1)) and
 * then takes exactly the float32 path: the results are those of sa_process_f32() on the converted frames, bit for bit,
 * without the conversion pass and with half the input bytes (32 KiB per frame, one fetch round instead of two). */
int sa_process_f32_i16(sa_handle *h, const int16_t *in, float scale, void *out, int batch, int out_kind, void *stream);

/* Host helper: view of one frame as the byte stream sequ2 emits.  On little-endian hosts the
 * Q15 output already is that stream; this copies 65536 bytes and is provided for symmetry with
 * gui.py:250-260 (decode side). */
int sa_pack_frame(const int16_t *iq_host /* [16384,2] */, uint8_t *frame_bytes /* [65536] */);

/* Introspection for tests / tuning: the float IIR plan the kernels consume (chunked-scan form of
 * the cascade: per section 5 taps, predictor taps, state-transition powers).  Writes at most `cap`
 * floats, returns the count needed (or a negative error). */
int sa_debug_iir_plan_f32(const sa_handle *h, float *out, int cap);

/* Same plan computed on the host from an SOS (scipy row order, a0-normalised here), without a
 * handle or a GPU: pure host logic, used by the CPU tests to check the chunked-scan algebra. */
int sa_iir_plan_from_sos(const double *sos, int n_sections, float *out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* SPECAN_H_ */
