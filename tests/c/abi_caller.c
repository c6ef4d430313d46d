/* A plain C caller of the C ABI (include/specan.h): what INTEGRATION.md section 3 shows, compiled for real.
 *
 *   abi_caller MODE IN.bin OUT.bin [N_FRAMES] [COEFFS.bin]
 *
 * MODE is the filter-select byte (0x00 / 0xA1 / 0xB1); IN.bin holds N_FRAMES x 16384 int16 samples (host
 * order), COEFFS.bin 12 int8 coefficients for mode 0xA1, sent through the byte-stream front door exactly as the
 * UART would deliver them (0xF1 + 12 bytes + the mode byte).  OUT.bin receives N_FRAMES x 65536 frame bytes
 * through sa_pack_frame.  Exit code 0 on success, 3 when no GPU is usable (sa_create says so), 1 otherwise.
 * Built by tests/test_c_caller.py with gcc against libspecan_hip.so and libamdhip64.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "specan.h"

#define CHECK_SA(call)                                                              \
    do {                                                                            \
        int rc_ = (call);                                                           \
        if (rc_ != SA_OK) {                                                         \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, sa_last_error(h));        \
            return 1;                                                               \
        }                                                                           \
    } while (0)
#define CHECK_HIP(call)                                                             \
    do {                                                                            \
        hipError_t e_ = (call);                                                     \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_));            \
            return 1;                                                               \
        }                                                                           \
    } while (0)

int main(int argc, char **argv)
{
    sa_handle *h = NULL;
    if (sa_abi_version() != SA_ABI_VERSION) {
        fprintf(stderr, "header / library ABI mismatch\n");
        return 1;
    }
    if (argc < 4) {
        /* no arguments: the loud-failure probe used on a box without a GPU */
        int rc = sa_create(0, &h);
        if (rc != SA_OK) {
            fprintf(stderr, "sa_create -> %d: %s\n", rc, sa_last_error(NULL));
            return rc == SA_EHIP ? 3 : 1;
        }
        sa_destroy(h);
        return 0;
    }
    const unsigned mode = (unsigned)strtoul(argv[1], NULL, 0);
    const int nframes = argc > 4 ? atoi(argv[4]) : 1;
    const size_t n_in = (size_t)nframes * SA_N, n_out = (size_t)nframes * SA_N * 2;
    int16_t *x = (int16_t *)malloc(n_in * sizeof(int16_t)), *iq = (int16_t *)malloc(n_out * sizeof(int16_t));
    uint8_t *frame = (uint8_t *)malloc(SA_FRAME_BYTES);
    FILE *fi = fopen(argv[2], "rb");
    if (!x || !iq || !frame || !fi || fread(x, sizeof(int16_t), n_in, fi) != n_in) {
        fprintf(stderr, "cannot read %zu samples from %s\n", n_in, argv[2]);
        return 1;
    }
    fclose(fi);

    int rc = sa_create(0, &h);
    if (rc != SA_OK) {
        fprintf(stderr, "sa_create -> %d: %s\n", rc, sa_last_error(NULL));
        return rc == SA_EHIP ? 3 : 1;
    }
    /* control plane through the UART byte stream */
    uint8_t cmd[16];
    size_t ncmd = 0;
    if (argc > 5) {
        FILE *fc = fopen(argv[5], "rb");
        cmd[ncmd++] = SA_CMD_FILTER_UPDATE;
        if (!fc || fread(cmd + ncmd, 1, 12, fc) != 12) {
            fprintf(stderr, "cannot read 12 coefficients from %s\n", argv[5]);
            return 1;
        }
        fclose(fc);
        ncmd += 12;
    }
    cmd[ncmd++] = (uint8_t)mode;
    sa_cmd_events ev;
    memset(&ev, 0, sizeof ev);
    CHECK_SA(sa_feed_command_bytes_ex(h, cmd, ncmd, &ev));
    uint8_t got_mode = 0;
    CHECK_SA(sa_get_filter_mode(h, &got_mode));
    if (got_mode != mode || ev.n_uploads != (argc > 5 ? 1 : 0) || ev.transport != SA_CMD_ETHERNET_MODE) {
        fprintf(stderr, "command bytes not decoded as expected\n");
        return 1;
    }

    hipStream_t stream;
    int16_t *d_in = NULL, *d_out = NULL;
    CHECK_HIP(hipStreamCreate(&stream));
    CHECK_HIP(hipMalloc((void **)&d_in, n_in * sizeof(int16_t)));
    CHECK_HIP(hipMalloc((void **)&d_out, n_out * sizeof(int16_t)));
    CHECK_HIP(hipMemcpyAsync(d_in, x, n_in * sizeof(int16_t), hipMemcpyHostToDevice, stream));
    CHECK_SA(sa_reserve(h, nframes));
    CHECK_SA(sa_process_q15(h, d_in, d_out, nframes, (void *)stream));
    CHECK_HIP(hipMemcpyAsync(iq, d_out, n_out * sizeof(int16_t), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));

    FILE *fo = fopen(argv[3], "wb");
    if (!fo) return 1;
    for (int f = 0; f < nframes; ++f) {
        CHECK_SA(sa_pack_frame(iq + (size_t)f * SA_N * 2, frame));
        if (fwrite(frame, 1, SA_FRAME_BYTES, fo) != SA_FRAME_BYTES) return 1;
    }
    fclose(fo);
    /* argument errors come back as codes, never as aborts */
    if (sa_process_q15(h, NULL, d_out, 1, (void *)stream) != SA_EINVAL || sa_set_filter_mode(h, 0x42) != SA_EINVAL) {
        fprintf(stderr, "bad arguments were not rejected\n");
        return 1;
    }
    /* Stream lifetime: the library may use a stream only inside the call that passed it.  Destroy stream A right
     * after its work, run the same batch on a new stream B, then upload a table (default ROM again) and run once
     * more: every result equals the first one, and sa_destroy() comes after the last stream is gone. */
    {
        hipStream_t sb;
        int16_t *iq2 = (int16_t *)malloc(n_out * sizeof(int16_t));
        CHECK_HIP(hipStreamDestroy(stream));
        CHECK_HIP(hipStreamCreate(&sb));
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) CHECK_SA(sa_set_window_q15(h, NULL));
            CHECK_HIP(hipMemsetAsync(d_out, 0, n_out * sizeof(int16_t), sb));
            CHECK_SA(sa_process_q15(h, d_in, d_out, nframes, (void *)sb));
            CHECK_HIP(hipMemcpyAsync(iq2, d_out, n_out * sizeof(int16_t), hipMemcpyDeviceToHost, sb));
            CHECK_HIP(hipStreamSynchronize(sb));
            if (memcmp(iq, iq2, n_out * sizeof(int16_t)) != 0) {
                fprintf(stderr, "results changed after the first stream was destroyed (pass %d)\n", pass);
                return 1;
            }
        }
        CHECK_HIP(hipStreamDestroy(sb));
        free(iq2);
    }
    hipFree(d_in);
    hipFree(d_out);
    sa_destroy(h);
    free(x); free(iq); free(frame);
    return 0;
}
