/* Host-side driver for the sanitizer build of libimpdar_hip (tests/test_sanitizer.py).  Runs on the CPU
 * container only (no GPU, never on the GPU box): walks the argument-error and no-device paths of the C ABI
 * (include/impdar_hip.h) so that AddressSanitizer / UBSan see the host code of the shim.  Exit code 0 = every
 * call returned the status expected here; the sanitizers abort the process on a finding. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/impdar_hip.h"

static int failures = 0;
#define EXPECT(cond, what)                                                                   \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            fprintf(stderr, "san_driver: %s (last error: %s)\n", what, impdar_last_error()); \
            ++failures;                                                                      \
        }                                                                                    \
    } while (0)

int main(void)
{
    impdar_ctx *ctx = NULL;
    int ndev = impdar_device_count();
    int rc = impdar_ctx_create(0, &ctx);
    if (ndev > 0 && rc == IMPDAR_OK) {
        fprintf(stderr, "san_driver: a GPU is visible; this driver is for the CPU container\n");
        impdar_ctx_destroy(ctx);
        return 77;
    }
    EXPECT(rc == IMPDAR_ERR_NODEV && strlen(impdar_last_error()) > 0, "ctx_create without a device must be ERR_NODEV");
    EXPECT(impdar_ctx_create(0, NULL) == IMPDAR_ERR_ARG, "null out pointer");
    EXPECT(impdar_ctx_sync(NULL) == IMPDAR_ERR_ARG, "ctx_sync(NULL)");
    impdar_ctx_destroy(NULL);

    /* a stand-in context: opaque to callers, the library only reads its device index before the first HIP call */
    void *fake = calloc(1, 4096);
    impdar_ctx *fctx = (impdar_ctx *)fake;
    enum { SNUM = 64, TNUM = 40 };
    double tt[SNUM], dist[TNUM], bad_tt[SNUM], jit[TNUM], zig[TNUM];
    for (int k = 0; k < SNUM; ++k) { tt[k] = k * 1e-8; bad_tt[k] = (k == 7 ? 3 : k) * 1e-8; }
    for (int j = 0; j < TNUM; ++j) { dist[j] = j * 1.0; jit[j] = j * 1.0 + (j % 3) * 0.2; zig[j] = j * 1.0 + (j % 2) * 2.5; }
    impdar_kirch_plan *plan = NULL;
    /* argument checks */
    EXPECT(impdar_kirch_plan_create(NULL, IMPDAR_F32, SNUM, TNUM, dist, tt, 1.69e8, 0, 1, 1e-8, 0, 0, 0, 0, 1, &plan) == IMPDAR_ERR_ARG, "null ctx");
    EXPECT(impdar_kirch_plan_create(fctx, 7, SNUM, TNUM, dist, tt, 1.69e8, 0, 1, 1e-8, 0, 0, 0, 0, 1, &plan) == IMPDAR_ERR_ARG, "bad dtype");
    EXPECT(impdar_kirch_plan_create(fctx, IMPDAR_F32, 1, TNUM, dist, tt, 1.69e8, 0, 1, 1e-8, 0, 0, 0, 0, 1, &plan) == IMPDAR_ERR_ARG, "snum 1");
    EXPECT(impdar_kirch_plan_create(fctx, IMPDAR_F32, SNUM, 0, dist, tt, 1.69e8, 0, 1, 1e-8, 0, 0, 0, 0, 1, &plan) == IMPDAR_ERR_ARG, "tnum 0");
    EXPECT(impdar_kirch_plan_create(fctx, IMPDAR_F32, SNUM, TNUM, NULL, tt, 1.69e8, 0, 1, 1e-8, 0, 0, 0, 0, 1, &plan) == IMPDAR_ERR_ARG, "null dist");
    EXPECT(impdar_kirch_plan_create(fctx, IMPDAR_F32, SNUM, TNUM, dist, tt, -1.0, 0, 1, 1e-8, 0, 0, 0, 0, 1, &plan) == IMPDAR_ERR_ARG, "vel < 0");
    EXPECT(impdar_kirch_plan_create(fctx, IMPDAR_F32, SNUM, TNUM, dist, tt, 1.69e8, 0, 1, 1e-8, 0, 0, 0, 0, 0, &plan) == IMPDAR_ERR_ARG, "nranks 0");
    EXPECT(impdar_kirch_plan_create(fctx, IMPDAR_F32, SNUM, TNUM, dist, tt, 1.69e8, 0, 0, 1e-8, 0, 0, 0, 0, 1, &plan) == IMPDAR_ERR_ARG, "non-uniform gradient without coefficients");
    /* host geometry analysis (runs before the first HIP call) */
    EXPECT(impdar_kirch_plan_create(fctx, IMPDAR_F32, SNUM, TNUM, dist, bad_tt, 1.69e8, 0, 1, 1e-8, 0, 0, 0, 0, 1, &plan) == IMPDAR_ERR_ARG
           && strstr(impdar_last_error(), "increasing"), "non-monotonic travel_time");
    /* (round 4: float32 data on a sorted non-uniform spacing has a fast kernel of its own, kirch_gen_kernel -- the plan
       gets as far as the device; a spacing that is not sorted has none) */
    EXPECT(impdar_kirch_plan_create(fctx, IMPDAR_F32, SNUM, TNUM, jit, tt, 1.69e8, 0, 1, 1e-8, 0, 0, 0, IMPDAR_KIRCH_FAST, 1, &plan) == IMPDAR_ERR_HIP,
           "fast kernel on a sorted non-uniform trace spacing: as far as the first HIP call");
    EXPECT(impdar_kirch_plan_create(fctx, IMPDAR_F32, SNUM, TNUM, zig, tt, 1.69e8, 0, 1, 1e-8, 0, 0, 0, IMPDAR_KIRCH_FAST, 1, &plan) == IMPDAR_ERR_UNSUPPORTED,
           "fast kernel on an unsorted trace spacing");
    EXPECT(impdar_kirch_plan_create(fctx, IMPDAR_F64, SNUM, TNUM, dist, tt, 1.69e8, 0, 1, 1e-8, 0, 0, 0, IMPDAR_KIRCH_FAST, 1, &plan) == IMPDAR_ERR_UNSUPPORTED,
           "fast kernel on float64 data");
    /* a valid plan gets as far as the device and fails there, releasing what it built */
    for (int mode = 0; mode <= 2; ++mode)
        for (int nranks = 1; nranks <= 8; nranks += 7)
            EXPECT(impdar_kirch_plan_create(fctx, IMPDAR_F32, SNUM, TNUM, dist, tt, 1.69e8, mode & 1, 1, 1e-8, 0, 0, 0, mode, nranks, &plan) == IMPDAR_ERR_HIP,
                   "valid geometry must fail at the first HIP call without a device");
    EXPECT(impdar_kirch_count_pairs(NULL, 0, 1) == -1, "count_pairs(NULL)");
    EXPECT(impdar_kirch_plan_mode(NULL) == IMPDAR_ERR_ARG && impdar_kirch_plan_tnum_pad(NULL) == IMPDAR_ERR_ARG, "plan getters on NULL");
    impdar_kirch_plan_destroy(NULL);
    EXPECT(impdar_kirch_prep(NULL, tt, 1, 0, 1) == IMPDAR_ERR_ARG, "prep(NULL)");
    EXPECT(impdar_kirch_migrate(NULL, tt, 0, 1) == IMPDAR_ERR_ARG, "migrate(NULL)");
    EXPECT(impdar_kirch_allgather(NULL) == IMPDAR_ERR_ARG, "allgather(NULL)");
    EXPECT(impdar_kirch_exchange(NULL, 0, 0, 0, 0, 0, 0, 0, 0) == IMPDAR_ERR_ARG, "exchange(NULL)");
    float ms;
    EXPECT(impdar_kirch_history_ms(NULL, 0, &ms, &ms, &ms) == IMPDAR_ERR_ARG, "history(NULL)");
    EXPECT(impdar_ctx_last_ms(NULL, &ms) == IMPDAR_ERR_ARG, "last_ms(NULL)");
    /* the other entry points: null / shape errors are reported before any device work */
    double out[4];
    EXPECT(impdar_kirchhoff(NULL, tt, IMPDAR_F64, SNUM, TNUM, dist, tt, 1.69e8, 0, 1, 1e-8, 0, 0, 0, 0, out) == IMPDAR_ERR_ARG, "kirchhoff(NULL)");
    EXPECT(impdar_stolt(NULL, tt, IMPDAR_F64, SNUM, TNUM, dist, tt, 1.68e8, 10, 10, out) == IMPDAR_ERR_ARG, "stolt(NULL)");
    EXPECT(impdar_stolt_dev(fctx, tt, 9, SNUM, TNUM, dist, tt, 1.68e8, 10, 10, out) == IMPDAR_ERR_ARG, "stolt bad dtype");
    EXPECT(impdar_stolt_dev(fctx, tt, IMPDAR_F32, 1, TNUM, dist, tt, 1.68e8, 10, 10, out) == IMPDAR_ERR_ARG, "stolt snum 1");
    EXPECT(impdar_phaseshift_dev(fctx, tt, IMPDAR_F32, SNUM, TNUM, 32, dist, tt, 1e-8, tt, 1.69e8, NULL, 0, 10, 10, out) == IMPDAR_ERR_ARG, "phase shift nt < snum");
    EXPECT(impdar_phaseshift_dev(fctx, tt, IMPDAR_F32, SNUM, TNUM, 64, dist, tt, 1e-8, tt, 1.69e8, tt, 5, 10, 10, out) == IMPDAR_ERR_ARG
           && strstr(impdar_last_error(), "velocity profile"), "phase shift velocity profile of the wrong length");
    EXPECT(impdar_filtfilt_dev(fctx, out, IMPDAR_F64, 10, 4, tt, tt + 1, 5, tt) == IMPDAR_ERR_ARG && strstr(impdar_last_error(), "padlen"),
           "filtfilt on a trace shorter than the padding");
    EXPECT(impdar_filtfilt_dev(fctx, out, IMPDAR_F64, 100, 4, tt, tt, 5, tt) == IMPDAR_ERR_ARG, "filtfilt a[0] == 0");
    EXPECT(impdar_fir_shift_dev(fctx, out, IMPDAR_F64, 100, 4, tt, 0) == IMPDAR_ERR_ARG, "fir with no taps");
    int lo[2] = {0, 99}, hi[2] = {1, 2};
    EXPECT(impdar_trace_lerp_dev(fctx, tt, IMPDAR_F64, 8, 4, lo, hi, tt, tt, 2, out) == IMPDAR_ERR_ARG, "lerp column out of range");
    EXPECT(impdar_cast_dev(fctx, tt, 5, out, IMPDAR_F64, 4) == IMPDAR_ERR_ARG, "cast bad dtype");
    EXPECT(impdar_comm_init(NULL, (const char *)tt, 0, 1) == IMPDAR_ERR_ARG, "comm_init(NULL)");
    EXPECT(impdar_comm_init(fctx, (const char *)tt, 3, 2) == IMPDAR_ERR_ARG, "rank outside the communicator");
    EXPECT(impdar_comm_rank(NULL) == IMPDAR_ERR_ARG && impdar_comm_size(NULL) == IMPDAR_ERR_ARG, "comm getters on NULL");
    EXPECT(impdar_dev_alloc(NULL, 16, (void **)&fake) == IMPDAR_ERR_ARG, "dev_alloc(NULL)");
    free(fctx);
    if (failures) return 1;
    printf("san_driver ok\n");
    return 0;
}
