// Context, error channel and raw device-memory plumbing of the C ABI.
#include "common.h"

static thread_local char g_err[1024] = "";

void impdar_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *impdar_last_error(void) { return g_err; }

extern "C" int impdar_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int impdar_ctx_create(int device, impdar_ctx **out)
{
    IMPDAR_ARG_CHECK(out, "null output pointer");
    int n = impdar_device_count();
    if (n <= 0) {
        impdar_set_error("no HIP device visible (hipGetDeviceCount = %d)", n);
        return IMPDAR_ERR_NODEV;
    }
    IMPDAR_ARG_CHECK(device >= 0 && device < n, "device %d out of range [0,%d)", device, n);
    IMPDAR_HIP_CHECK(hipSetDevice(device));
    impdar_ctx *c = new impdar_ctx();
    c->device = device;
    // compute stream at the highest priority, producer stream at the lowest: the next
    // radargram's prep then fills the tail of the current diffraction sum instead of
    // competing with it for CUs
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    hipError_t e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_hi);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->aux, hipStreamNonBlocking, prio_lo);
    if (e != hipSuccess) {
        delete c;
        impdar_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        return IMPDAR_ERR_HIP;
    }
    *out = c;
    return IMPDAR_OK;
}

void impdar_comm_destroy(impdar_ctx *ctx);   // comm.hip

extern "C" void impdar_ctx_destroy(impdar_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->aux);
    (void)hipStreamSynchronize(ctx->stream);
    impdar_comm_destroy(ctx);
    (void)hipStreamDestroy(ctx->aux);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int impdar_ctx_sync(impdar_ctx *ctx)
{
    IMPDAR_ARG_CHECK(ctx, "null context");
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    IMPDAR_HIP_CHECK(hipStreamSynchronize(ctx->aux));
    IMPDAR_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return IMPDAR_OK;
}

extern "C" int impdar_dev_alloc(impdar_ctx *ctx, size_t bytes, void **dptr)
{
    IMPDAR_ARG_CHECK(ctx && dptr, "null context/pointer");
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    IMPDAR_HIP_CHECK(hipMalloc(dptr, bytes ? bytes : 8));
    return IMPDAR_OK;
}

extern "C" int impdar_dev_free(impdar_ctx *ctx, void *dptr)
{
    IMPDAR_ARG_CHECK(ctx, "null context");
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    IMPDAR_HIP_CHECK(hipStreamSynchronize(ctx->aux));
    IMPDAR_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    IMPDAR_HIP_CHECK(hipFree(dptr));
    return IMPDAR_OK;
}

extern "C" int impdar_dev_upload(impdar_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes)
{
    IMPDAR_ARG_CHECK(ctx && dst_dev && src_host, "null context/pointer");
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    IMPDAR_HIP_CHECK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    IMPDAR_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return IMPDAR_OK;
}

extern "C" int impdar_dev_download(impdar_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes)
{
    IMPDAR_ARG_CHECK(ctx && dst_host && src_dev, "null context/pointer");
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    IMPDAR_HIP_CHECK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    IMPDAR_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return IMPDAR_OK;
}

extern "C" int impdar_dev_memset(impdar_ctx *ctx, void *dst_dev, int value, size_t bytes)
{
    IMPDAR_ARG_CHECK(ctx && dst_dev, "null context/pointer");
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    IMPDAR_HIP_CHECK(hipMemsetAsync(dst_dev, value, bytes, ctx->stream));
    return IMPDAR_OK;
}
