// Shared host-side plumbing for the impdar HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <string>
#include <vector>
#include "../../include/impdar_hip.h"

struct ncclComm;

struct impdar_ctx {
    int device = 0;
    hipStream_t stream = nullptr;   // compute stream (migration kernels, copies)
    hipStream_t aux = nullptr;      // producer stream: prep / table / all-gather of the NEXT radargram
    // RCCL communicator (null until impdar_comm_init)
    ncclComm *comm = nullptr;
    int rank = 0;
    int nranks = 1;
    // pinned host staging for the one-shot (host-buffer) entry points: device -> pinned at PCIe speed,
    // then pinned -> the caller's pageable buffer on several threads (first touch of a fresh result
    // array is a page fault per 4 KiB; the faults parallelise, a single D2H into it does not)
    void *pinned = nullptr;
    size_t pinned_bytes = 0;
    std::mutex pinned_mu;           // held for a whole staged download (the buffer may be re-allocated by the next one)
    // a staging buffer on its way: allocated by a thread of its own while the call that will need it plans, uploads and
    // computes (hipHostMalloc pins ~4.5 GB/s: 36 ms for the 164 MB of a config-3 image -- the largest term of a first call)
    std::thread pin_thread;
    void *pin_next = nullptr;
    size_t pin_next_bytes = 0;
    // last work enqueued on `stream` by a *_dev entry point that WRITES a caller-visible device array; consumers
    // on the producer stream (impdar_kirch_prep) wait for it
    hipEvent_t ev_produced = nullptr;
    bool produced = false;
    // device-side duration of the last Stolt / phase-shift call enqueued on `stream` (impdar_ctx_last_ms)
    hipEvent_t ev_tic = nullptr, ev_toc = nullptr;
    bool timed = false;
    // ... and of its dominant kernel alone (the frequency sum of a phase-shift call: impdar_ctx_last_kernel_ms)
    hipEvent_t ev_ktic = nullptr, ev_ktoc = nullptr;
    bool ktimed = false;
    // what the last migration entry point on this context did (impdar_ctx_last_metrics): its name, the kernel that did
    // the sums, that kernel's time when the entry point measured it itself (< 0: read ev_ktic / ev_ktoc), and further
    // "key": value pairs of JSON
    const char *m_entry = nullptr, *m_kernel = nullptr;
    float m_kernel_ms = -1.f;
    char m_extra[320] = "";
};

// the staging ring of impdar_download_blocks_f64
constexpr size_t IMPDAR_STAGE_RING_BYTES = (size_t)64 << 20;
// start allocating a pinned staging buffer of `bytes` unless the context has one (impdar_ctx_pinned adopts it)
void impdar_ctx_pinned_prefetch(impdar_ctx *ctx, size_t bytes);
// bracket the device work of one call on ctx->stream (read back by impdar_ctx_last_ms)
int impdar_ctx_tic(impdar_ctx *ctx);
int impdar_ctx_toc(impdar_ctx *ctx);
// bracket the dominant kernel(s) inside such a call (read back by impdar_ctx_last_kernel_ms)
int impdar_ctx_ktic(impdar_ctx *ctx);
int impdar_ctx_ktoc(impdar_ctx *ctx);

// record "everything enqueued on ctx->stream so far produces data a later call may read" (see impdar_kirch_prep)
int impdar_ctx_mark_produced(impdar_ctx *ctx);

// grow-only pinned staging buffer of the context; nullptr when the allocation fails (callers fall back to a
// direct pageable copy)
void *impdar_ctx_pinned(impdar_ctx *ctx, size_t bytes);
// copy (or convert float -> double) `n` elements from pinned staging into pageable memory on a few threads
void impdar_host_copy_f64(double *dst, const void *src, size_t n, bool src_is_f32);
// device -> host through the pinned staging buffer and a threaded copy (falls back to a plain D2H); synchronises `st`
int impdar_download(impdar_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes, hipStream_t st);
// a (rows x width) device block of `dtype` into columns [col0, col0 + width) of a (rows x ld) float64 host array, on `st`
int impdar_download_block_f64(impdar_ctx *ctx, double *dst_host, size_t ld, size_t col0, const void *src_dev, int dtype,
                              size_t rows, size_t width, hipStream_t st);

// several output-trace blocks of one image, each after its own event, copies and host widening pipelined across blocks
int impdar_download_blocks_f64(impdar_ctx *ctx, double *dst_host, size_t ld, size_t rows, int dtype, int nblk,
                               const size_t *col0, const size_t *width, const void *const *src, const hipEvent_t *after,
                               hipStream_t st);

// device arrays of the one-shot calls, from a small cache of freed ones (api.hip); the caller of _free has synchronised their users
int impdar_devcache_alloc(int device, size_t bytes, void **dptr);
void impdar_devcache_free(void *p);

void impdar_set_error(const char *fmt, ...);
// IMPDAR_TRACE=1: one line on stderr per milestone of a call ("[impdar +12.3 ms] stolt: plans ready"), the time since the
// library's first trace point -- where a first call's time goes (profiles/r05_first_call.txt).  A no-op otherwise.
bool impdar_trace_on();
void impdar_trace(const char *fmt, ...);

#define IMPDAR_HIP_CHECK(expr)                                                        \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            impdar_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,            \
                             hipGetErrorString(_e));                                  \
            return IMPDAR_ERR_HIP;                                                    \
        }                                                                             \
    } while (0)

#define IMPDAR_ARG_CHECK(cond, ...)                                                   \
    do {                                                                              \
        if (!(cond)) {                                                                \
            impdar_set_error(__VA_ARGS__);                                            \
            return IMPDAR_ERR_ARG;                                                    \
        }                                                                             \
    } while (0)

static inline size_t impdar_dtype_size(int dtype) { return dtype == IMPDAR_F64 ? 8 : 4; }

// Drop what the entry points keep between calls -- the one-shot Kirchhoff plan with its images and staging copies
// (hundreds of MB at config 3), the mig_kirch_loop plan, the Stolt and phase-shift plans with their spectra (GBs at
// 8192^2) -- except what a call in progress on this thread's stack is using (their mutexes are only tried).  Called by
// DevBuf::ensure when hipMalloc reports out of memory, before it tries once more: a large call after a large call
// of another kind must not fail on memory that only a cache is holding.
void impdar_release_caches();
// "this thread is inside the entry point that owns cache X": its trim must not even try the (non-recursive) mutex
struct ImpdarBusy {
    bool &flag;
    explicit ImpdarBusy(bool &f) : flag(f) { flag = true; }
    ~ImpdarBusy() { flag = false; }
};

// RAII device buffer bound to a context's device.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    hipError_t ensure(size_t n) {
        if (n <= bytes) return hipSuccess;
        release();
        hipError_t e = hipMalloc(&p, n);
        if (e == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            p = nullptr;
            // the trims destroy cached plans of other contexts and set THEIR device while doing so: come back to ours
            int dev = -1;
            const bool have_dev = hipGetDevice(&dev) == hipSuccess;
            impdar_release_caches();
            if (have_dev) (void)hipSetDevice(dev);
            e = hipMalloc(&p, n);
        }
        if (e == hipSuccess) bytes = n;
        else p = nullptr;
        return e;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};
