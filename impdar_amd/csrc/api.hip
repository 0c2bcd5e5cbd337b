// Context, error channel and raw device-memory plumbing of the C ABI.
#include "common.h"
#include <algorithm>
#include <thread>
#include <unistd.h>
#include <condition_variable>
#include <functional>

static thread_local char g_err[1024] = "";

void impdar_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *impdar_last_error(void) { return g_err; }

#include <chrono>
bool impdar_trace_on()
{
    const char *e = getenv("IMPDAR_TRACE");          // read per call (a handful of trace points per migration)
    return e && atoi(e) != 0;
}
void impdar_trace(const char *fmt, ...)
{
    if (!impdar_trace_on()) return;
    static const auto t0 = std::chrono::steady_clock::now();
    static std::mutex mu;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    const double unix_s = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
    std::lock_guard<std::mutex> lk(mu);
    fprintf(stderr, "[impdar +%9.2f ms, unix %.3f] %s\n", ms, unix_s, buf);
}

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
    impdar_trace("ctx_create: enter");
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
    impdar_trace("ctx_create: device %d set, two streams made", device);
    return IMPDAR_OK;
}

int impdar_ctx_mark_produced(impdar_ctx *ctx)
{
    if (!ctx->ev_produced) IMPDAR_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_produced, hipEventDisableTiming));
    IMPDAR_HIP_CHECK(hipEventRecord(ctx->ev_produced, ctx->stream));
    ctx->produced = true;
    return IMPDAR_OK;
}

int impdar_ctx_tic(impdar_ctx *ctx)
{
    if (!ctx->ev_tic) IMPDAR_HIP_CHECK(hipEventCreate(&ctx->ev_tic));
    if (!ctx->ev_toc) IMPDAR_HIP_CHECK(hipEventCreate(&ctx->ev_toc));
    ctx->timed = false;
    ctx->ktimed = false;
    IMPDAR_HIP_CHECK(hipEventRecord(ctx->ev_tic, ctx->stream));
    return IMPDAR_OK;
}

int impdar_ctx_ktic(impdar_ctx *ctx)
{
    if (!ctx->ev_ktic) IMPDAR_HIP_CHECK(hipEventCreate(&ctx->ev_ktic));
    if (!ctx->ev_ktoc) IMPDAR_HIP_CHECK(hipEventCreate(&ctx->ev_ktoc));
    ctx->ktimed = false;
    IMPDAR_HIP_CHECK(hipEventRecord(ctx->ev_ktic, ctx->stream));
    return IMPDAR_OK;
}

int impdar_ctx_ktoc(impdar_ctx *ctx)
{
    IMPDAR_HIP_CHECK(hipEventRecord(ctx->ev_ktoc, ctx->stream));
    ctx->ktimed = true;
    return IMPDAR_OK;
}

int impdar_ctx_toc(impdar_ctx *ctx)
{
    IMPDAR_HIP_CHECK(hipEventRecord(ctx->ev_toc, ctx->stream));
    ctx->timed = true;
    return IMPDAR_OK;
}

extern "C" int impdar_ctx_last_ms(impdar_ctx *ctx, float *ms)
{
    IMPDAR_ARG_CHECK(ctx && ms, "null context/pointer");
    IMPDAR_ARG_CHECK(ctx->timed, "no timed call (impdar_stolt_dev / impdar_phaseshift_dev) has run on this context");
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    IMPDAR_HIP_CHECK(hipEventSynchronize(ctx->ev_toc));
    IMPDAR_HIP_CHECK(hipEventElapsedTime(ms, ctx->ev_tic, ctx->ev_toc));
    return IMPDAR_OK;
}

extern "C" int impdar_ctx_last_kernel_ms(impdar_ctx *ctx, float *ms)
{
    IMPDAR_ARG_CHECK(ctx && ms, "null context/pointer");
    IMPDAR_ARG_CHECK(ctx->ktimed, "the last timed call on this context bracketed no kernel (only impdar_phaseshift[_dev] does)");
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    IMPDAR_HIP_CHECK(hipEventSynchronize(ctx->ev_ktoc));
    IMPDAR_HIP_CHECK(hipEventElapsedTime(ms, ctx->ev_ktic, ctx->ev_ktoc));
    return IMPDAR_OK;
}

// One JSON object about the last migration entry point that ran on this context ("emit a JSON
// line per run beside the print"; the reference only prints 'complete in N seconds', mig_python.py:121-122,206-207,
// 285-286).  The Python entry points add the sizes, traces per second and the device count and print the line on
// stderr when IMPDAR_METRICS is set.
extern "C" int impdar_ctx_last_metrics(impdar_ctx *ctx, char *json, size_t cap)
{
    IMPDAR_ARG_CHECK(ctx && json && cap > 0, "null context/buffer");
    IMPDAR_ARG_CHECK(ctx->m_entry, "no migration entry point has run on this context");
    float dev_ms = -1.f, k_ms = ctx->m_kernel_ms;
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    if (ctx->timed && hipEventSynchronize(ctx->ev_toc) == hipSuccess) (void)hipEventElapsedTime(&dev_ms, ctx->ev_tic, ctx->ev_toc);
    if (k_ms < 0.f && ctx->ktimed && hipEventSynchronize(ctx->ev_ktoc) == hipSuccess)
        (void)hipEventElapsedTime(&k_ms, ctx->ev_ktic, ctx->ev_ktoc);
    (void)hipGetLastError();
    int n = snprintf(json, cap, "{\"entry\": \"%s\", \"kernel\": \"%s\", \"device\": %d", ctx->m_entry,
                     ctx->m_kernel ? ctx->m_kernel : "", ctx->device);
    if (n > 0 && (size_t)n < cap && k_ms >= 0.f) n += snprintf(json + n, cap - n, ", \"kernel_ms\": %.4f", k_ms);
    if (n > 0 && (size_t)n < cap && dev_ms >= 0.f) n += snprintf(json + n, cap - n, ", \"device_ms\": %.4f", dev_ms);
    if (n > 0 && (size_t)n < cap && ctx->m_extra[0]) n += snprintf(json + n, cap - n, ", %s", ctx->m_extra);
    if (n > 0 && (size_t)n < cap) n += snprintf(json + n, cap - n, "}");
    IMPDAR_ARG_CHECK(n > 0 && (size_t)n < cap, "buffer of %zu bytes is too small for the metrics object", cap);
    return IMPDAR_OK;
}

void impdar_comm_destroy(impdar_ctx *ctx);   // comm.hip

void impdar_stolt_forget(const impdar_ctx *ctx);   // stolt.hip
void impdar_ps_forget(const impdar_ctx *ctx);      // phaseshift.hip
void impdar_kirch_forget(const impdar_ctx *ctx);   // kirchhoff.hip
void impdar_preproc_forget(impdar_ctx *ctx);       // preproc.hip

void impdar_kirch_trim();    // kirchhoff.hip
void impdar_stolt_trim();    // stolt.hip
void impdar_ps_trim();       // phaseshift.hip

void impdar_devcache_trim(int device);

void impdar_release_caches()
{
    impdar_kirch_trim();
    impdar_stolt_trim();
    impdar_ps_trim();
    impdar_devcache_trim(-1);
}

static void pinned_adopt(impdar_ctx *ctx)
{
    if (ctx->pin_thread.joinable()) ctx->pin_thread.join();
    if (ctx->pin_next) {
        if (ctx->pin_next_bytes > ctx->pinned_bytes) {
            if (ctx->pinned) (void)hipHostFree(ctx->pinned);
            ctx->pinned = ctx->pin_next;
            ctx->pinned_bytes = ctx->pin_next_bytes;
        } else {
            (void)hipHostFree(ctx->pin_next);
        }
        ctx->pin_next = nullptr;
        ctx->pin_next_bytes = 0;
    }
}

void impdar_ctx_pinned_prefetch(impdar_ctx *ctx, size_t bytes)
{
    std::lock_guard<std::mutex> lock(ctx->pinned_mu);
    if (bytes < ((size_t)1 << 20) || bytes <= ctx->pinned_bytes || bytes > ((size_t)1 << 30)) return;
    if (ctx->pin_thread.joinable()) {
        if (ctx->pin_next_bytes >= bytes) return;          // one on its way that will do
        pinned_adopt(ctx);
        if (bytes <= ctx->pinned_bytes) return;
    }
    ctx->pin_next_bytes = bytes;
    const int device = ctx->device;
    ctx->pin_thread = std::thread([ctx, device, bytes] {
        void *p = nullptr;
        impdar_trace("pinned prefetch thread: hipHostMalloc of %zu MB: start", bytes >> 20);
        if (hipSetDevice(device) != hipSuccess || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
        }
        impdar_trace("pinned prefetch thread: done");
        ctx->pin_next = p;          // (read after the join)
    });
}

// (callers hold ctx->pinned_mu)
void *impdar_ctx_pinned(impdar_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->pinned_bytes) return ctx->pinned;      // (a larger one on its way is not waited for)
    pinned_adopt(ctx);
    if (bytes <= ctx->pinned_bytes) return ctx->pinned;
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    ctx->pinned = nullptr;
    ctx->pinned_bytes = 0;
    void *p = nullptr;
    impdar_trace("impdar_ctx_pinned: synchronous hipHostMalloc of %zu MB", bytes >> 20);
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    impdar_trace("impdar_ctx_pinned: done");
    ctx->pinned = p;
    ctx->pinned_bytes = bytes;
    return p;
}

// Host worker threads that live as long as the process (round 6): a parallel copy used to start and join 16 threads of its own
// -- 0.3-0.5 ms each time, which is why download pieces below 48 MB did not pay.  The pool is created on first use and never
// destroyed (its threads sleep on a condition variable; a static destructor that joined them would run at exit, in the order
// the runtime tears its own threads down).  One job at a time: callers hold ctx->pinned_mu or are the only caller.
struct ImpdarPool {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::function<void(unsigned)> job;
    unsigned nthr = 0, generation = 0, pending = 0;
    std::mutex user_mu;                                     // one parallel_for at a time
    explicit ImpdarPool(unsigned n) : nthr(n)
    {
        for (unsigned t = 0; t < n; ++t)
            std::thread([this, t] {
                unsigned seen = 0;
                for (;;) {
                    std::function<void(unsigned)> f;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv_work.wait(lk, [&] { return generation != seen; });
                        seen = generation;
                        f = job;
                    }
                    f(t);
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        if (--pending == 0) cv_done.notify_all();
                    }
                }
            }).detach();
    }
    void run(const std::function<void(unsigned)> &f)
    {
        std::lock_guard<std::mutex> user(user_mu);
        {
            std::lock_guard<std::mutex> lk(mu);
            job = f;
            pending = nthr;
            ++generation;
        }
        cv_work.notify_all();
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
};
static ImpdarPool *impdar_pool()
{
    // (leaked on purpose; a forked child has the pointer but none of the threads: it makes its own)
    static std::mutex mu;
    static ImpdarPool *pool = nullptr;
    static pid_t owner = 0;
    std::lock_guard<std::mutex> lk(mu);
    if (!pool || owner != getpid()) {
        pool = new ImpdarPool(std::max(1u, std::min(16u, std::thread::hardware_concurrency())));
        owner = getpid();
    }
    return pool;
}

// `n` items on the host worker threads: fn(begin, end)
template <typename F> static void impdar_parallel_for(size_t n, size_t align, F fn)
{
    ImpdarPool *pool = impdar_pool();
    const unsigned nthr = pool->nthr;
    if (n * 1 <= (size_t)nthr * align) {                   // (nothing to share out)
        fn(0, n);
        return;
    }
    pool->run([&](unsigned t) {
        const size_t a = (n * t / nthr) / align * align, b = t + 1 == nthr ? n : (n * (t + 1) / nthr) / align * align;
        if (b > a) fn(a, b);
    });
}

void impdar_host_copy_f64(double *dst, const void *src, size_t n, bool src_is_f32)
{
    if (src_is_f32) {
        const float *f = reinterpret_cast<const float *>(src);
        impdar_parallel_for(n, 16, [=](size_t a, size_t b) { for (size_t i = a; i < b; ++i) dst[i] = (double)f[i]; });
    } else {
        const double *d = reinterpret_cast<const double *>(src);
        impdar_parallel_for(n, 16, [=](size_t a, size_t b) { memcpy(dst + a, d + a, (b - a) * sizeof(double)); });
    }
}

// Device -> pageable host memory in pieces: every piece is DMA-ed into the pinned staging buffer and copied out
// (or widened to float64) on the host threads while the DMA of the following pieces is still running.
// elem_out == 0: plain byte copy; elem_out == 8 with elem_in == 4: float32 -> float64.
// width != 0: the device array is a (n / width) x width block, contiguous, and lands in columns [col0, col0 + width) of a
// host array with `ld` elements per row (an output-trace block of a sharded or split migration).
static int impdar_download_piped(impdar_ctx *ctx, void *host_dst, const void *dev_src, size_t n, size_t elem_in,
                                 bool widen, hipStream_t st, size_t width = 0, size_t ld = 0, size_t col0 = 0)
{
    const size_t bytes = n * elem_in;
    const bool block = width != 0 && !(width == ld && col0 == 0);
    const size_t elem_out = widen ? 8 : elem_in;
    // one staged download per context at a time: the staging buffer is shared (and may be re-allocated) and
    // ctypes callers run without the GIL
    std::lock_guard<std::mutex> lock(ctx->pinned_mu);
    // The staging buffer is a RING of at most 64 MB (round 6; the whole image before -- 268 MB at 8192^2 float32, 59 ms of
    // hipHostMalloc in a process's first phase-shift call): pieces of <= 16 MB go round it, up to four DMAs in flight, the
    // copy of piece c + 4 enqueued once the host threads have copied (or widened) piece c out of its slot.
    const size_t cap = std::min(bytes, IMPDAR_STAGE_RING_BYTES);
    char *stage = bytes >= (1u << 20) ? reinterpret_cast<char *>(impdar_ctx_pinned(ctx, cap)) : nullptr;
    std::vector<char> fallback;
    size_t ring = cap;
    if (!stage) {
        if (!widen && !block) {
            IMPDAR_HIP_CHECK(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, st));
            IMPDAR_HIP_CHECK(hipStreamSynchronize(st));
            return IMPDAR_OK;
        }
        fallback.resize(bytes);
        stage = fallback.data();
        ring = bytes;
    }
    const size_t gran = block ? width : 16;                  // block pieces are whole rows
    constexpr int NSLOT = 4;
    const size_t slot_bytes = ring / NSLOT >= ((size_t)1 << 20) ? ring / NSLOT / 256 * 256 : ring;
    const int nslot = slot_bytes == ring ? 1 : NSLOT;
    size_t per = std::max<size_t>(slot_bytes / elem_in / gran * gran, gran);     // elements per piece
    if (per * elem_in > slot_bytes) {                        // (a row wider than a slot: one slot, the whole ring)
        if (gran * elem_in > ring) {
            impdar_set_error("device -> host copy: a row of %zu bytes does not fit the staging ring", gran * elem_in);
            return IMPDAR_ERR_ARG;
        }
        per = std::max<size_t>(ring / elem_in / gran * gran, gran);
    }
    const int slots = per * elem_in > slot_bytes ? 1 : nslot;
    const size_t slot_stride = slots == 1 ? 0 : slot_bytes;
    const size_t npiece = (n + per - 1) / per;
    hipEvent_t ev[NSLOT] = {};
    int rc = IMPDAR_OK;
    for (int q = 0; q < slots && rc == IMPDAR_OK; ++q)
        if (hipEventCreateWithFlags(&ev[q], hipEventDisableTiming) != hipSuccess) {
            ev[q] = nullptr;
            rc = IMPDAR_ERR_HIP;
        }
    auto issue = [&](size_t c) {
        const size_t a = c * per, cnt = std::min(per, n - a);
        const int q = (int)(c % (size_t)slots);
        if (hipMemcpyAsync(stage + q * slot_stride, (const char *)dev_src + a * elem_in, cnt * elem_in, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipEventRecord(ev[q], st) != hipSuccess)
            rc = IMPDAR_ERR_HIP;
    };
    size_t issued = 0;
    for (; issued < npiece && issued < (size_t)slots && rc == IMPDAR_OK; ++issued) issue(issued);
    for (size_t c = 0; c < npiece && rc == IMPDAR_OK; ++c) {
        const int q = (int)(c % (size_t)slots);
        if (hipEventSynchronize(ev[q]) != hipSuccess) {
            rc = IMPDAR_ERR_HIP;
            break;
        }
        const size_t base = c * per, cnt = std::min(per, n - base);
        const char *sp = stage + q * slot_stride;
        if (block) {
            // rows base / width .. of the block -> host rows of `ld` elements, columns col0 ..
            const size_t r0 = base / width, nr = cnt / width;
            char *dp = reinterpret_cast<char *>(host_dst);
            impdar_parallel_for(nr, 1, [=](size_t a, size_t b) {
                for (size_t r = a; r < b; ++r) {
                    const char *src = sp + r * width * elem_in;
                    char *dst = dp + ((r0 + r) * ld + col0) * elem_out;
                    if (widen) {
                        const float *f = reinterpret_cast<const float *>(src);
                        double *d = reinterpret_cast<double *>(dst);
                        for (size_t j = 0; j < width; ++j) d[j] = (double)f[j];
                    } else {
                        memcpy(dst, src, width * elem_in);
                    }
                }
            });
        } else if (widen) {
            const float *f = reinterpret_cast<const float *>(sp);
            double *d = reinterpret_cast<double *>(host_dst) + base;
            impdar_parallel_for(cnt, 16, [=](size_t a, size_t b) { for (size_t i = a; i < b; ++i) d[i] = (double)f[i]; });
        } else {
            char *dp = reinterpret_cast<char *>(host_dst) + base * elem_in;
            impdar_parallel_for(cnt * elem_in, 64, [=](size_t a, size_t b) { memcpy(dp + a, sp + a, b - a); });
        }
        if (issued < npiece) issue(issued++);                // (the slot just emptied takes the next piece)
    }
    if (rc) {
        // no DMA into the staging buffer may still be in flight when it is handed to the next caller
        (void)hipStreamSynchronize(st);
        impdar_set_error("device -> host copy failed: %s", hipGetErrorString(hipGetLastError()));
    }
    for (int q = 0; q < slots; ++q)
        if (ev[q]) (void)hipEventDestroy(ev[q]);
    return rc;
}

int impdar_download(impdar_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes, hipStream_t st)
{
    return impdar_download_piped(ctx, host_dst, dev_src, bytes, 1, false, st);
}

// a (rows x width) device block of `dtype` into columns [col0, col0 + width) of a (rows x ld) float64 host array, on `st`
int impdar_download_block_f64(impdar_ctx *ctx, double *dst_host, size_t ld, size_t col0, const void *src_dev, int dtype,
                              size_t rows, size_t width, hipStream_t st)
{
    if (rows == 0 || width == 0) return IMPDAR_OK;
    if (dtype == IMPDAR_F64) return impdar_download_piped(ctx, dst_host, src_dev, rows * width * 8, 1, false, st, width * 8, ld * 8, col0 * 8);
    return impdar_download_piped(ctx, dst_host, src_dev, rows * width, 4, true, st, width, ld, col0);
}

// Several (rows x width[i]) device blocks of `dtype` into columns [col0[i], col0[i] + width[i]) of ONE (rows x ld) float64
// host array, each as soon as `after[i]` (the event behind the kernel that produces it) has passed: all copies are
// enqueued on `st` at once, in pieces, and the host threads widen piece k while the DMA of piece k + 1 (of this block
// or the next) runs -- the pipelined one-shot Kirchhoff call's way out.
int impdar_download_blocks_f64(impdar_ctx *ctx, double *dst_host, size_t ld, size_t rows, int dtype, int nblk,
                               const size_t *col0, const size_t *width, const void *const *src, const hipEvent_t *after,
                               hipStream_t st)
{
    const size_t elem_in = impdar_dtype_size(dtype);
    const bool widen = dtype == IMPDAR_F32;
    size_t total = 0;
    for (int i = 0; i < nblk; ++i) total += rows * width[i] * elem_in;
    if (total == 0) return IMPDAR_OK;
    // The staging buffer is a RING of at most 64 MB (round 4; the whole image before): pinning costs 0.22 ms per MB, and
    // the 164 MB of a config-3 image were the largest term of a process's first call.  Pieces of <= 16 MB go round it:
    // the copy of a piece is enqueued as soon as the ring has room, i.e. once the host has widened the piece that held
    // that room; up to four are in flight.
    // (same-box A/B of the steady call at config 3, ring 32 / 64 / 96 / 128 MB / whole image: 16.4 / 13.2 / 12.9 / 13.0 / 13.0 ms;
    // the first call of a process: 52 ms with 64 MB, 66 ms with 96 MB -- more than the call can pin behind its plan
    // before the first block is ready to leave.  64 MB.)
    const size_t cap = std::min(total, IMPDAR_STAGE_RING_BYTES);
    std::unique_lock<std::mutex> lock(ctx->pinned_mu);
    char *stage = reinterpret_cast<char *>(impdar_ctx_pinned(ctx, cap));
    if (!stage) {
        // no pinned memory: block by block through the single-block path
        lock.unlock();
        for (int i = 0; i < nblk; ++i) {
            if (after[i]) IMPDAR_HIP_CHECK(hipStreamWaitEvent(st, after[i], 0));
            const int rc = impdar_download_block_f64(ctx, dst_host, ld, col0[i], src[i], dtype, rows, width[i], st);
            if (rc) return rc;
        }
        return IMPDAR_OK;
    }
    struct Piece {
        int blk;
        size_t r0, nr, off, bytes;
        hipEvent_t ev;
    };
    std::vector<Piece> pieces;
    for (int i = 0; i < nblk; ++i) {
        if (width[i] == 0) continue;
        const size_t bytes = rows * width[i] * elem_in;
        const size_t np = std::min(rows, std::max<size_t>(1, (bytes + (cap / 4) - 1) / (cap / 4)));
        for (size_t c = 0; c < np; ++c) {
            Piece q;
            q.blk = i;
            q.r0 = rows * c / np;
            q.nr = rows * (c + 1) / np - q.r0;
            q.off = 0;
            q.bytes = q.nr * width[i] * elem_in;
            q.ev = nullptr;
            if (q.bytes > cap) {                              // (a single row wider than the ring: cannot happen below 64 MB per row)
                lock.unlock();
                impdar_set_error("download: a piece of %zu bytes does not fit the staging ring", q.bytes);
                return IMPDAR_ERR_ARG;
            }
            pieces.push_back(q);
        }
    }
    int rc = IMPDAR_OK;
    size_t head = 0, inflight = 0, done_k = 0;          // ring write offset, bytes enqueued and not yet widened, next piece to widen
    int last_waited = -1;
    auto widen_next = [&]() {
        Piece &q = pieces[done_k++];
        if (rc == IMPDAR_OK && hipEventSynchronize(q.ev) != hipSuccess) rc = IMPDAR_ERR_HIP;
        if (rc == IMPDAR_OK) {
            const size_t w = width[q.blk], c0 = col0[q.blk], r0 = q.r0;
            const char *sp = stage + q.off;
            impdar_parallel_for(q.nr, 1, [=](size_t a, size_t b) {
                for (size_t r = a; r < b; ++r) {
                    const char *s = sp + r * w * elem_in;
                    double *d = dst_host + (r0 + r) * ld + c0;
                    if (widen) {
                        const float *f = reinterpret_cast<const float *>(s);
                        for (size_t j = 0; j < w; ++j) d[j] = (double)f[j];
                    } else {
                        memcpy(d, s, w * 8);
                    }
                }
            });
        }
        inflight -= q.bytes;
    };
    for (size_t k = 0; k < pieces.size() && rc == IMPDAR_OK; ++k) {
        Piece &q = pieces[k];
        // room: contiguous from `head`, else from the ring's start once everything in flight has been widened past it
        for (;;) {
            if (head + q.bytes <= cap && inflight + q.bytes <= cap) {
                // (pieces are widened in order, so what lies at [head, head + bytes) is either free or belongs to a piece
                // still in flight further round: the second condition keeps the total inside the ring, and a wrap below
                // only happens onto widened pieces)
                bool clash = false;
                for (size_t j = done_k; j < k; ++j)
                    if (pieces[j].off < head + q.bytes && head < pieces[j].off + pieces[j].bytes) clash = true;
                if (!clash) break;
            } else if (head + q.bytes > cap) {
                head = 0;
                continue;
            }
            if (done_k >= k) {                                // nothing left to free: cannot happen (q.bytes <= cap)
                rc = IMPDAR_ERR_HIP;
                break;
            }
            widen_next();
        }
        if (rc) break;
        if (after[q.blk] && last_waited != q.blk) {
            if (hipStreamWaitEvent(st, after[q.blk], 0) != hipSuccess) rc = IMPDAR_ERR_HIP;
            last_waited = q.blk;
        }
        q.off = head;
        if (rc == IMPDAR_OK && hipEventCreateWithFlags(&q.ev, hipEventDisableTiming) != hipSuccess) {
            q.ev = nullptr;
            rc = IMPDAR_ERR_HIP;
        }
        if (rc == IMPDAR_OK &&
            (hipMemcpyAsync(stage + q.off, reinterpret_cast<const char *>(src[q.blk]) + q.r0 * width[q.blk] * elem_in, q.bytes,
                            hipMemcpyDeviceToHost, st) != hipSuccess ||
             hipEventRecord(q.ev, st) != hipSuccess))
            rc = IMPDAR_ERR_HIP;
        head += q.bytes;
        inflight += q.bytes;
    }
    while (rc == IMPDAR_OK && done_k < pieces.size() && pieces[done_k].ev) widen_next();
    if (rc) {
        // no DMA into the staging buffer may still be in flight when it is handed to the next caller
        (void)hipStreamSynchronize(st);
        impdar_set_error("device -> host copy failed: %s", hipGetErrorString(hipGetLastError()));
    }
    for (const Piece &q : pieces)
        if (q.ev) (void)hipEventDestroy(q.ev);
    return rc;
}

extern "C" int impdar_dev_download_f64(impdar_ctx *ctx, double *dst_host, const void *src_dev, int dtype, size_t n)
{
    IMPDAR_ARG_CHECK(ctx && dst_host && src_dev, "impdar_dev_download_f64: null argument");
    IMPDAR_ARG_CHECK(dtype == IMPDAR_F32 || dtype == IMPDAR_F64, "impdar_dev_download_f64: dtype must be float32 or float64");
    if (n == 0) return IMPDAR_OK;
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    if (dtype == IMPDAR_F64) return impdar_download_piped(ctx, dst_host, src_dev, n * 8, 1, false, ctx->stream);
    return impdar_download_piped(ctx, dst_host, src_dev, n, 4, true, ctx->stream);
}

extern "C" void impdar_ctx_destroy(impdar_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->aux);
    (void)hipStreamSynchronize(ctx->stream);
    impdar_stolt_forget(ctx);
    impdar_ps_forget(ctx);
    impdar_kirch_forget(ctx);
    impdar_preproc_forget(ctx);
    impdar_devcache_trim(ctx->device);
    pinned_adopt(ctx);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->ev_produced) (void)hipEventDestroy(ctx->ev_produced);
    if (ctx->ev_tic) (void)hipEventDestroy(ctx->ev_tic);
    if (ctx->ev_toc) (void)hipEventDestroy(ctx->ev_toc);
    if (ctx->ev_ktic) (void)hipEventDestroy(ctx->ev_ktic);
    if (ctx->ev_ktoc) (void)hipEventDestroy(ctx->ev_ktoc);
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

// Device arrays of the one-shot calls -- the radargram a call uploads, the image it downloads -- come from a small cache
// (round 6): hipMalloc + hipFree of a 64 ... 268 MB array were ~1.5 ms of every call through host arrays (a third of a Stolt
// call at 4096^2).  A freed array is kept (at most four, 2 GiB in total, per process) and handed to the next request of
// [its size / 1.25, its size]; impdar_release_caches and impdar_ctx_destroy empty it.  Everything a cached array was used
// for has been waited for before it went back (impdar_devcache_free synchronises the context's streams first).
struct DevCacheEntry {
    void *p;
    size_t bytes;
    int device;
    bool free_;
};
static std::mutex g_devcache_mu;
static std::vector<DevCacheEntry> g_devcache;                  // arrays handed out (free_ = false) and kept (free_ = true)
constexpr size_t DEVCACHE_MAX_BYTES = (size_t)2 << 30;
constexpr int DEVCACHE_MAX_KEPT = 4;

int impdar_devcache_alloc(int device, size_t bytes, void **dptr)
{
    if (bytes == 0) bytes = 8;
    {
        std::lock_guard<std::mutex> lk(g_devcache_mu);
        DevCacheEntry *best = nullptr;
        for (DevCacheEntry &e : g_devcache)
            if (e.free_ && e.device == device && e.bytes >= bytes && e.bytes <= bytes + bytes / 4 + ((size_t)1 << 20) && (!best || e.bytes < best->bytes)) best = &e;
        if (best) {
            best->free_ = false;
            *dptr = best->p;
            return IMPDAR_OK;
        }
    }
    DevBuf b;                                                   // (its ensure() retries after impdar_release_caches when memory is short)
    IMPDAR_HIP_CHECK(b.ensure(bytes));
    *dptr = b.p;
    b.p = nullptr;
    b.bytes = 0;
    std::lock_guard<std::mutex> lk(g_devcache_mu);
    g_devcache.push_back(DevCacheEntry{*dptr, bytes, device, false});
    return IMPDAR_OK;
}

// the caller has synchronised whatever used the array
void impdar_devcache_free(void *p)
{
    if (!p) return;
    void *drop[DEVCACHE_MAX_KEPT + 2] = {};
    int ndrop = 0;
    {
        std::lock_guard<std::mutex> lk(g_devcache_mu);
        size_t kept_bytes = 0;
        int kept = 0;
        bool found = false;
        for (DevCacheEntry &e : g_devcache) {
            if (e.p == p) {
                e.free_ = true;
                found = true;
            }
            if (e.free_) {
                kept_bytes += e.bytes;
                ++kept;
            }
        }
        if (!found) drop[ndrop++] = p;                           // (not one of ours: plain hipFree)
        // over the limits: the oldest kept arrays go
        for (size_t i = 0; i < g_devcache.size() && (kept > DEVCACHE_MAX_KEPT || kept_bytes > DEVCACHE_MAX_BYTES) && ndrop < DEVCACHE_MAX_KEPT + 2;) {
            if (g_devcache[i].free_) {
                drop[ndrop++] = g_devcache[i].p;
                kept_bytes -= g_devcache[i].bytes;
                --kept;
                g_devcache.erase(g_devcache.begin() + (long)i);
            } else {
                ++i;
            }
        }
    }
    for (int i = 0; i < ndrop; ++i) (void)hipFree(drop[i]);
}

// device < 0: every device
void impdar_devcache_trim(int device)
{
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> lk(g_devcache_mu);
        for (size_t i = 0; i < g_devcache.size();) {
            if (g_devcache[i].free_ && (device < 0 || g_devcache[i].device == device)) {
                drop.push_back(g_devcache[i].p);
                g_devcache.erase(g_devcache.begin() + (long)i);
            } else {
                ++i;
            }
        }
    }
    for (void *p : drop) (void)hipFree(p);
}

extern "C" int impdar_dev_alloc(impdar_ctx *ctx, size_t bytes, void **dptr)
{
    IMPDAR_ARG_CHECK(ctx && dptr, "null context/pointer");
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    return impdar_devcache_alloc(ctx->device, bytes, dptr);
}

extern "C" int impdar_dev_free(impdar_ctx *ctx, void *dptr)
{
    IMPDAR_ARG_CHECK(ctx, "null context");
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    IMPDAR_HIP_CHECK(hipStreamSynchronize(ctx->aux));
    IMPDAR_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    impdar_devcache_free(dptr);
    return IMPDAR_OK;
}

extern "C" int impdar_dev_upload(impdar_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes)
{
    IMPDAR_ARG_CHECK(ctx && dst_dev && src_host, "null context/pointer");
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    // an image that goes up usually comes down again, through the context's pinned staging buffer: have it pinned by a
    // thread of its own while the upload and the migration run (44 ms for the 256 MB of an 8192^2 float32 image when the
    // download has to do it itself: profiles/r05_first_call.txt)
    impdar_ctx_pinned_prefetch(ctx, std::min(bytes, IMPDAR_STAGE_RING_BYTES));
    // straight from pageable memory: the runtime's own staging pipeline reaches 17 GB/s here; copying into the
    // context's pinned buffer on host threads first was slower (9.8 -> 17 ms for 164 MB)
    IMPDAR_HIP_CHECK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    IMPDAR_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return IMPDAR_OK;
}

extern "C" int impdar_dev_download(impdar_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes)
{
    IMPDAR_ARG_CHECK(ctx && dst_host && src_dev, "null context/pointer");
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    // through the pinned staging buffer and a threaded copy: the destination is usually a fresh pageable array
    return impdar_download(ctx, dst_host, src_dev, bytes, ctx->stream);
}

extern "C" int impdar_dev_memset(impdar_ctx *ctx, void *dst_dev, int value, size_t bytes)
{
    IMPDAR_ARG_CHECK(ctx && dst_dev, "null context/pointer");
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    IMPDAR_HIP_CHECK(hipMemsetAsync(dst_dev, value, bytes, ctx->stream));
    return impdar_ctx_mark_produced(ctx);
}
