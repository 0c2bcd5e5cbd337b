// Stolt f-k migration on gfx950: rocFFT transforms + hand-written taper /
// transpose / Stolt-stretch kernels.
//
// Behaviour restated from src/impdar/lib/migrationlib/mig_python.py:126-208:
//   :152-157  linear edge taper, product cast back to the data dtype
//   :159      FK = rfft2(data, axes=(1,0))  (real FFT over time, complex over traces)
//   :171-190  KK[zj,xi] = FK interpolated linearly along omega at
//             w' = (v/2) sqrt(kz^2+kx^2), clamped to the last knot; only rows
//             zj < snum//2 are written
//   :192-200  obliquity scaling kz/sqrt(kx^2+kz^2), KK[0,0] = 0
//   :202      irfft2(KK, axes=(1,0)) -> 2*(snum//2) rows
//
// Device layout is trace-major: X[tnum][snum] real, F[tnum][m] complex with
// m = snum/2+1, so the real transforms and the omega-interpolation run along
// the contiguous axis and the trace transform is a strided batched C2C.
#include "fft.h"
#include "own_fft.h"
#include <mutex>
#include <string>
#include <sys/stat.h>
#include <unistd.h>

static std::once_flag g_fft_once;
static int g_fft_rc = IMPDAR_OK;

int impdar_fft_global_setup()
{
    std::call_once(g_fft_once, [] {
        // rocFFT keeps the kernels of the plans it has built in a small user database; without one every new process
        // fetches them again from the 1.8 GB system database (0.25 s per plan with that file in the page cache, up to
        // 1.7 s from a cold disk: profiles/r05_first_call.txt) instead of 25 ms.  Where the user has not chosen a place
        // (ROCFFT_RTC_CACHE_PATH) and rocFFT's own default is not writable ($HOME on a batch node), give it one.
        if (!getenv("ROCFFT_RTC_CACHE_PATH")) {
            std::string dir;
            const char *xdg = getenv("XDG_CACHE_HOME"), *home = getenv("HOME");
            auto usable = [](const std::string &d) {
                if (d.empty()) return false;
                (void)mkdir(d.c_str(), 0700);
                return access(d.c_str(), W_OK | X_OK) == 0;
            };
            if (xdg && *xdg && usable(xdg)) dir = std::string(xdg) + "/impdar_amd";
            else if (home && *home && usable(std::string(home) + "/.cache")) dir = std::string(home) + "/.cache/impdar_amd";
            else dir = "/tmp/impdar_amd-" + std::to_string((long)getuid());
            if (usable(dir)) {
                const std::string path = dir + "/rocfft_kernel_cache.db";
                (void)setenv("ROCFFT_RTC_CACHE_PATH", path.c_str(), 0);
                impdar_trace("rocFFT user kernel cache: %s", path.c_str());
            }
        }
        impdar_trace("rocfft_setup: start");
        if (rocfft_setup() != rocfft_status_success) {
            g_fft_rc = IMPDAR_ERR_FFT;
        }
        impdar_trace("rocfft_setup: done");
    });
    if (g_fft_rc) impdar_set_error("rocfft_setup failed");
    return g_fft_rc;
}

std::mutex &impdar_fft_plan_mutex()
{
    static std::mutex mu;
    return mu;
}

extern "C" int impdar_fft_rows_dev(impdar_ctx *ctx, int mode, int dtype, int n, int batch, const void *d_in, void *d_out, double scale)
{
    IMPDAR_ARG_CHECK(ctx && d_in && d_out, "null argument");
    IMPDAR_ARG_CHECK(mode >= 0 && mode <= 4 && (dtype == IMPDAR_F32 || dtype == IMPDAR_F64) && batch >= 1, "bad mode / dtype / batch");
    const int M = (mode == OWN_R2C || mode == OWN_C2R) ? n / 2 : n;
    IMPDAR_ARG_CHECK(n >= 2 && (n & (n - 1)) == 0 && own_fft_len_ok(M), "length %d is not a power of two in range", n);
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    OwnTwiddles tw;
    int rc;
    const size_t cin = mode == OWN_R2C ? (size_t)n : (mode == OWN_C2R ? (size_t)M + 1 : (size_t)n);
    const size_t cout = mode == OWN_R2C ? (size_t)M + 1 : (size_t)n;
    if (dtype == IMPDAR_F32) {
        if ((rc = tw.ensure<float>(n, ctx->stream))) return rc;
        rc = own_fft_launch<float>(mode, n, (size_t)batch, d_in, d_out, cin, cout, scale, tw, ctx->stream);
    } else {
        if ((rc = tw.ensure<double>(n, ctx->stream))) return rc;
        rc = own_fft_launch<double>(mode, n, (size_t)batch, d_in, d_out, cin, cout, scale, tw, ctx->stream);
    }
    if (rc) return rc;
    IMPDAR_HIP_CHECK(hipStreamSynchronize(ctx->stream));         // (the table is this call's)
    return impdar_ctx_mark_produced(ctx);
}

template <typename T> struct Cx { T x, y; };

// (snum,tnum) row-major -> tapered, transposed X[tnum][snum]
template <typename T>
__global__ __launch_bounds__(256) void stolt_taper_transpose(const T *__restrict__ in, T *__restrict__ X, int snum,
                                                             int tnum, double htaper, double vtaper, int do_taper)
{
    __shared__ T tile[64][65];
    const int k0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    // the trace weight is the thread's own for all of its rows; the 64 sample weights of the tile are computed once
    // (two float64 divisions per element were most of this kernel's time)
    __shared__ double wrow[64];
    if (threadIdx.x < 64) wrow[threadIdx.x] = impdar_taper_w(min(k0 + (int)threadIdx.x, snum - 1), snum, vtaper);
    const double h = impdar_taper_w(min(j0 + tx, tnum - 1), tnum, htaper);
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int k = k0 + r, j = j0 + tx;
        T v = 0;
        if (k < snum && j < tnum) {
            v = in[(size_t)k * tnum + j];
            if (do_taper) v = (T)(((double)v * h) * wrow[r]);      // (data*H)*V then astype(dtype), :157
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int j = j0 + r, k = k0 + tx;
        if (j < tnum && k < snum) X[(size_t)j * snum + k] = tile[tx][r];
    }
}

// Y[tnum][nout] -> out (nout,tnum) row-major
template <typename T>
__global__ __launch_bounds__(256) void stolt_transpose_back(const T *__restrict__ Y, T *__restrict__ out, int nout,
                                                            int tnum)
{
    __shared__ T tile[64][65];
    const int k0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int j = j0 + r, k = k0 + tx;
        tile[r][tx] = (j < tnum && k < nout) ? Y[(size_t)j * nout + k] : (T)0;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int k = k0 + r, j = j0 + tx;
        if (k < nout && j < tnum) out[(size_t)k * tnum + j] = tile[tx][r];
    }
}

// Stolt stretch + obliquity.  One thread per (xi, zj); zj is the fast index so
// loads of F[xi][i0], F[xi][i0+1] walk monotonically along a contiguous row.
template <typename T>
__global__ __launch_bounds__(256) void stolt_stretch(const Cx<T> *__restrict__ F, Cx<T> *__restrict__ K,
                                                     const double *__restrict__ kx, const double *__restrict__ ws,
                                                     int m, int nz, int tnum, double vel)
{
    const int zj = blockIdx.x * 256 + threadIdx.x;
    const int xi = blockIdx.y;
    if (zj >= m) return;
    Cx<T> o;
    o.x = 0;
    o.y = 0;
    if (zj < nz) {
        const double kxi = kx[xi];
        const double kz = ws[zj] * 2.0 / vel;                       // :180
        const double kk = sqrt(kz * kz + kxi * kxi);                // :188 and :196 (kx^2 + kz^2: the same sum)
        double wq = vel / 2.0 * kk;
        const double wlast = ws[m - 1];
        if (wq > wlast) wq = wlast;                                 // FITPACK clamps to the last knot
        const double dw = ws[1] - ws[0];
        int i0 = (int)floor(wq / dw);
        i0 = min(max(i0, 0), m - 2);
        while (i0 > 0 && ws[i0] > wq) --i0;
        while (i0 < m - 2 && ws[i0 + 1] <= wq) ++i0;
        const double w = (wq - ws[i0]) / (ws[i0 + 1] - ws[i0]);
        const Cx<T> a = F[(size_t)xi * m + i0], b = F[(size_t)xi * m + i0 + 1];
        // interpolated value is stored into the (complex64/128) array, then
        // scaled in place in double and rounded again (:190, :198)
        const T re = (T)((1.0 - w) * (double)a.x + w * (double)b.x);
        const T im = (T)((1.0 - w) * (double)a.y + w * (double)b.y);
        const double sc = kz / kk;                                  // :196
        o.x = (T)((double)re * sc);
        o.y = (T)((double)im * sc);
        if (zj == 0 && xi == 0) {                                   // :200
            o.x = 0;
            o.y = 0;
        }
    }
    K[(size_t)xi * m + zj] = o;
}

// The same on the spectrum as the own transforms leave it after the pass over the traces: frequency-major, F[w][kx].  One
// thread per (zj, xi) with xi the fast index: the two rows a wavenumber interpolates between move slowly with xi (kk grows
// with |kx|), so neighbouring lanes read neighbouring words of the same or the next row -- no transposes around the stretch.
template <typename T>
__global__ __launch_bounds__(256) void stolt_stretch_t(const Cx<T> *__restrict__ F, Cx<T> *__restrict__ K,
                                                       const double *__restrict__ kx, const double *__restrict__ ws,
                                                       int m, int nz, int tnum, double vel)
{
    const int xi = blockIdx.x * 256 + threadIdx.x;
    const int zj = blockIdx.y;
    if (xi >= tnum) return;
    Cx<T> o;
    o.x = 0;
    o.y = 0;
    if (zj < nz) {
        const double kxi = kx[xi];
        const double kz = ws[zj] * 2.0 / vel;                       // :180
        const double kk = sqrt(kz * kz + kxi * kxi);                // :188 and :196
        double wq = vel / 2.0 * kk;
        const double wlast = ws[m - 1];
        if (wq > wlast) wq = wlast;                                 // FITPACK clamps to the last knot
        const double dw = ws[1] - ws[0];
        int i0 = (int)floor(wq / dw);
        i0 = min(max(i0, 0), m - 2);
        while (i0 > 0 && ws[i0] > wq) --i0;
        while (i0 < m - 2 && ws[i0 + 1] <= wq) ++i0;
        const double w = (wq - ws[i0]) / (ws[i0 + 1] - ws[i0]);
        const Cx<T> a = F[(size_t)i0 * tnum + xi], b = F[(size_t)(i0 + 1) * tnum + xi];
        const T re = (T)((1.0 - w) * (double)a.x + w * (double)b.x);    // (:190, :198: stored, then scaled in double and rounded again)
        const T im = (T)((1.0 - w) * (double)a.y + w * (double)b.y);
        const double sc = kz / kk;                                  // :196
        o.x = (T)((double)re * sc);
        o.y = (T)((double)im * sc);
        if (zj == 0 && xi == 0) {                                   // :200
            o.x = 0;
            o.y = 0;
        }
    }
    K[(size_t)zj * tnum + xi] = o;
}

// The same on the spectrum as [kx >= 0][all frequencies] (round 6: the transform over the traces taken first, on the radargram's own
// rows; rows k = 0 .. tnum/2 of snum complex numbers in FFT order).  irfft2 (:202) extends KK Hermitian-wise, KK[-w][kx] =
// conj KK[w][-kx], and conj FK[i][-kx] = FK[-i][kx] for a real radargram: the negative-frequency half of row kx is the same
// interpolation, with the same weights (they depend on |kx| only), between the mirrored knots of the SAME row.  One thread per
// (kx, zj), zj the fast index; the zero-frequency and Nyquist rows of KK are zero (kz = 0 scales the first to zero, :196; KK[:nz]
// alone is filled, :190), so numpy's dropping of their imaginary parts has nothing to drop.
template <typename T>
__global__ __launch_bounds__(256) void stolt_stretch_rows(const Cx<T> *__restrict__ B, Cx<T> *__restrict__ K,
                                                          const double *__restrict__ kx, const double *__restrict__ ws,
                                                          int m, int nz, int snum, double vel)
{
    const int zj = blockIdx.x * 256 + threadIdx.x;
    const int xi = blockIdx.y;
    if (zj > nz) return;
    const Cx<T> *row = B + (size_t)xi * snum;
    Cx<T> *orow = K + (size_t)xi * snum;
    Cx<T> o, om;
    o.x = o.y = om.x = om.y = 0;
    if (zj < nz) {
        const double kxi = kx[xi];
        const double kz = ws[zj] * 2.0 / vel;                       // :180
        const double kk = sqrt(kz * kz + kxi * kxi);                // :188 and :196
        double wq = vel / 2.0 * kk;
        const double wlast = ws[m - 1];
        if (wq > wlast) wq = wlast;                                 // FITPACK clamps to the last knot
        const double dw = ws[1] - ws[0];
        int i0 = (int)floor(wq / dw);
        i0 = min(max(i0, 0), m - 2);
        while (i0 > 0 && ws[i0] > wq) --i0;
        while (i0 < m - 2 && ws[i0 + 1] <= wq) ++i0;
        const double w = (wq - ws[i0]) / (ws[i0 + 1] - ws[i0]);
        const double sc = kz / kk;                                  // :196
        {
            const Cx<T> a = row[i0], b = row[i0 + 1];
            const T re = (T)((1.0 - w) * (double)a.x + w * (double)b.x);    // (:190, :198: stored, then scaled in double and rounded again)
            const T im = (T)((1.0 - w) * (double)a.y + w * (double)b.y);
            o.x = (T)((double)re * sc);
            o.y = (T)((double)im * sc);
        }
        if (zj > 0) {
            const Cx<T> a = row[i0 == 0 ? 0 : snum - i0], b = row[snum - i0 - 1];
            const T re = (T)((1.0 - w) * (double)a.x + w * (double)b.x);
            const T im = (T)((1.0 - w) * (double)a.y + w * (double)b.y);
            om.x = (T)((double)re * sc);
            om.y = (T)((double)im * sc);
        }
        if (zj == 0 && xi == 0) {                                   // :200
            o.x = 0;
            o.y = 0;
        }
    }
    orow[zj] = o;                                                   // (zj = nz: the Nyquist row, zero)
    if (zj > 0 && zj < nz) orow[snum - zj] = om;
}

// C2R ignores the imaginary part of the DC and Nyquist bins (numpy irfft):
// clear them so any Hermitian-assuming backend agrees.
template <typename T>
__global__ void stolt_fix_hermitian(Cx<T> *K, int m, int tnum)
{
    const int xi = blockIdx.x * 256 + threadIdx.x;
    if (xi >= tnum) return;
    K[(size_t)xi * m].y = 0;
    K[(size_t)xi * m + m - 1].y = 0;
}

struct StoltPlan {
    // power-of-two sizes run on the library's own row transforms (own_fft.h) and need no rocFFT plan
    bool plans_ready = false;
    int own_calls = 0;
    OwnTwiddles tw_time, tw_trace;
    int dtype = -1, snum = 0, tnum = 0;
    const impdar_ctx *owner = nullptr;   // plans and buffers live on this context's device and stream
    FftPlan r2c, c2c_f, c2c_b, c2r;     // separate passes (IMPDAR_STOLT_FFT=1d)
    FftPlan fwd2d, inv2d;               // the same two pairs as 2-D real transforms (default)
    bool use2d = true, no_own = false;
    DevBuf X, F, K, Y, d_kx, d_ws;
    DevBuf d_taper;                     // [tnum + snum] float64 taper weights (the transform over the traces taken first)
    std::vector<double> h_taper;        // ... on the host (alive until the next call: async copy), and what they were made from
    double taper_h = -1.0, taper_v = -1.0;
};

static std::mutex g_stolt_mu;
static StoltPlan *g_stolt_plan = nullptr;     // last-used plan (sizes repeat across calls)

// called by impdar_ctx_destroy: a cached plan must not outlive the stream it was created on
// impdar_release_caches: out of device memory somewhere -- drop the cached plan unless a Stolt call is using it
static thread_local bool t_stolt_busy = false;
void impdar_stolt_trim()
{
    if (t_stolt_busy) return;
    std::unique_lock<std::mutex> lk(g_stolt_mu, std::try_to_lock);
    if (lk.owns_lock() && g_stolt_plan) {
        delete g_stolt_plan;
        g_stolt_plan = nullptr;
    }
}

void impdar_stolt_forget(const impdar_ctx *ctx)
{
    std::lock_guard<std::mutex> lk(g_stolt_mu);
    if (g_stolt_plan && g_stolt_plan->owner == ctx) {
        delete g_stolt_plan;
        g_stolt_plan = nullptr;
    }
}

template <typename T>
static int stolt_run(impdar_ctx *ctx, StoltPlan &pl, const void *d_data, int snum, int tnum, const double *kx,
                     const double *ws, double vel, double htaper, double vtaper, void *d_out)
{
    const int m = snum / 2 + 1, nz = snum / 2, nout = 2 * (snum / 2);
    hipStream_t st = ctx->stream;
    const bool dbl = sizeof(T) == 8;
    bool want2d = true, no_own = false;
    {
        // tuning knob: "1d" = four 1-D rocFFT passes; "rocfft" = rocFFT's 2-D real plans also where the own transforms apply;
        // "own" = the default, spelled out (tests that pin one implementation)
        const char *e = getenv("IMPDAR_STOLT_FFT");
        want2d = !(e && !strcmp(e, "1d"));
        no_own = e && !strcmp(e, "rocfft");
    }
    // power-of-two sizes run on the library's own row transforms (own_fft.h), every call: 0.35 ms at 4096^2 against 0.37 on
    // rocFFT's 2-D plans (round 5: the stretch on the frequency-major spectrum, two transposes fewer), nothing compiled at
    // run time, no plan to make (rocFFT: 0.3-3 s per process for lengths above 1024)
    const bool own_ok = want2d && !no_own && snum % 2 == 0 && own_fft_len_ok(snum / 2) && own_fft_len_ok(tnum);
    if (pl.owner != ctx || pl.dtype != (dbl ? IMPDAR_F64 : IMPDAR_F32) || pl.snum != snum || pl.tnum != tnum || pl.use2d != want2d || pl.no_own != no_own) {
        pl.own_calls = 0;
        pl.plans_ready = false;
        pl.dtype = -1;
        pl.h_taper.clear();
        if (pl.owner != ctx) {               // another device / stream: drop everything bound to the old one
            pl.X.release(); pl.F.release(); pl.K.release(); pl.Y.release(); pl.d_kx.release(); pl.d_ws.release(); pl.d_taper.release();
            pl.owner = ctx;
        }
        int rc;
        pl.use2d = want2d;
        pl.no_own = no_own;
        // rfft2(axes=(1,0)) (:159) = real transform over time (contiguous here), complex over the traces (rows);
        // irfft2 (:202) = complex inverse over the traces, then C2R over time.  rocFFT's 2-D real plans do
        // exactly these two passes each, with its own blocked column kernels instead of a strided batch.
        if (pl.use2d && !own_ok) {
            // (asking rocFFT for the spectrum frequency-major -- its layout before the plan's last transpose -- removes two
            // transposes but makes it pick slower kernels: 0.76 ms against 0.37 ms at 4096 x 4096 float32, round 2)
            if ((rc = impdar_parallel_plans(ctx->device, {
                     [&] { return pl.fwd2d.create2d(rocfft_transform_type_real_forward, dbl, false, snum, tnum, rocfft_array_type_real,
                                                    rocfft_array_type_hermitian_interleaved, snum, m, 1.0, st); },
                     [&] { return pl.inv2d.create2d(rocfft_transform_type_real_inverse, dbl, false, nout, tnum,
                                                    rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, m, nout,
                                                    1.0 / ((double)nout * tnum), st); }})))
                return rc;
            pl.plans_ready = true;
        }
        if (!pl.use2d) {
            // (round 3 built these four beside the 2-D pair on every new size: four run-time compilations nobody ran)
            if ((rc = pl.r2c.create(rocfft_transform_type_real_forward, dbl, false, snum, tnum, rocfft_array_type_real,
                                    rocfft_array_type_hermitian_interleaved, 1, snum, 1, m, 1.0, st)))
                return rc;
            if ((rc = pl.c2c_f.create(rocfft_transform_type_complex_forward, dbl, true, tnum, m,
                                      rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved, m, 1,
                                      m, 1, 1.0, st)))
                return rc;
            if ((rc = pl.c2c_b.create(rocfft_transform_type_complex_inverse, dbl, true, tnum, m,
                                      rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved, m, 1,
                                      m, 1, 1.0 / tnum, st)))
                return rc;
            if ((rc = pl.c2r.create(rocfft_transform_type_real_inverse, dbl, false, nout, tnum,
                                    rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, 1, m, 1, nout,
                                    1.0 / nout, st)))
                return rc;
        }
        IMPDAR_HIP_CHECK(pl.X.ensure((size_t)tnum * snum * sizeof(T)));
        // (the rows layout, below: [snum][tnum/2 + 1] and [tnum/2 + 1][snum] complex)
        const size_t spec = std::max((size_t)tnum * m, (size_t)snum * (tnum / 2 + 1));
        IMPDAR_HIP_CHECK(pl.F.ensure(spec * 2 * sizeof(T)));
        IMPDAR_HIP_CHECK(pl.K.ensure(spec * 2 * sizeof(T)));
        IMPDAR_HIP_CHECK(pl.Y.ensure((size_t)tnum * nout * sizeof(T)));
        IMPDAR_HIP_CHECK(pl.d_kx.ensure((size_t)tnum * 8));
        IMPDAR_HIP_CHECK(pl.d_ws.ensure((size_t)m * 8));
        impdar_trace("stolt: plans and buffers ready");
        pl.dtype = dbl ? IMPDAR_F64 : IMPDAR_F32;
        pl.snum = snum;
        pl.tnum = tnum;
    }
    const bool use_own = own_ok;
    if (use_own) {
        impdar_trace("stolt: transforms on the library's own row kernels");
        int rc;
        if ((rc = pl.tw_time.ensure<T>(snum, st)) || (rc = pl.tw_trace.ensure<T>(tnum, st))) return rc;
    }
    IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_kx.p, kx, (size_t)tnum * 8, hipMemcpyHostToDevice, st));
    IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_ws.p, ws, (size_t)m * 8, hipMemcpyHostToDevice, st));
    {
        int trc = impdar_ctx_tic(ctx);
        if (trc) return trc;
    }
    const int do_taper = !(htaper != htaper);     // NaN = caller already tapered (integer dtypes)
    // Round 6, power-of-two sizes: the transform over the TRACES first, on the radargram's own rows with the taper applied on load,
    // wavenumbers k = 0 .. tnum/2 only -- R2C over x -> [snum][hs]; transpose -> [hs][snum]; C2C over time; the stretch along the rows
    // (both signs of the frequency: stolt_stretch_rows); C2C back over time; transpose -> [snum][hs]; C2R over x straight into the
    // output: seven passes instead of ten (no transposes of the real arrays at either end, two half-size transposes instead of two
    // whole ones)
    const bool use_rows = use_own && own_fft_len_ok(snum) && tnum >= 64 && nout == snum;
    if (use_rows && do_taper && !(pl.h_taper.size() == (size_t)tnum + snum && pl.taper_h == htaper && pl.taper_v == vtaper && pl.d_taper.p)) {
        std::vector<double> &taps = pl.h_taper;
        taps.resize((size_t)tnum + snum);
        for (int j = 0; j < tnum; ++j) taps[j] = impdar_taper_w(j, tnum, htaper);
        for (int i = 0; i < snum; ++i) taps[(size_t)tnum + i] = impdar_taper_w(i, snum, vtaper);
        IMPDAR_HIP_CHECK(pl.d_taper.ensure(taps.size() * 8));
        IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_taper.p, taps.data(), taps.size() * 8, hipMemcpyHostToDevice, st));
        pl.taper_h = htaper;
        pl.taper_v = vtaper;
    }
    int rc;
    if (use_rows) {
        const int hs = tnum / 2 + 1;
        const double *th = do_taper ? pl.d_taper.as<double>() : nullptr, *tv = do_taper ? th + tnum : nullptr;
        if ((rc = own_fft_launch<T>(OWN_R2C, tnum, (size_t)snum, d_data, pl.F.p, (size_t)tnum, (size_t)hs, 1.0, pl.tw_trace, st, th, tv, 1))) return rc;
        own_launch_transpose<T>(pl.F.p, pl.K.p, snum, hs, st);
        if ((rc = own_fft_launch<T>(OWN_C2C_FWD, snum, (size_t)hs, pl.K.p, pl.K.p, (size_t)snum, (size_t)snum, 1.0, pl.tw_time, st))) return rc;
        hipLaunchKernelGGL((stolt_stretch_rows<T>), dim3((nz + 1 + 255) / 256, hs), dim3(256), 0, st, pl.K.as<Cx<T>>(), pl.F.as<Cx<T>>(),
                           pl.d_kx.as<double>(), pl.d_ws.as<double>(), m, nz, snum, vel);
        if ((rc = own_fft_launch<T>(OWN_C2C_INV, snum, (size_t)hs, pl.F.p, pl.F.p, (size_t)snum, (size_t)snum, 1.0 / snum, pl.tw_time, st))) return rc;
        own_launch_transpose<T>(pl.F.p, pl.K.p, hs, snum, st);
        if ((rc = own_fft_launch<T>(OWN_C2R, tnum, (size_t)snum, pl.K.p, d_out, (size_t)hs, (size_t)tnum, 1.0 / tnum, pl.tw_trace, st))) return rc;
        IMPDAR_HIP_CHECK(hipGetLastError());
        impdar_trace("stolt: all kernels enqueued (traces first)");
        ctx->m_entry = "impdar_stolt";
        ctx->m_kernel = "stolt_stretch_rows (+ the library's own row transforms, traces first)";
        ctx->m_kernel_ms = -1.f;
        ctx->ktimed = false;
        ctx->m_extra[0] = 0;
        return impdar_ctx_toc(ctx);
    }
    dim3 tgrid((tnum + 63) / 64, (snum + 63) / 64);
    hipLaunchKernelGGL((stolt_taper_transpose<T>), tgrid, dim3(256), 0, st, (const T *)d_data, pl.X.as<T>(), snum, tnum,
                       htaper, vtaper, do_taper);
    if (use_own) {
        // the four 1-D passes of the "1d" form on the library's own row kernels: real-to-complex over time (rows of X), over
        // the traces on contiguous rows of the transposed spectrum (K is free until the stretch writes it)
        if ((rc = own_fft_launch<T>(OWN_R2C, snum, (size_t)tnum, pl.X.p, pl.F.p, (size_t)snum, (size_t)m, 1.0, pl.tw_time, st))) return rc;
        own_launch_transpose<T>(pl.F.p, pl.K.p, tnum, m, st);
        if ((rc = own_fft_launch<T>(OWN_C2C_FWD, tnum, (size_t)m, pl.K.p, pl.K.p, (size_t)tnum, (size_t)tnum, 1.0, pl.tw_trace, st))) return rc;
    } else if (pl.use2d) {
        if ((rc = pl.fwd2d.exec(pl.X.p, pl.F.p))) return rc;
    } else {
        if ((rc = pl.r2c.exec(pl.X.p, pl.F.p))) return rc;
        if ((rc = pl.c2c_f.exec(pl.F.p, nullptr))) return rc;
    }
    if (use_own)      // K [w][kx] -> F [w][kx]
        hipLaunchKernelGGL((stolt_stretch_t<T>), dim3((tnum + 255) / 256, m), dim3(256), 0, st, pl.K.as<Cx<T>>(),
                           pl.F.as<Cx<T>>(), pl.d_kx.as<double>(), pl.d_ws.as<double>(), m, nz, tnum, vel);
    else
        hipLaunchKernelGGL((stolt_stretch<T>), dim3((m + 255) / 256, tnum), dim3(256), 0, st, pl.F.as<Cx<T>>(),
                           pl.K.as<Cx<T>>(), pl.d_kx.as<double>(), pl.d_ws.as<double>(), m, nz, tnum, vel);
    if (use_own) {
        if ((rc = own_fft_launch<T>(OWN_C2C_INV, tnum, (size_t)m, pl.F.p, pl.F.p, (size_t)tnum, (size_t)tnum, 1.0 / tnum, pl.tw_trace, st))) return rc;
        own_launch_transpose<T>(pl.F.p, pl.K.p, m, tnum, st);
        hipLaunchKernelGGL((stolt_fix_hermitian<T>), dim3((tnum + 255) / 256), dim3(256), 0, st, pl.K.as<Cx<T>>(), m, tnum);
        if ((rc = own_fft_launch<T>(OWN_C2R, nout, (size_t)tnum, pl.K.p, pl.Y.p, (size_t)m, (size_t)nout, 1.0 / nout, pl.tw_time, st))) return rc;
    } else if (pl.use2d) {
        if ((rc = pl.inv2d.exec(pl.K.p, pl.Y.p))) return rc;
    } else {
        if ((rc = pl.c2c_b.exec(pl.K.p, nullptr))) return rc;
        hipLaunchKernelGGL((stolt_fix_hermitian<T>), dim3((tnum + 255) / 256), dim3(256), 0, st, pl.K.as<Cx<T>>(), m, tnum);
        if ((rc = pl.c2r.exec(pl.K.p, pl.Y.p))) return rc;
    }
    dim3 bgrid((tnum + 63) / 64, (nout + 63) / 64);
    hipLaunchKernelGGL((stolt_transpose_back<T>), bgrid, dim3(256), 0, st, pl.Y.as<T>(), (T *)d_out, nout, tnum);
    IMPDAR_HIP_CHECK(hipGetLastError());
    impdar_trace("stolt: all kernels enqueued");
    ctx->m_entry = "impdar_stolt";
    ctx->m_kernel = use_own ? "stolt_stretch (+ the library's own row transforms)" : "stolt_stretch (+ rocFFT 2-D real transforms)";
    ctx->m_kernel_ms = -1.f;
    ctx->ktimed = false;
    ctx->m_extra[0] = 0;
    return impdar_ctx_toc(ctx);
}

extern "C" int impdar_stolt_dev(impdar_ctx *ctx, const void *d_data, int dtype, int snum, int tnum, const double *kx,
                                const double *ws, double vel, double htaper, double vtaper, void *d_out)
{
    IMPDAR_ARG_CHECK(ctx && d_data && d_out && kx && ws, "null argument");
    IMPDAR_ARG_CHECK(dtype == IMPDAR_F32 || dtype == IMPDAR_F64, "dtype must be 0 (f32) or 1 (f64)");
    IMPDAR_ARG_CHECK(snum >= 2 && tnum >= 1, "need snum >= 2 and tnum >= 1 (got %d, %d)", snum, tnum);
    IMPDAR_ARG_CHECK(vel > 0, "vel must be positive");
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> lk(g_stolt_mu);
    ImpdarBusy busy(t_stolt_busy);
    if (!g_stolt_plan) g_stolt_plan = new StoltPlan();
    int rc = dtype == IMPDAR_F32
                 ? stolt_run<float>(ctx, *g_stolt_plan, d_data, snum, tnum, kx, ws, vel, htaper, vtaper, d_out)
                 : stolt_run<double>(ctx, *g_stolt_plan, d_data, snum, tnum, kx, ws, vel, htaper, vtaper, d_out);
    if (rc) return rc;
    // kx/ws were staged from caller memory: complete before returning
    IMPDAR_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    impdar_trace("stolt: device work complete");
    return impdar_ctx_mark_produced(ctx);
}

extern "C" int impdar_stolt(impdar_ctx *ctx, const void *data, int dtype, int snum, int tnum, const double *kx,
                            const double *ws, double vel, double htaper, double vtaper, void *out)
{
    IMPDAR_ARG_CHECK(ctx && data && out, "null argument");
    IMPDAR_ARG_CHECK(dtype == IMPDAR_F32 || dtype == IMPDAR_F64, "dtype must be 0 (f32) or 1 (f64)");
    IMPDAR_ARG_CHECK(snum >= 2 && tnum >= 1, "need snum >= 2 and tnum >= 1 (got %d, %d)", snum, tnum);
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t esz = impdar_dtype_size(dtype);
    const size_t inb = (size_t)snum * tnum * esz, outb = (size_t)(2 * (snum / 2)) * tnum * esz;
    impdar_trace("impdar_stolt: enter (%d x %d)", snum, tnum);
    impdar_ctx_pinned_prefetch(ctx, std::min(outb, IMPDAR_STAGE_RING_BYTES));       // the download's staging ring, pinned while the call works
    // (the two device arrays of the call come from the cache of freed ones: no hipMalloc / hipFree per call)
    void *din = nullptr, *dout = nullptr;
    int rc = impdar_devcache_alloc(ctx->device, inb, &din);
    if (rc == IMPDAR_OK) rc = impdar_devcache_alloc(ctx->device, outb ? outb : 8, &dout);
    if (rc == IMPDAR_OK && hipMemcpyAsync(din, data, inb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
        impdar_set_error("impdar_stolt: upload failed: %s", hipGetErrorString(hipGetLastError()));
        rc = IMPDAR_ERR_HIP;
    }
    impdar_trace("impdar_stolt: upload enqueued");
    if (rc == IMPDAR_OK) rc = impdar_stolt_dev(ctx, din, dtype, snum, tnum, kx, ws, vel, htaper, vtaper, dout);
    if (rc == IMPDAR_OK) rc = impdar_download(ctx, out, dout, outb, ctx->stream);
    impdar_trace("impdar_stolt: downloaded");
    (void)hipStreamSynchronize(ctx->stream);                  // (nothing in flight may still touch the arrays)
    impdar_devcache_free(din);
    impdar_devcache_free(dout);
    return rc;
}
