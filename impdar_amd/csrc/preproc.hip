// The two processing steps that sit in front of a migration in an impproc/impdar chain, kept on the device
// so that a radargram stays resident from the first filter to the migrated image:
//
//   * vertical_band_pass  (reference src/impdar/lib/RadarData/_RadarDataFiltering.py:469-549):
//       IIR designs -> scipy.signal.filtfilt(b, a, data, axis=0): odd extension by 3*ntaps samples, steady-state
//       initial conditions, forward pass, backward pass, cast back to the data's dtype;
//       FIR design  -> lfilter(taps, 1, data) shifted up by `order` rows, the last `order` rows untouched.
//   * constant_space      (reference src/impdar/lib/RadarData/_RadarDataProcessing.py:499-583):
//       linear interpolation of every sample row onto equally spaced distances (scipy interp1d, slope form).
//
// All three are HBM-streaming kernels; the arithmetic is fp64 in SciPy's operation order whatever the data
// type (the file is compiled with -ffp-contract=off).  The IIR recurrence is a serial chain along time, so it
// runs one trace per lane (rows of the (snum, tnum) array are contiguous across traces: every access is a
// coalesced row segment) with the loads of the next 8 samples in flight while 8 are filtered.
#include "common.h"

#define FF_MAX_COEF 33

struct FiltCoefs {
    double b[FF_MAX_COEF];
    double a[FF_MAX_COEF];
    double zi[FF_MAX_COEF];
};

// sample i of the odd extension of trace j (scipy.signal._arraytools.odd_ext): computed in the data's own
// arithmetic (2*x[0] - x[edge-i] is a float32 expression for float32 data), then widened
template <typename T>
__device__ __forceinline__ double ff_ext(const T *__restrict__ x, int i, int j, int snum, int ld, int edge)
{
    if (i < edge) {
        const T e = x[j], v = x[(size_t)(edge - i) * ld + j];
        return (double)(T)((T)2 * e - v);
    }
    i -= edge;
    if (i < snum) return (double)x[(size_t)i * ld + j];
    i -= snum;
    const T e = x[(size_t)(snum - 1) * ld + j], v = x[(size_t)(snum - 2 - i) * ld + j];
    return (double)(T)((T)2 * e - v);
}

// one step of the transposed direct-form II recurrence, in the order of SciPy's C loop (_lfilter.c.in):
//   y = z[0] + b[0]*x;  z[n] = z[n+1] + x*b[n+1] - y*a[n+1];  z[last] = x*b[last] - y*a[last]
template <int NC> __device__ __forceinline__ double ff_step(double (&z)[NC - 1], const FiltCoefs &c, double xn)
{
    const double y = z[0] + c.b[0] * xn;
#pragma unroll
    for (int n = 0; n < NC - 2; ++n) z[n] = z[n + 1] + xn * c.b[n + 1] - y * c.a[n + 1];
    z[NC - 2] = xn * c.b[NC - 1] - y * c.a[NC - 1];
    return y;
}

// forward pass over the extended trace; Y is (snum + 2*edge, tnum) fp64
template <typename T, int NC>
__global__ __launch_bounds__(64) void ff_forward_kernel(const T *__restrict__ x, double *__restrict__ Y, int snum,
                                                        int tnum, int edge, FiltCoefs c)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j >= tnum) return;
    const int L = snum + 2 * edge;
    double z[NC - 1];
    const double x0 = ff_ext(x, 0, j, snum, tnum, edge);
#pragma unroll
    for (int n = 0; n < NC - 1; ++n) z[n] = c.zi[n] * x0;
    int i = 0;
    for (; i + 8 <= L; i += 8) {
        double xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) xv[u] = ff_ext(x, i + u, j, snum, tnum, edge);
#pragma unroll
        for (int u = 0; u < 8; ++u) Y[(size_t)(i + u) * tnum + j] = ff_step<NC>(z, c, xv[u]);
    }
    for (; i < L; ++i) Y[(size_t)i * tnum + j] = ff_step<NC>(z, c, ff_ext(x, i, j, snum, tnum, edge));
}

// backward pass: filters Y from its last row to its first and writes rows [edge, edge+snum) back into the
// data array in its own dtype (the rows in front of `edge` are never needed)
template <typename T, int NC>
__global__ __launch_bounds__(64) void ff_backward_kernel(const double *__restrict__ Y, T *__restrict__ out, int snum,
                                                         int tnum, int edge, FiltCoefs c)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j >= tnum) return;
    const int L = snum + 2 * edge;
    double z[NC - 1];
    const double y0 = Y[(size_t)(L - 1) * tnum + j];
#pragma unroll
    for (int n = 0; n < NC - 1; ++n) z[n] = c.zi[n] * y0;
    int p = L - 1;   // position in the extended trace
    for (; p - 7 >= edge; p -= 8) {
        double yv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) yv[u] = Y[(size_t)(p - u) * tnum + j];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double v = ff_step<NC>(z, c, yv[u]);
            const int r = p - u - edge;
            if (r < snum) out[(size_t)r * tnum + j] = (T)v;
        }
    }
    for (; p >= edge; --p) {
        const double v = ff_step<NC>(z, c, Y[(size_t)p * tnum + j]);
        if (p - edge < snum) out[(size_t)(p - edge) * tnum + j] = (T)v;
    }
}

struct FirTaps {
    double t[256];
};

// out[k, j] = sum_i taps[i] * x[k + order - i, j] for k < snum - order (lfilter delayed by `order` rows and
// shifted back, _RadarDataFiltering.py:536-540); `out` is a separate array
template <typename T>
__global__ __launch_bounds__(256) void fir_shift_kernel(const T *__restrict__ x, T *__restrict__ out, int snum, int tnum,
                                                        int ntaps, FirTaps taps)
{
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int order = ntaps - 1;
    const size_t n = (size_t)(snum - order) * tnum;
    if (id >= n) return;
    const int k = (int)(id / tnum), j = (int)(id % tnum);
    double s = 0.0;
    for (int i = 0; i < ntaps; ++i) s += taps.t[i] * (double)x[(size_t)(k + order - i) * tnum + j];
    out[id] = (T)s;
}

// out[k, m] = (y_hi - y_lo) / den[m] * t[m] + y_lo   (scipy interp1d._call_linear: the difference is taken in
// the data's own arithmetic, everything after it in fp64)
template <typename T>
__global__ __launch_bounds__(256) void trace_lerp_kernel(const T *__restrict__ x, double *__restrict__ out, int snum,
                                                         int tnum, int n_new, const int *__restrict__ lo,
                                                         const int *__restrict__ hi, const double *__restrict__ den,
                                                         const double *__restrict__ t)
{
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (size_t)snum * n_new) return;
    const int k = (int)(id / n_new), m = (int)(id % n_new);
    const T ylo = x[(size_t)k * tnum + lo[m]], yhi = x[(size_t)k * tnum + hi[m]];
    const double slope = (double)(T)(yhi - ylo) / den[m];
    out[id] = slope * t[m] + (double)ylo;
}

// ------------------------------------------------------------------------------------------------ host side

struct PreprocScratch {
    impdar_ctx *owner = nullptr;
    DevBuf y, data, aux, idx;
};
static PreprocScratch g_scr;

static void scratch_bind(impdar_ctx *ctx)
{
    if (g_scr.owner != ctx) {
        g_scr.y.release();
        g_scr.data.release();
        g_scr.aux.release();
        g_scr.idx.release();
        g_scr.owner = ctx;
    }
}

void impdar_preproc_forget(impdar_ctx *ctx)
{
    if (g_scr.owner == ctx) {
        g_scr.y.release();
        g_scr.data.release();
        g_scr.aux.release();
        g_scr.idx.release();
        g_scr.owner = nullptr;
    }
}

template <typename T, int NC>
static int filtfilt_launch(impdar_ctx *ctx, T *d, double *Y, int snum, int tnum, int edge, const FiltCoefs &c)
{
    const int nb = (tnum + 63) / 64;
    hipLaunchKernelGGL((ff_forward_kernel<T, NC>), dim3(nb), dim3(64), 0, ctx->stream, d, Y, snum, tnum, edge, c);
    hipLaunchKernelGGL((ff_backward_kernel<T, NC>), dim3(nb), dim3(64), 0, ctx->stream, Y, d, snum, tnum, edge, c);
    IMPDAR_HIP_CHECK(hipGetLastError());
    return IMPDAR_OK;
}

template <typename T>
static int filtfilt_dispatch(impdar_ctx *ctx, T *d, double *Y, int snum, int tnum, int edge, int ncoef, const FiltCoefs &c)
{
    if (ncoef <= 5) return filtfilt_launch<T, 5>(ctx, d, Y, snum, tnum, edge, c);
    if (ncoef <= 11) return filtfilt_launch<T, 11>(ctx, d, Y, snum, tnum, edge, c);
    if (ncoef <= 21) return filtfilt_launch<T, 21>(ctx, d, Y, snum, tnum, edge, c);
    return filtfilt_launch<T, FF_MAX_COEF>(ctx, d, Y, snum, tnum, edge, c);
}

extern "C" int impdar_filtfilt_dev(impdar_ctx *ctx, void *d_data, int dtype, int snum, int tnum, const double *b,
                                   const double *a, int ncoef, const double *zi)
{
    IMPDAR_ARG_CHECK(ctx && d_data && b && a && zi, "impdar_filtfilt: null argument");
    IMPDAR_ARG_CHECK(dtype == IMPDAR_F32 || dtype == IMPDAR_F64, "impdar_filtfilt: dtype must be float32 or float64");
    IMPDAR_ARG_CHECK(ncoef >= 2 && ncoef <= FF_MAX_COEF, "impdar_filtfilt: %d filter coefficients (2..%d supported)", ncoef,
                     FF_MAX_COEF);
    IMPDAR_ARG_CHECK(a[0] != 0.0, "impdar_filtfilt: a[0] is zero");
    const int edge = 3 * ncoef;
    // scipy.signal.filtfilt's own guard and message
    IMPDAR_ARG_CHECK(snum > edge, "The length of the input vector x must be greater than padlen, which is %d.", edge);
    IMPDAR_ARG_CHECK(tnum >= 1, "impdar_filtfilt: empty radargram");
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    FiltCoefs c;
    memset(&c, 0, sizeof(c));
    for (int n = 0; n < ncoef; ++n) {   // SciPy normalises by a[0] once, up front
        c.b[n] = b[n] / a[0];
        c.a[n] = a[n] / a[0];
    }
    for (int n = 0; n < ncoef - 1; ++n) c.zi[n] = zi[n];
    scratch_bind(ctx);
    IMPDAR_HIP_CHECK(g_scr.y.ensure((size_t)(snum + 2 * edge) * tnum * sizeof(double)));
    if (dtype == IMPDAR_F32) return filtfilt_dispatch<float>(ctx, (float *)d_data, g_scr.y.as<double>(), snum, tnum, edge, ncoef, c);
    return filtfilt_dispatch<double>(ctx, (double *)d_data, g_scr.y.as<double>(), snum, tnum, edge, ncoef, c);
}

extern "C" int impdar_fir_shift_dev(impdar_ctx *ctx, void *d_data, int dtype, int snum, int tnum, const double *taps,
                                    int ntaps)
{
    IMPDAR_ARG_CHECK(ctx && d_data && taps, "impdar_fir_shift: null argument");
    IMPDAR_ARG_CHECK(dtype == IMPDAR_F32 || dtype == IMPDAR_F64, "impdar_fir_shift: dtype must be float32 or float64");
    IMPDAR_ARG_CHECK(ntaps >= 1 && ntaps <= 256, "impdar_fir_shift: %d taps (1..256 supported)", ntaps);
    IMPDAR_ARG_CHECK(tnum >= 1 && snum >= 1, "impdar_fir_shift: empty radargram");
    const int order = ntaps - 1;
    if (snum <= order) return IMPDAR_OK;   // data[:-order] is empty: nothing is assigned
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    FirTaps t;
    memset(&t, 0, sizeof(t));
    for (int i = 0; i < ntaps; ++i) t.t[i] = taps[i];
    const size_t es = impdar_dtype_size(dtype), n = (size_t)(snum - order) * tnum;
    scratch_bind(ctx);
    IMPDAR_HIP_CHECK(g_scr.aux.ensure(n * es));
    const unsigned nb = (unsigned)((n + 255) / 256);
    if (dtype == IMPDAR_F32)
        hipLaunchKernelGGL(fir_shift_kernel<float>, dim3(nb), dim3(256), 0, ctx->stream, (const float *)d_data,
                           g_scr.aux.as<float>(), snum, tnum, ntaps, t);
    else
        hipLaunchKernelGGL(fir_shift_kernel<double>, dim3(nb), dim3(256), 0, ctx->stream, (const double *)d_data,
                           g_scr.aux.as<double>(), snum, tnum, ntaps, t);
    IMPDAR_HIP_CHECK(hipGetLastError());
    IMPDAR_HIP_CHECK(hipMemcpyAsync(d_data, g_scr.aux.p, n * es, hipMemcpyDeviceToDevice, ctx->stream));
    return IMPDAR_OK;
}

extern "C" int impdar_trace_lerp_dev(impdar_ctx *ctx, const void *d_data, int dtype, int snum, int tnum, const int *lo,
                                     const int *hi, const double *den, const double *t, int n_new, double *d_out)
{
    IMPDAR_ARG_CHECK(ctx && d_data && lo && hi && den && t && d_out, "impdar_trace_lerp: null argument");
    IMPDAR_ARG_CHECK(dtype == IMPDAR_F32 || dtype == IMPDAR_F64, "impdar_trace_lerp: dtype must be float32 or float64");
    IMPDAR_ARG_CHECK(snum >= 1 && tnum >= 2 && n_new >= 0, "impdar_trace_lerp: bad shape %d x %d -> %d", snum, tnum, n_new);
    if (n_new == 0) return IMPDAR_OK;
    for (int m = 0; m < n_new; ++m)
        IMPDAR_ARG_CHECK(lo[m] >= 0 && lo[m] < tnum && hi[m] >= 0 && hi[m] < tnum, "impdar_trace_lerp: column index out of range at %d",
                         m);
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    scratch_bind(ctx);
    const size_t ib = (size_t)n_new * sizeof(int), db = (size_t)n_new * sizeof(double);
    IMPDAR_HIP_CHECK(g_scr.idx.ensure(2 * ib + 2 * db + 64));
    char *base = g_scr.idx.as<char>();
    double *d_den = (double *)base, *d_t = (double *)(base + db);
    int *d_lo = (int *)(base + 2 * db), *d_hi = (int *)(base + 2 * db + ib);
    // the tables are small; synchronous copies keep the caller's host arrays free to go away
    IMPDAR_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    IMPDAR_HIP_CHECK(hipMemcpy(d_den, den, db, hipMemcpyHostToDevice));
    IMPDAR_HIP_CHECK(hipMemcpy(d_t, t, db, hipMemcpyHostToDevice));
    IMPDAR_HIP_CHECK(hipMemcpy(d_lo, lo, ib, hipMemcpyHostToDevice));
    IMPDAR_HIP_CHECK(hipMemcpy(d_hi, hi, ib, hipMemcpyHostToDevice));
    const size_t n = (size_t)snum * n_new;
    const unsigned nb = (unsigned)((n + 255) / 256);
    if (dtype == IMPDAR_F32)
        hipLaunchKernelGGL(trace_lerp_kernel<float>, dim3(nb), dim3(256), 0, ctx->stream, (const float *)d_data, d_out, snum,
                           tnum, n_new, d_lo, d_hi, d_den, d_t);
    else
        hipLaunchKernelGGL(trace_lerp_kernel<double>, dim3(nb), dim3(256), 0, ctx->stream, (const double *)d_data, d_out, snum,
                           tnum, n_new, d_lo, d_hi, d_den, d_t);
    IMPDAR_HIP_CHECK(hipGetLastError());
    return IMPDAR_OK;
}

// ---- host-buffer forms: upload, run, download ------------------------------------------------------------

static int stage_in(impdar_ctx *ctx, const void *host, size_t bytes)
{
    scratch_bind(ctx);
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    IMPDAR_HIP_CHECK(g_scr.data.ensure(bytes));
    IMPDAR_HIP_CHECK(hipMemcpyAsync(g_scr.data.p, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return IMPDAR_OK;
}

extern "C" int impdar_filtfilt(impdar_ctx *ctx, void *data, int dtype, int snum, int tnum, const double *b,
                               const double *a, int ncoef, const double *zi)
{
    IMPDAR_ARG_CHECK(ctx && data, "impdar_filtfilt: null argument");
    IMPDAR_ARG_CHECK(snum >= 1 && tnum >= 1, "impdar_filtfilt: empty radargram");
    const size_t bytes = (size_t)snum * tnum * impdar_dtype_size(dtype);
    int rc = stage_in(ctx, data, bytes);
    if (rc) return rc;
    rc = impdar_filtfilt_dev(ctx, g_scr.data.p, dtype, snum, tnum, b, a, ncoef, zi);
    if (rc) return rc;
    return impdar_download(ctx, data, g_scr.data.p, bytes, ctx->stream);
}

extern "C" int impdar_fir_shift(impdar_ctx *ctx, void *data, int dtype, int snum, int tnum, const double *taps, int ntaps)
{
    IMPDAR_ARG_CHECK(ctx && data, "impdar_fir_shift: null argument");
    IMPDAR_ARG_CHECK(snum >= 1 && tnum >= 1, "impdar_fir_shift: empty radargram");
    const size_t bytes = (size_t)snum * tnum * impdar_dtype_size(dtype);
    int rc = stage_in(ctx, data, bytes);
    if (rc) return rc;
    rc = impdar_fir_shift_dev(ctx, g_scr.data.p, dtype, snum, tnum, taps, ntaps);
    if (rc) return rc;
    return impdar_download(ctx, data, g_scr.data.p, bytes, ctx->stream);
}

extern "C" int impdar_trace_lerp(impdar_ctx *ctx, const void *data, int dtype, int snum, int tnum, const int *lo,
                                 const int *hi, const double *den, const double *t, int n_new, double *out)
{
    IMPDAR_ARG_CHECK(ctx && data && out, "impdar_trace_lerp: null argument");
    IMPDAR_ARG_CHECK(snum >= 1 && tnum >= 2 && n_new >= 0, "impdar_trace_lerp: bad shape %d x %d -> %d", snum, tnum, n_new);
    if (n_new == 0) return IMPDAR_OK;
    const size_t bytes = (size_t)snum * tnum * impdar_dtype_size(dtype);
    int rc = stage_in(ctx, data, bytes);
    if (rc) return rc;
    IMPDAR_HIP_CHECK(g_scr.aux.ensure((size_t)snum * n_new * sizeof(double)));
    rc = impdar_trace_lerp_dev(ctx, g_scr.data.p, dtype, snum, tnum, lo, hi, den, t, n_new, g_scr.aux.as<double>());
    if (rc) return rc;
    return impdar_download(ctx, out, g_scr.aux.p, (size_t)snum * n_new * sizeof(double), ctx->stream);
}
