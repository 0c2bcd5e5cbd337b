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
// type (the file is compiled with -ffp-contract=off).  The IIR recurrence is a serial chain along time: a
// wavefront runs 16 traces, four lanes each (rows of the (snum, tnum) array are contiguous across traces, so
// every access is a coalesced row segment), with the loads of the next 32 samples in flight while 32 are filtered.
#include "common.h"
#include <mutex>

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

// The recurrence of one trace is spread over the four lanes of a quad: lane q keeps the K delays
// z[qK .. qK+K-1] (K = ceil((NC-1)/4), padded with zero coefficients) and their coefficients in registers.
// fp64 issues at half rate and a trace-per-lane mapping leaves 85 % of the SIMDs idle at 10000 traces (157
// wavefronts for 1024 SIMDs), so the ~4(NC-1) fp64 operations of a step are the whole cost: split four ways
// they take a third of the issue slots per wavefront on four times as many SIMDs.  Every delay is still
// updated by exactly SciPy's expression (_lfilter.c.in),
//   y = z[0] + b[0]*x;  z[n] = z[n+1] + x*b[n+1] - y*a[n+1];  z[last] = x*b[last] - y*a[last],
// from the old value of its neighbour, fetched across lanes with quad DPP moves before anything is updated.
__device__ __forceinline__ double ff_quad_bcast0(double v)
{
#ifdef FF_DIAG_NODPP   // timing ablation only
    return v;
#endif
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x00, 0xf, 0xf, true);   // quad_perm [0,0,0,0]
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x00, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double ff_quad_next(double v)
{
#ifdef FF_DIAG_NODPP
    return v * 0.5;
#endif
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0xF9, 0xf, 0xf, true);   // quad_perm [1,2,3,3]
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0xF9, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

template <int K> struct FfLane {
    double z[K], B[K], A[K];
    double b0;
    bool last;   // lane 3 of the quad: nothing follows its last delay
    __device__ __forceinline__ void init(const FiltCoefs &c, int q, int nc, double x0)
    {
        b0 = c.b[0];
        last = q == 3;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int n = q * K + k;   // delay index; its coefficients are b[n+1], a[n+1]
            const bool real = n + 1 < nc;
            B[k] = real ? c.b[n + 1] : 0.0;
            A[k] = real ? c.a[n + 1] : 0.0;
            z[k] = (n < nc - 1) ? c.zi[n] * x0 : 0.0;
        }
    }
    __device__ __forceinline__ double step(double xn)
    {
        const double y = ff_quad_bcast0(z[0]) + b0 * xn;
        double zn = ff_quad_next(z[0]);
        if (last) zn = 0.0;
#pragma unroll
        for (int k = 0; k < K - 1; ++k) z[k] = z[k + 1] + xn * B[k] - y * A[k];
        z[K - 1] = zn + xn * B[K - 1] - y * A[K - 1];
        return y;
    }
};

#define FF_TRACES 16   // traces per 64-lane wavefront
#define FF_CH 32       // samples per chunk (64 needs more than 256 VGPRs for the 21- and 33-coefficient kernels)
#define FF_PK (FF_CH / 4)

// Memory side.  625 wavefronts with a 64-byte row segment each cannot keep HBM busy unless many rows are in
// flight per wavefront (16 rows in flight = 640 KB on the whole chip = 0.6 TB/s at ~1 us latency), so input
// and output are quad-packed: the four lanes of a quad hold four consecutive samples of their trace (lane q of
// register k holds sample 4k+q of the chunk), one load / store instruction moves four rows, and a 32-sample
// chunk is 8 instructions.  A step takes its sample from lane u&3 of register u>>2 with a quad-broadcast DPP
// move, and lane u&3 keeps the step's output.
//
// With plain loads hipcc waits for the prefetched chunk before the first step of the current one (it treats
// loads and stores pending on vmcnt as completing out of order and drains), which exposes the memory latency
// of every chunk.  The prefetch of the next chunk is therefore issued from inline asm, invisible to that
// bookkeeping, and retired by one explicit s_waitcnt vmcnt(0) at the END of the chunk, tied to the loaded
// registers.  Nothing is assumed about the relative completion order of loads and stores: the outputs of a
// chunk stay in registers and are stored at the top of the next chunk, in front of the next prefetch, so that
// by the time of the wait everything outstanding was issued a whole chunk (~2.5 us) earlier.
// The kernels must stay within the 256 architectural VGPRs: beyond that the compiler parks live values in
// AccVGPRs, and a copy of a register whose asm load has not landed yet copies garbage.
__device__ __forceinline__ void ff_load_async(float &dst, const float *p)
{
    asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void ff_load_async(double &dst, const double *p)
{
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
#define FF_TIE8(r) "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
template <typename T> __device__ __forceinline__ void ff_wait_chunk(T (&r)[FF_PK])
{
    static_assert(FF_PK == 8, "operand list above");
    asm volatile("s_waitcnt vmcnt(0)" : FF_TIE8(r) : : "memory");
}

// value of lane L (0..3) of the quad
template <int L> __device__ __forceinline__ float ff_quad_pick(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), L * 0x55, 0xf, 0xf, true));
}
template <int L> __device__ __forceinline__ double ff_quad_pick(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), L * 0x55, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), L * 0x55, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// sample i of the odd extension in the data's own type (see ff_ext); i may lie past the end (clamped)
template <typename T> __device__ __forceinline__ T ff_ext_t(const T *__restrict__ x, int i, int j, int snum, int ld, int edge)
{
    const int L = snum + 2 * edge;
    i = i < L ? i : L - 1;
    if (i < edge) return (T)((T)2 * x[j] - x[(size_t)(edge - i) * ld + j]);
    i -= edge;
    if (i < snum) return x[(size_t)i * ld + j];
    i -= snum;
    return (T)((T)2 * x[(size_t)(snum - 1) * ld + j] - x[(size_t)(snum - 2 - i) * ld + j]);
}

// FF_CH steps on the packed samples in `cur`; the packed outputs go to `held` (converted to TO)
template <typename TI, typename TO, int K>
__device__ __forceinline__ void ff_chunk_steps(FfLane<K> &f, const TI (&cur)[FF_PK], TO (&held)[FF_PK], int q)
{
#pragma unroll
    for (int k = 0; k < FF_PK; ++k) {
        const double y0 = f.step((double)ff_quad_pick<0>(cur[k]));
        const double y1 = f.step((double)ff_quad_pick<1>(cur[k]));
        const double y2 = f.step((double)ff_quad_pick<2>(cur[k]));
        const double y3 = f.step((double)ff_quad_pick<3>(cur[k]));
        const double lo = (q & 1) ? y1 : y0, hi = (q & 1) ? y3 : y2;
        held[k] = (TO)((q & 2) ? hi : lo);
    }
}

// forward pass over the extended trace; Y is (snum + 2*edge, tnum) fp64
template <typename T, int NC>
__global__ __launch_bounds__(64) void ff_forward_kernel(const T *__restrict__ x, double *__restrict__ Y, int snum,
                                                        int tnum, int edge, int nc, FiltCoefs c)
{
    constexpr int K = (NC - 1 + 3) / 4;
    const int j = blockIdx.x * FF_TRACES + (threadIdx.x >> 2), q = threadIdx.x & 3;
    if (j >= tnum) return;   // whole quads leave together
    const int L = snum + 2 * edge;
    FfLane<K> f;
    f.init(c, q, nc, (double)ff_ext_t(x, 0, j, snum, tnum, edge));
    T cur[FF_PK], nxt[FF_PK];
    double held[FF_PK];
    int held_i = -1;   // first sample of the chunk whose outputs are still in registers (uniform)
#pragma unroll
    for (int k = 0; k < FF_PK; ++k) cur[k] = ff_ext_t(x, 4 * k + q, j, snum, tnum, edge);
    for (int i = 0; i < L; i += FF_CH) {
        if (held_i >= 0) {
#pragma unroll
            for (int k = 0; k < FF_PK; ++k) {
                const int r = held_i + 4 * k + q;
                if (r < L) Y[(size_t)r * tnum + j] = held[k];
            }
        }
        const int r0 = i + FF_CH;                                      // first sample of the next chunk
        const bool inside = r0 >= edge && r0 + FF_CH <= edge + snum;   // uniform: plain rows of the data
        if (inside) {
            const T *row = x + (size_t)(r0 - edge + q) * tnum + j;
#pragma unroll
            for (int k = 0; k < FF_PK; ++k) {
#ifdef FF_DIAG_NOLOAD   // timing ablation only
                nxt[k] = (T)(k + threadIdx.x) + (T)(size_t)row;
#else
                ff_load_async(nxt[k], row + (size_t)(4 * k) * tnum);
#endif
            }
        } else if (r0 < L) {
#pragma unroll
            for (int k = 0; k < FF_PK; ++k) nxt[k] = ff_ext_t(x, r0 + 4 * k + q, j, snum, tnum, edge);
        }
        ff_chunk_steps(f, cur, held, q);   // samples past L-1 are filtered too; their outputs are never stored
        held_i = i;
#ifndef FF_DIAG_NOLOAD
        if (inside) ff_wait_chunk(nxt);
#endif
#pragma unroll
        for (int k = 0; k < FF_PK; ++k) cur[k] = nxt[k];
    }
#pragma unroll
    for (int k = 0; k < FF_PK; ++k) {
        const int r = held_i + 4 * k + q;
        if (r < L) Y[(size_t)r * tnum + j] = held[k];
    }
}

// backward pass: filters Y from its last row to its first and writes rows [edge, edge+snum) back into the
// data array in its own dtype (the rows in front of `edge` are never needed)
template <typename T, int NC>
__global__ __launch_bounds__(64) void ff_backward_kernel(const double *__restrict__ Y, T *__restrict__ out, int snum,
                                                         int tnum, int edge, int nc, FiltCoefs c)
{
    constexpr int K = (NC - 1 + 3) / 4;
    const int j = blockIdx.x * FF_TRACES + (threadIdx.x >> 2), q = threadIdx.x & 3;
    if (j >= tnum) return;
    const int L = snum + 2 * edge;
    FfLane<K> f;
    f.init(c, q, nc, Y[(size_t)(L - 1) * tnum + j]);
    double cur[FF_PK], nxt[FF_PK];
    T held[FF_PK];
    int held_p = -1;   // position (in the extended trace) of the first sample of the chunk held in registers
#pragma unroll
    for (int k = 0; k < FF_PK; ++k) {
        const int r = L - 1 - (4 * k + q);
        cur[k] = Y[(size_t)(r > 0 ? r : 0) * tnum + j];
    }
    for (int p = L - 1; p >= edge; p -= FF_CH) {   // p: position of the chunk's first sample
        if (held_p >= 0) {
#pragma unroll
            for (int k = 0; k < FF_PK; ++k) {
                const int r = held_p - (4 * k + q) - edge;
                if (r >= 0 && r < snum) out[(size_t)r * tnum + j] = held[k];
            }
        }
        const bool inside = p - 2 * FF_CH + 1 >= 0;   // uniform: the next chunk's rows all exist
        if (inside) {
            const double *row = Y + (size_t)(p - FF_CH - q) * tnum + j;
#pragma unroll
            for (int k = 0; k < FF_PK; ++k) {
#ifdef FF_DIAG_NOLOAD
                nxt[k] = (double)(k + threadIdx.x) + (double)(size_t)row;
#else
                ff_load_async(nxt[k], row - (size_t)(4 * k) * tnum);
#endif
            }
        } else {
#pragma unroll
            for (int k = 0; k < FF_PK; ++k) {
                const int r = p - FF_CH - (4 * k + q);
                nxt[k] = Y[(size_t)(r > 0 ? r : 0) * tnum + j];
            }
        }
        ff_chunk_steps(f, cur, held, q);
        held_p = p;
#ifndef FF_DIAG_NOLOAD
        if (inside) ff_wait_chunk(nxt);
#endif
#pragma unroll
        for (int k = 0; k < FF_PK; ++k) cur[k] = nxt[k];
    }
#pragma unroll
    for (int k = 0; k < FF_PK; ++k) {
        const int r = held_p - (4 * k + q) - edge;
        if (r >= 0 && r < snum) out[(size_t)r * tnum + j] = held[k];
    }
}

#define FIR_ROWS 8   // consecutive output rows per thread

struct FirTaps {
    double t[256 + 2 * FIR_ROWS];   // taps[i] at t[FIR_ROWS + i], zeros on both sides
};

// out[k, j] = sum_i taps[i] * x[k + order - i, j] for k < snum - order (lfilter delayed by `order` rows and
// shifted back, _RadarDataFiltering.py:536-540); `out` is a separate array.  A thread owns FIR_ROWS
// consecutive rows of one trace and walks the order + FIR_ROWS input rows they touch once: one coalesced row
// load and one scalar load of FIR_ROWS consecutive taps per input row, FIR_ROWS fp64 FMAs.
template <typename T>
__global__ __launch_bounds__(256) void fir_shift_kernel(const T *__restrict__ x, T *__restrict__ out, int snum, int tnum,
                                                        int ntaps, FirTaps taps)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int k0 = blockIdx.y * FIR_ROWS;
    const int order = ntaps - 1, nout = snum - order;
    if (j >= tnum) return;
    double acc[FIR_ROWS];
#pragma unroll
    for (int u = 0; u < FIR_ROWS; ++u) acc[u] = 0.0;
    // input row k0 + s contributes to output row k0 + u through tap i = u + order - s
#pragma unroll 4
    for (int s = 0; s < order + FIR_ROWS; ++s) {
        const int r = k0 + s;
        const double xv = r < snum ? (double)x[(size_t)r * tnum + j] : 0.0;
        const double *tp = &taps.t[FIR_ROWS + order - s];
#pragma unroll
        for (int u = 0; u < FIR_ROWS; ++u) acc[u] = __builtin_fma(tp[u], xv, acc[u]);
    }
#pragma unroll
    for (int u = 0; u < FIR_ROWS; ++u)
        if (k0 + u < nout) out[(size_t)(k0 + u) * tnum + j] = (T)acc[u];
}

// out[k, m] = (y_hi - y_lo) / den[m] * t[m] + y_lo   (scipy interp1d._call_linear: the difference is taken in
// the data's own arithmetic, everything after it in fp64)
#define LERP_ROWS 8   // sample rows per thread: the four table entries of a column are loaded once for all of them

template <typename T>
__global__ __launch_bounds__(256) void trace_lerp_kernel(const T *__restrict__ x, double *__restrict__ out, int snum,
                                                         int tnum, int n_new, const int *__restrict__ lo,
                                                         const int *__restrict__ hi, const double *__restrict__ den,
                                                         const double *__restrict__ t)
{
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= n_new) return;
    const int k0 = blockIdx.y * LERP_ROWS;
    const int jl = lo[m], jh = hi[m];
    const double dm = den[m], tm = t[m];
    T ylo[LERP_ROWS], yhi[LERP_ROWS];
#pragma unroll
    for (int u = 0; u < LERP_ROWS; ++u) {
        const int k = k0 + u < snum ? k0 + u : snum - 1;
        ylo[u] = x[(size_t)k * tnum + jl];
        yhi[u] = x[(size_t)k * tnum + jh];
    }
#pragma unroll
    for (int u = 0; u < LERP_ROWS; ++u) {
        if (k0 + u < snum) {
            const double slope = (double)(T)(yhi[u] - ylo[u]) / dm;
            out[(size_t)(k0 + u) * n_new + m] = slope * tm + (double)ylo[u];
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side

struct PreprocScratch {
    impdar_ctx *owner = nullptr;
    DevBuf y, data, aux, idx;
};
static PreprocScratch g_scr;
// one scratch set per process: entry points of different contexts / threads take turns (re-entrant because the
// host-buffer forms call the resident ones)
static std::recursive_mutex g_scr_mu;
#define PREPROC_LOCK() std::lock_guard<std::recursive_mutex> preproc_lock_(g_scr_mu)

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
    PREPROC_LOCK();
    if (g_scr.owner == ctx) {
        g_scr.y.release();
        g_scr.data.release();
        g_scr.aux.release();
        g_scr.idx.release();
        g_scr.owner = nullptr;
    }
}

template <typename T, int NC>
static int filtfilt_launch(impdar_ctx *ctx, T *d, double *Y, int snum, int tnum, int edge, int nc, const FiltCoefs &c)
{
    const int nb = (tnum + FF_TRACES - 1) / FF_TRACES;
    hipLaunchKernelGGL((ff_forward_kernel<T, NC>), dim3(nb), dim3(64), 0, ctx->stream, d, Y, snum, tnum, edge, nc, c);
    hipLaunchKernelGGL((ff_backward_kernel<T, NC>), dim3(nb), dim3(64), 0, ctx->stream, Y, d, snum, tnum, edge, nc, c);
    IMPDAR_HIP_CHECK(hipGetLastError());
    return IMPDAR_OK;
}

template <typename T>
static int filtfilt_dispatch(impdar_ctx *ctx, T *d, double *Y, int snum, int tnum, int edge, int ncoef, const FiltCoefs &c)
{
    if (ncoef <= 5) return filtfilt_launch<T, 5>(ctx, d, Y, snum, tnum, edge, ncoef, c);
    if (ncoef <= 11) return filtfilt_launch<T, 11>(ctx, d, Y, snum, tnum, edge, ncoef, c);
    if (ncoef <= 21) return filtfilt_launch<T, 21>(ctx, d, Y, snum, tnum, edge, ncoef, c);
    return filtfilt_launch<T, FF_MAX_COEF>(ctx, d, Y, snum, tnum, edge, ncoef, c);
}

extern "C" int impdar_filtfilt_dev(impdar_ctx *ctx, void *d_data, int dtype, int snum, int tnum, const double *b,
                                   const double *a, int ncoef, const double *zi)
{
    PREPROC_LOCK();
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
    const int rc = dtype == IMPDAR_F32
                       ? filtfilt_dispatch<float>(ctx, (float *)d_data, g_scr.y.as<double>(), snum, tnum, edge, ncoef, c)
                       : filtfilt_dispatch<double>(ctx, (double *)d_data, g_scr.y.as<double>(), snum, tnum, edge, ncoef, c);
    return rc ? rc : impdar_ctx_mark_produced(ctx);
}

extern "C" int impdar_fir_shift_dev(impdar_ctx *ctx, void *d_data, int dtype, int snum, int tnum, const double *taps,
                                    int ntaps)
{
    PREPROC_LOCK();
    IMPDAR_ARG_CHECK(ctx && d_data && taps, "impdar_fir_shift: null argument");
    IMPDAR_ARG_CHECK(dtype == IMPDAR_F32 || dtype == IMPDAR_F64, "impdar_fir_shift: dtype must be float32 or float64");
    IMPDAR_ARG_CHECK(ntaps >= 1 && ntaps <= 256, "impdar_fir_shift: %d taps (1..256 supported)", ntaps);
    IMPDAR_ARG_CHECK(tnum >= 1 && snum >= 1, "impdar_fir_shift: empty radargram");
    const int order = ntaps - 1;
    if (snum <= order) return IMPDAR_OK;   // data[:-order] is empty: nothing is assigned
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    FirTaps t;
    memset(&t, 0, sizeof(t));
    for (int i = 0; i < ntaps; ++i) t.t[FIR_ROWS + i] = taps[i];
    const size_t es = impdar_dtype_size(dtype), n = (size_t)(snum - order) * tnum;
    scratch_bind(ctx);
    IMPDAR_HIP_CHECK(g_scr.aux.ensure(n * es));
    const dim3 grid((tnum + 255) / 256, (snum - order + FIR_ROWS - 1) / FIR_ROWS);
    if (dtype == IMPDAR_F32)
        hipLaunchKernelGGL(fir_shift_kernel<float>, grid, dim3(256), 0, ctx->stream, (const float *)d_data,
                           g_scr.aux.as<float>(), snum, tnum, ntaps, t);
    else
        hipLaunchKernelGGL(fir_shift_kernel<double>, grid, dim3(256), 0, ctx->stream, (const double *)d_data,
                           g_scr.aux.as<double>(), snum, tnum, ntaps, t);
    IMPDAR_HIP_CHECK(hipGetLastError());
    IMPDAR_HIP_CHECK(hipMemcpyAsync(d_data, g_scr.aux.p, n * es, hipMemcpyDeviceToDevice, ctx->stream));
    return impdar_ctx_mark_produced(ctx);
}

extern "C" int impdar_trace_lerp_dev(impdar_ctx *ctx, const void *d_data, int dtype, int snum, int tnum, const int *lo,
                                     const int *hi, const double *den, const double *t, int n_new, double *d_out)
{
    PREPROC_LOCK();
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
    // the tables are small: one packed synchronous copy keeps the caller's host arrays free to go away (the
    // stream is drained first because the previous launch may still read the table buffer)
    std::vector<char> pack(2 * ib + 2 * db);
    memcpy(pack.data(), den, db);
    memcpy(pack.data() + db, t, db);
    memcpy(pack.data() + 2 * db, lo, ib);
    memcpy(pack.data() + 2 * db + ib, hi, ib);
    IMPDAR_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    IMPDAR_HIP_CHECK(hipMemcpy(base, pack.data(), pack.size(), hipMemcpyHostToDevice));
    const dim3 grid((n_new + 255) / 256, (snum + LERP_ROWS - 1) / LERP_ROWS);
    if (dtype == IMPDAR_F32)
        hipLaunchKernelGGL(trace_lerp_kernel<float>, grid, dim3(256), 0, ctx->stream, (const float *)d_data, d_out, snum,
                           tnum, n_new, d_lo, d_hi, d_den, d_t);
    else
        hipLaunchKernelGGL(trace_lerp_kernel<double>, grid, dim3(256), 0, ctx->stream, (const double *)d_data, d_out, snum,
                           tnum, n_new, d_lo, d_hi, d_den, d_t);
    IMPDAR_HIP_CHECK(hipGetLastError());
    return impdar_ctx_mark_produced(ctx);
}

// element-wise float32 <-> float64 conversion of a resident array (what NumPy's astype does on the host)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const TI *__restrict__ in, TO *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (TO)in[i];
}

extern "C" int impdar_cast_dev(impdar_ctx *ctx, const void *d_src, int src_dtype, void *d_dst, int dst_dtype, size_t n)
{
    IMPDAR_ARG_CHECK(ctx && d_src && d_dst, "impdar_cast_dev: null argument");
    IMPDAR_ARG_CHECK((src_dtype == IMPDAR_F32 || src_dtype == IMPDAR_F64) && (dst_dtype == IMPDAR_F32 || dst_dtype == IMPDAR_F64),
                     "impdar_cast_dev: dtypes must be float32 or float64");
    if (n == 0) return IMPDAR_OK;
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    if (src_dtype == dst_dtype) {
        IMPDAR_HIP_CHECK(hipMemcpyAsync(d_dst, d_src, n * impdar_dtype_size(src_dtype), hipMemcpyDeviceToDevice, ctx->stream));
        return impdar_ctx_mark_produced(ctx);
    }
    const unsigned nb = (unsigned)((n + 255) / 256);
    if (src_dtype == IMPDAR_F64)
        hipLaunchKernelGGL((cast_kernel<double, float>), dim3(nb), dim3(256), 0, ctx->stream, (const double *)d_src, (float *)d_dst, n);
    else
        hipLaunchKernelGGL((cast_kernel<float, double>), dim3(nb), dim3(256), 0, ctx->stream, (const float *)d_src, (double *)d_dst, n);
    IMPDAR_HIP_CHECK(hipGetLastError());
    return impdar_ctx_mark_produced(ctx);
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
    PREPROC_LOCK();
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
    PREPROC_LOCK();
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
    PREPROC_LOCK();
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
