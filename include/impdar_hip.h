/*
 * impdar_hip.h -- C ABI of the MI355X (gfx950) migration engine.
 *
 * This is the drop-in boundary for the migration hot path of dlilien/ImpDAR
 * (reference paths are relative to the ImpDAR source tree):
 *
 *   - the reference's only native hook is
 *       src/impdar/lib/migrationlib/mig_cython.h:11   (mig_kirch_loop)
 *     bound by src/impdar/lib/migrationlib/_mig_cython.pyx:19-20 and selected
 *     in src/impdar/lib/migrationlib/__init__.py:16-19.  That exact symbol is
 *     exported below.
 *   - Stolt / phase-shift / T-K have no native hook in the reference; their
 *     boundary is the Python function (mig_python.py:126, :211, :290).  The
 *     impdar_* entry points below are what a ctypes binding of those
 *     functions calls.
 *
 * Conventions: plain pointers and sizes only.  All host buffers are
 * caller-owned, C-contiguous, row-major; radargrams have shape (snum, tnum)
 * (a row is one time sample across all traces).  Calls that take HOST buffers
 * are blocking.  Calls that take DEVICE pointers (the *_dev forms, impdar_dev_memset and the
 * impdar_kirch_prep / _allgather / _exchange / _migrate family) only enqueue work: they are ordered
 * among themselves on the device, and impdar_ctx_sync / any download waits for them.  Every call returns 0 on success or a negative
 * impdar_status, never throws or exits.  impdar_last_error() returns a
 * thread-local message for the last failing call.
 */
#ifndef IMPDAR_HIP_H
#define IMPDAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum impdar_status {
    IMPDAR_OK = 0,
    IMPDAR_ERR_ARG = -1,      /* bad argument (maps to ValueError in Python)  */
    IMPDAR_ERR_HIP = -2,      /* HIP runtime failure                           */
    IMPDAR_ERR_FFT = -3,      /* rocFFT failure                                */
    IMPDAR_ERR_COMM = -4,     /* RCCL failure                                  */
    IMPDAR_ERR_NODEV = -5,    /* no usable GPU                                 */
    IMPDAR_ERR_UNSUPPORTED = -6
} impdar_status;

typedef enum impdar_dtype { IMPDAR_F32 = 0, IMPDAR_F64 = 1 } impdar_dtype;

/* Kirchhoff kernel selection.  AUTO: F64 data -> EXACT, F32 data on a uniform
 * (dist, travel_time) grid -> FAST, otherwise EXACT. */
typedef enum impdar_kirch_mode {
    IMPDAR_KIRCH_AUTO = 0,
    IMPDAR_KIRCH_EXACT = 1,   /* per-pair fp64 index math, any geometry        */
    IMPDAR_KIRCH_FAST = 2     /* fp32 LDS-ring kernel, uniform grids only      */
} impdar_kirch_mode;

typedef struct impdar_ctx impdar_ctx;           /* device + stream + workspaces */
typedef struct impdar_kirch_plan impdar_kirch_plan;

/* ---- library / device ------------------------------------------------- */
const char *impdar_last_error(void);
int impdar_device_count(void);
int impdar_ctx_create(int device, impdar_ctx **out);
void impdar_ctx_destroy(impdar_ctx *ctx);
int impdar_ctx_sync(impdar_ctx *ctx);
/* device-side duration (HIP events on the compute stream, ms) of the kernels of the last impdar_stolt[_dev] /
 * impdar_phaseshift[_dev] call on this context; blocks until they have completed */
int impdar_ctx_last_ms(impdar_ctx *ctx, float *ms);
/* ... and of the frequency-sum kernels alone (the rotate-accumulate work of mig_python.py:396-487, without the
 * transforms and transposes around it) of the last impdar_phaseshift[_dev] call */
int impdar_ctx_last_kernel_ms(impdar_ctx *ctx, float *ms);
/* One JSON object about the last migration entry point that ran on this context (impdar_kirchhoff, impdar_stolt,
 * impdar_phaseshift[_ffd], impdar_taper): {"entry", "kernel" (the kernel that did the sums), "device", "kernel_ms",
 * "device_ms", and per entry point e.g. "plan": "new" | "cached", "launches"}.  SURVEY.md section 5 "metrics": the
 * reference prints 'complete in N seconds' only (mig_python.py:121-122,206-207,285-286); the Python entry points add
 * sizes, traces per second and the device count and print the line on stderr when IMPDAR_METRICS is set. */
int impdar_ctx_last_metrics(impdar_ctx *ctx, char *json, size_t cap);
/* raw device-memory plumbing for resident data (bench, multi-GPU) */
int impdar_dev_alloc(impdar_ctx *ctx, size_t bytes, void **dptr);
int impdar_dev_free(impdar_ctx *ctx, void *dptr);
int impdar_dev_upload(impdar_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int impdar_dev_download(impdar_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int impdar_dev_memset(impdar_ctx *ctx, void *dst_dev, int value, size_t bytes);
/* device (float32 or float64, `n` elements) -> a float64 host array: what the reference's migrations hand back
 * whatever the input type (mig_python.py:118, :282).  Goes through the context's pinned staging buffer and
 * converts on several host threads (a single-threaded astype of a fresh 512 MB array costs more than the
 * phase-shift kernel that produced it). */
int impdar_dev_download_f64(impdar_ctx *ctx, double *dst_host, const void *src_dev, int dtype, size_t n);

/* ---- reference-compatible native hook ---------------------------------
 * Replaces mig_cython.h:11.  Same argument meaning as
 * mig_python.py:35 migrationKirchhoffLoop: writes migdata (snum x tnum,
 * row-major, float64) in place.  The reference prototype carries no `data`
 * pointer, so nearfield != 0 cannot be honoured through it (the reference's
 * own defect); it is rejected with a message on stderr and
 * migdata is left untouched.  Uses device 0. */
void mig_kirch_loop(double *migdata, int tnum, int snum, double *dist,
                    double *zs, double *zs2, double *tt_sec, double vel,
                    double *gradD, double max_travel_time, int nearfield);

/* ---- Kirchhoff (mig_python.py:35-123) ---------------------------------
 * One-shot host-buffer form: data (snum,tnum) of `dtype`; gradient
 * coefficients come from impdar's host shim (they restate numpy.gradient's
 * choice of the uniform / non-uniform formula, mig_python.py:93):
 *   grad_uniform != 0: interior (f[k+1]-f[k-1])/(2*grad_h), ends (f1-f0)/grad_h
 *   grad_uniform == 0: interior ga[k]*f[k-1]+gb[k]*f[k]+gc[k]*f[k+1],
 *                      ends (f[1]-f[0])/ga[0], (f[n-1]-f[n-2])/ga[n-1]
 * out is float64 (snum,tnum) like the reference's dat.data. */
int impdar_kirchhoff(impdar_ctx *ctx, const void *data, int dtype, int snum, int tnum,
                     const double *dist_m, const double *tt_sec, double vel, int nearfield,
                     int grad_uniform, double grad_h, const double *ga, const double *gb,
                     const double *gc, int mode, double *out);

/* Resident / sharded form.  A plan owns the trace-major gradient image
 * GT[tnum_pad][snum] (and the data image for the near-field term), the
 * per-sample and per-offset tables and the launch geometry.
 *   prep     : gradient + transpose of the caller's LOCAL column block
 *              d_data (snum x nloc, row-major, leading dimension ld) into
 *              rows [jlo, jlo+nloc) of the image
 *   allgather: RCCL all-gather of the image rows across the communicator
 *              (equal blocks of tnum_pad/nranks traces per rank)
 *   migrate  : diffraction sum for output traces [xlo,xhi) into d_out
 *              (snum x (xhi-xlo), row-major, element type = plan dtype)
 * All of them are asynchronous; call impdar_ctx_sync to wait.  prep and allgather run on
 * the context's producer stream into one of two buffer sets, migrate on its compute stream,
 * so the prep/all-gather of the next radargram overlap the diffraction sum of the current
 * one; the first prep after a migrate starts a new radargram (switches buffer set).  prep waits
 * (on the device) for whatever the *_dev entry points and impdar_dev_memset have enqueued on the
 * compute stream before it, so a resident chain filter -> prep needs no host synchronisation;
 * impdar_dev_upload is blocking. */
int impdar_kirch_plan_create(impdar_ctx *ctx, int dtype, int snum, int tnum,
                             const double *dist_m, const double *tt_sec, double vel,
                             int nearfield, int grad_uniform, double grad_h,
                             const double *ga, const double *gb, const double *gc,
                             int mode, int nranks, impdar_kirch_plan **out);
void impdar_kirch_plan_destroy(impdar_kirch_plan *plan);
int impdar_kirch_plan_mode(const impdar_kirch_plan *plan);      /* resolved mode */
int impdar_kirch_plan_tnum_pad(const impdar_kirch_plan *plan);
/* Position noise of the profile in units of the trace spacing: how far dist[j] - dist[xi] can be from (j - xi) dx
 * (deviation from the fitted grid + rounding of the largest |dist|).  The float64 kernels that weight a pair by its
 * trace offset (ring, tabulated) meet  max(1e-12, 0.1 xnoise)  of the image maximum against mig_python.py:44-60. */
double impdar_kirch_plan_xnoise(const impdar_kirch_plan *plan);
/* which diffraction-sum kernel the plan will launch */
typedef enum impdar_kirch_kernel {
    IMPDAR_KERNEL_EXACT_PAIR = 0,   /* per-pair fp64 arithmetic in the reference's order: any geometry            */
    IMPDAR_KERNEL_EXACT_TAB = 1,    /* fp64 picks/weights tabulated per (sample, |offset|), global-memory gather  */
    IMPDAR_KERNEL_DQUAD = 2,        /* float64 LDS ring (the float64 default on uniform grids)                    */
    IMPDAR_KERNEL_QUAD = 3,         /* float32 LDS ring, ds_read_b128 (the fast path)                             */
    IMPDAR_KERNEL_TAB = 4,          /* float32 LDS ring, trace-major, for steep moveout                           */
    IMPDAR_KERNEL_GEN = 5           /* float32, non-uniform (sorted) dist: picks computed per pair, LDS-staged     */
} impdar_kirch_kernel;
int impdar_kirch_plan_kernel(const impdar_kirch_plan *plan);
int impdar_kirch_prep(impdar_kirch_plan *plan, const void *d_data, int ld, int jlo, int nloc);
int impdar_kirch_allgather(impdar_kirch_plan *plan);
/* Halo form of the exchange ("grouped ncclSend/ncclRecv of halos when H < shard"): this rank sends
 * image rows [slo[i], shi[i]) to rank speer[i] and receives rows [rlo[i], rhi[i]) from rank rpeer[i], all in one
 * RCCL group on the producer stream.  Ranges are whole 8-trace groups (multiples of 8 inside [0, tnum_pad)).
 * impdar_amd/parallel.py (plan_exchange) derives them from the output blocks and the aperture half width. */
int impdar_kirch_exchange(impdar_kirch_plan *plan, int nsend, const int *speer, const int *slo, const int *shi,
                          int nrecv, const int *rpeer, const int *rlo, const int *rhi);
int impdar_kirch_migrate(impdar_kirch_plan *plan, void *d_out, int xlo, int xhi);
/* HIP-event durations (ms) of the last prep / allgather / migrate enqueued on
 * the plan's stream; blocks until they have completed. */
int impdar_kirch_last_ms(impdar_kirch_plan *plan, float *prep_ms, float *gather_ms,
                         float *migrate_ms);
/* same for the step `back` steps before the last one (a step starts at each
 * impdar_kirch_prep; 64 steps of history are kept), so a timed loop can read
 * its per-step kernel durations after the loop without synchronising in it */
int impdar_kirch_history_ms(impdar_kirch_plan *plan, int back, float *prep_ms, float *gather_ms,
                            float *migrate_ms);
/* exact number of in-aperture (output sample, input trace) pairs for output
 * traces [xlo,xhi) under the plan's geometry (uniform grids only; -1 else) */
long long impdar_kirch_count_pairs(const impdar_kirch_plan *plan, int xlo, int xhi);

/* ---- Stolt f-k (mig_python.py:126-208) --------------------------------
 * data (snum,tnum) of dtype already tapered-cast by the caller? NO: the taper
 * (mig_python.py:152-157) runs on the device.  kx has tnum entries, ws has
 * snum/2+1 entries (host shim restates :161-168).  out has 2*(snum/2) rows,
 * same dtype as data. */
int impdar_stolt(impdar_ctx *ctx, const void *data, int dtype, int snum, int tnum,
                 const double *kx, const double *ws, double vel, double htaper,
                 double vtaper, void *out);
/* resident form used by bench/tests: d_data and d_out are device pointers */
int impdar_stolt_dev(impdar_ctx *ctx, const void *d_data, int dtype, int snum, int tnum,
                     const double *kx, const double *ws, double vel, double htaper,
                     double vtaper, void *d_out);

/* ---- the library's own batched power-of-two row transforms (csrc/own_fft.h) ----
 * What the Stolt / phase-shift calls of power-of-two sizes run their transforms on (rocFFT compiles the kernels of
 * lengths above 1024 at run time: 0.25-3 s per plan, profiles/r05_first_call.txt).  numpy.fft conventions
 * (mig_python.py:159, 202, 270, 282), unnormalised, times `scale`.
 * mode 0: complex forward, 1: complex inverse -- d_in / d_out [batch][n] complex (d_out may be d_in);
 * mode 2: real forward -- d_in [batch][n] real, d_out [batch][n/2 + 1] complex;
 * mode 3: real inverse -- d_in [batch][n/2 + 1] complex, d_out [batch][n] real (imaginary parts of the first and last
 * entry are the caller's business: numpy.fft.irfft ignores them);
 * mode 4: complex inverse, real parts only -- d_in [batch][n] complex, d_out [batch][n] real (mig_python.py:282).
 * n a power of two; complex length (n, or n/2 for the real modes) 16 .. 8192.  dtype IMPDAR_F32 / IMPDAR_F64. */
int impdar_fft_rows_dev(impdar_ctx *ctx, int mode, int dtype, int n, int batch, const void *d_in, void *d_out, double scale);

/* ---- phase shift / Gazdag (mig_python.py:211-287, :361-493) ------------
 * vmig_len == 0: constant velocity `vconst`; vmig_len == snum: 1-D v(z).
 * kx has tnum entries, ws has nt entries (two-sided, :268), tt_us has snum
 * entries (microseconds).  out float64/float32 (snum,tnum) = dtype. */
int impdar_phaseshift(impdar_ctx *ctx, const void *data, int dtype, int snum, int tnum,
                      int nt, const double *kx, const double *ws, double dt,
                      const double *tt_us, double vconst, const double *vmig, int vmig_len,
                      double htaper, double vtaper, void *out);

/* resident form: d_data and d_out are device arrays of `dtype` (snum, tnum) */
int impdar_phaseshift_dev(impdar_ctx *ctx, const void *d_data, int dtype, int snum, int tnum,
                          int nt, const double *kx, const double *ws, double dt,
                          const double *tt_us, double vconst, const double *vmig, int vmig_len,
                          double htaper, double vtaper, void *d_out);
/* Phase shift sharded over the wavenumbers (every k is independent in phaseShift, mig_python.py:396-487):
 * a rank calls, on the whole radargram resident on its device,
 *   impdar_phaseshift_tk_dev      -> d_tk [nk][snum] complex: TK (already / snum, :492) of wavenumbers [k0, k0 + nk)
 *   impdar_ps_alltoall_dev        -> d_t2 [tnum][tw] complex: all wavenumbers, its own depth rows (grouped RCCL
 *                                    send/recv; tau_edges / k_edges: nranks + 1 slab edges, the same on every rank)
 *   impdar_phaseshift_finish_dev  -> d_out (tw, tnum) real: ifft over k, real part (:282)
 * (d_tk holds the rows TK[k] themselves.  impdar_phaseshift / impdar_phaseshift_dev, which only need the real part of the inverse
 * transform, may sum rows k and tnum - k as their Hermitian combination (TK[k] + conj TK[tnum - k]) / 2 -- one transform for two
 * rows, csrc/ps_nufft.h, ps_series.h; the image is the same.) */
int impdar_phaseshift_tk_dev(impdar_ctx *ctx, const void *d_data, int dtype, int snum, int tnum, int nt,
                             const double *kx, const double *ws, double dt, const double *tt_us, double vconst,
                             const double *vmig, int vmig_len, double htaper, double vtaper, int k0, int nk, void *d_tk);
int impdar_ps_alltoall_dev(impdar_ctx *ctx, const void *d_tk, int dtype, int snum, int tnum, int nranks, int rank,
                           const int *tau_edges, const int *k_edges, void *d_t2);
int impdar_phaseshift_finish_dev(impdar_ctx *ctx, void *d_t2, int dtype, int tw, int tnum, void *d_out);

/* ---- phase shift, 2-D v(x,z): Fourier finite-difference branch ----------
 * Replaces the `hasattr(vmig[itau], "__len__")` path of phaseShift
 * (mig_python.py:428-432, 448-487) with fourierFiniteDiff (:496-525) and the
 * stencil of Sp_Matr (:528-540), behind migrationPhaseShift (:211-287).
 * float64 only.  data/out: host (snum, tnum) row-major; vmig2d: host
 * (snum, tnum) migration velocities (getVelocityProfile's 3-column output);
 * kx (tnum), ws (nt) as for impdar_phaseshift; dx_mean = mean(trace_int).
 * The (tau, omega) nest is one serial chain in the reference (a single
 * FFX_last); it is executed in that order. */
int impdar_phaseshift_ffd(impdar_ctx *ctx, const double *data, int snum, int tnum, int nt,
                          const double *kx, const double *ws, double dt, const double *tt_us,
                          const double *vmig2d, double dx_mean, double htaper, double vtaper,
                          double *out);

/* ---- taper only (what mtype='tk' does, mig_python.py:330-335) ---------- */
int impdar_taper(impdar_ctx *ctx, void *data_inout, int dtype, int snum, int tnum,
                 double htaper, double vtaper);

/* ---- processing steps in front of a migration ----------
 * The `_dev` forms take device pointers and run on the context's compute
 * stream without synchronising it, so a radargram can stay resident from the
 * first filter to the migrated image; the plain forms take host buffers.
 *
 * impdar_filtfilt: RadarData.vertical_band_pass with an IIR design
 * (_RadarDataFiltering.py:527-535) = scipy.signal.filtfilt(b, a, data, axis=0)
 * cast back to the data's dtype, in place.  b, a: `ncoef` (2..33)
 * coefficients each; zi: ncoef-1 steady-state initial conditions
 * (scipy.signal.lfilter_zi).  Fails with scipy's message when
 * snum <= 3*ncoef.
 * impdar_fir_shift: the FIR branch (:536-540): rows [0, snum-order) become
 * lfilter(taps, 1, data)[order:], the last `order` = ntaps-1 rows are left.
 * impdar_trace_lerp: the data part of RadarData.constant_space
 * (_RadarDataProcessing.py:549-553): out[k, m] = (data[k, hi[m]] -
 * data[k, lo[m]]) / den[m] * t[m] + data[k, lo[m]] (scipy interp1d's slope
 * form), out float64 (snum, n_new); lo/hi/den/t are host arrays. */
int impdar_filtfilt(impdar_ctx *ctx, void *data_inout, int dtype, int snum, int tnum,
                    const double *b, const double *a, int ncoef, const double *zi);
int impdar_filtfilt_dev(impdar_ctx *ctx, void *d_data_inout, int dtype, int snum, int tnum,
                        const double *b, const double *a, int ncoef, const double *zi);
int impdar_fir_shift(impdar_ctx *ctx, void *data_inout, int dtype, int snum, int tnum,
                     const double *taps, int ntaps);
int impdar_fir_shift_dev(impdar_ctx *ctx, void *d_data_inout, int dtype, int snum, int tnum,
                         const double *taps, int ntaps);
int impdar_trace_lerp(impdar_ctx *ctx, const void *data, int dtype, int snum, int tnum,
                      const int *lo, const int *hi, const double *den, const double *t,
                      int n_new, double *out);
int impdar_trace_lerp_dev(impdar_ctx *ctx, const void *d_data, int dtype, int snum, int tnum,
                          const int *lo, const int *hi, const double *den, const double *t,
                          int n_new, double *d_out);

/* float32 <-> float64 conversion of a resident array of `n` elements (NumPy's astype, on the device) */
int impdar_cast_dev(impdar_ctx *ctx, const void *d_src, int src_dtype, void *d_dst, int dst_dtype, size_t n);

/* ---- communicator (RCCL over xGMI) ------------------------------------- */
#define IMPDAR_UNIQUE_ID_BYTES 128
int impdar_comm_unique_id(char id[IMPDAR_UNIQUE_ID_BYTES]);
int impdar_comm_init(impdar_ctx *ctx, const char id[IMPDAR_UNIQUE_ID_BYTES], int rank, int nranks);
int impdar_comm_rank(const impdar_ctx *ctx);
int impdar_comm_size(const impdar_ctx *ctx);
/* RCCL's own view of the communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice / ncclGetVersion); any
 * pointer may be null */
int impdar_comm_info(const impdar_ctx *ctx, int *ranks, int *rank, int *device, int *version);
int impdar_comm_barrier(impdar_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* IMPDAR_HIP_H */
