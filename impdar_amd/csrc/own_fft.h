// Batched power-of-two FFTs over contiguous rows, in LDS -- the library's own: what Stolt and the phase shift run their
// transforms on at power-of-two sizes, on every call.
//
// Why.  rocFFT compiles the kernels of a plan at run time for every length its shipped database does not hold (it holds
// lengths up to 1024): a first Stolt call at 4096 x 4096 or phase-shift call at 8192 x 8192 spends 0.25-0.5 s per plan in
// the run-time compiler on a machine that has run before, 2-3 s on a fresh one (profiles/r05_first_call.txt; `impproc
// migrate` is one process per call, so the first call IS the call).  These kernels are part of the library's own code
// object -- nothing to compile, nothing to look up, no plan to make -- and run within 0.03 ms of rocFFT's plans at 8192^2
// (level at Stolt 4096^2), so power-of-two sizes use them on every call (IMPDAR_STOLT_FFT / IMPDAR_PS_FFT = rocfft ask for
// the plans; other sizes keep rocFFT).
//
// Reference semantics (numpy.fft, mig_python.py:159,202,270,282): unnormalised forward transforms with e^{-2 pi i k n / N},
// inverse with e^{+...} and no 1/N (the caller passes the scale, as with the rocFFT plans).
//
// One workgroup per row: the row sits in LDS as complex numbers (index i at i + i / 32: a pad per 32 breaks the
// power-of-two strides), decimation in frequency in place -- radix-4 passes, one radix-2 pass when log2 N is odd -- and the
// digit-reversed order is undone by the final read.  Twiddles from a table e^{-2 pi i k / NT} built by the host in float64
// (W^2j and W^3j by multiplication).  Real rows: N reals are N / 2 complex numbers; the even / odd split is resolved after
// (forward) or before (inverse) the complex transform of length N / 2, in LDS.
#pragma once
#include "common.h"
#include <cmath>
#include <vector>

template <typename T> struct OCp { T x, y; };

enum { OWN_C2C_FWD = 0, OWN_C2C_INV = 1, OWN_R2C = 2, OWN_C2R = 3, OWN_C2C_INV_RE = 4 };     // 4: the inverse's REAL parts only

__host__ __device__ constexpr int own_pad(int i) { return i + (i >> 5); }
// complex length M the kernel supports: a power of two, 16 ... 8192 (float64 rows of 8192 take 132 KB of LDS)
static inline bool own_fft_len_ok(long long m) { return m >= 16 && m <= 8192 && (m & (m - 1)) == 0; }

template <typename T> __device__ __forceinline__ OCp<T> own_mul(OCp<T> a, OCp<T> b)
{
    return OCp<T>{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
template <typename T> __device__ __forceinline__ OCp<T> own_add(OCp<T> a, OCp<T> b) { return OCp<T>{a.x + b.x, a.y + b.y}; }
template <typename T> __device__ __forceinline__ OCp<T> own_sub(OCp<T> a, OCp<T> b) { return OCp<T>{a.x - b.x, a.y - b.y}; }
template <typename T> __device__ __forceinline__ OCp<T> own_conj(OCp<T> a) { return OCp<T>{a.x, -a.y}; }

// position of output index k after the in-place passes: the digits of k, least significant first, are the digits of the
// position, most significant first (radix 4 ..., then 2 when log2 M is odd)
__device__ __forceinline__ int own_rev(int k, int M, int logm)
{
    int p = 0, span = M;
    for (int b = logm; b >= 2; b -= 2) {
        span >>= 2;
        p += (k & 3) * span;
        k >>= 2;
    }
    if (logm & 1) p += (k & 1);         // span is 2 here: the last digit has weight 1
    return p;
}

// The passes of a length-M transform on a row that sits in LDS at s[own_pad(i)] (decimation in frequency, in place, radix 4
// and a last radix 2 when log2 M is odd): the result of index k is left at position own_rev(k).  Every thread of the
// workgroup calls it (barriers inside).  tw: e^{-2 pi i k / (M tws)}.
// one radix-4 decimation-in-frequency butterfly with its twiddles: outputs in the order the passes store them (quarter 0 .. 3)
template <typename T, bool INV>
__device__ __forceinline__ void own_bfly4(OCp<T> &a0, OCp<T> &a1, OCp<T> &a2, OCp<T> &a3, OCp<T> w1)
{
    const OCp<T> t0 = own_add(a0, a2), t1 = own_sub(a0, a2), t2 = own_add(a1, a3), d = own_sub(a1, a3);
    // forward: -i d = (d.y, -d.x); inverse: +i d = (-d.y, d.x)
    const OCp<T> t3 = INV ? OCp<T>{-d.y, d.x} : OCp<T>{d.y, -d.x};
    if (INV) w1.y = -w1.y;
    const OCp<T> w2 = own_mul(w1, w1), w3 = own_mul(w2, w1);
    a0 = own_add(t0, t2);
    a1 = own_mul(own_add(t1, t3), w1);
    a2 = own_mul(own_sub(t0, t2), w2);
    a3 = own_mul(own_sub(t1, t3), w3);
}

template <typename T, bool INV, bool FUSE = false>
__device__ __forceinline__ void own_fft_passes(OCp<T> *s, int M, int logm, int tid, int nth, const OCp<T> *__restrict__ tw, int TWS)
{
    int ll = logm;                      // log2 of the span of the pass
    if constexpr (FUSE && sizeof(T) == 4) {
        // float32 rows of own_fft_rows: TWO radix-4 passes in one trip through LDS while the span allows (round 6) -- a thread
        // takes the 16 points (4 a + b) q + j of a span of 16 q, does the four butterflies of the outer pass (over a, twiddles of
        // the span) and the four of the inner one (over b, twiddles of a quarter span) in registers and stores where two separate
        // passes would have: 8192 points in 4 trips and barriers instead of 7.  Only while every other thread has such a
        // 16-point job: C2C over 8192 x 8192 365 -> 314 us (forward over 4097 rows 190 -> 168), but rows of 4096 points on 1024
        // threads -- a quarter of the threads busy -- ran 3 % SLOWER (Stolt 0.360 -> 0.373 ms), and inside ps_nufft_kernel the 32
        // registers of the points spilled 18 (kernel +1.5 %): those keep radix 4, as does float64 (16 points = 64 registers
        // of a 128 budget).  profiles/r06_transforms.txt
        while (ll >= 4 && (M >> 4) * 2 >= nth) {
            const int lq = ll - 4, q = 1 << lq, L = 1 << ll, tstep = (M >> ll) * TWS;
            for (int b = tid; b < (M >> 4); b += nth) {
                const int g = b >> lq, j = b & (q - 1), base = g * L + j;
                OCp<T> r[4][4];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c) r[a][c] = s[own_pad(base + (4 * a + c) * q)];
                // outer pass: span L, quarter 4 q; butterfly c works on position x = c q + j of the first quarter
#pragma unroll
                for (int c = 0; c < 4; ++c) own_bfly4<T, INV>(r[0][c], r[1][c], r[2][c], r[3][c], tw[(c * q + j) * tstep]);
                // inner pass: span 4 q, quarter q, inside every quarter a of the outer span
                const OCp<T> wi = tw[j * 4 * tstep];
#pragma unroll
                for (int a = 0; a < 4; ++a) own_bfly4<T, INV>(r[a][0], r[a][1], r[a][2], r[a][3], wi);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c) s[own_pad(base + (4 * a + c) * q)] = r[a][c];
            }
            __syncthreads();
            ll -= 4;
        }
    }
    while (ll >= 2) {
        const int lq = ll - 2, q = 1 << lq, L = 1 << ll, tstep = (M >> ll) * TWS;
        for (int b = tid; b < (M >> 2); b += nth) {
            const int g = b >> lq, j = b & (q - 1), base = g * L + j;
            OCp<T> a0 = s[own_pad(base)], a1 = s[own_pad(base + q)], a2 = s[own_pad(base + 2 * q)], a3 = s[own_pad(base + 3 * q)];
            own_bfly4<T, INV>(a0, a1, a2, a3, tw[j * tstep]);
            s[own_pad(base)] = a0;
            s[own_pad(base + q)] = a1;
            s[own_pad(base + 2 * q)] = a2;
            s[own_pad(base + 3 * q)] = a3;
        }
        __syncthreads();
        ll -= 2;
    }
    const int L = 1 << ll;
    if (L == 2) {
        for (int b = tid; b < (M >> 1); b += nth) {
            const OCp<T> a0 = s[own_pad(2 * b)], a1 = s[own_pad(2 * b + 1)];
            s[own_pad(2 * b)] = own_add(a0, a1);
            s[own_pad(2 * b + 1)] = own_sub(a0, a1);
        }
        __syncthreads();
    }
}

// in:  MODE 0/1: [batch][M] complex, rows in_dist COMPLEX elements apart;  MODE 2: [batch][2 M] real, rows in_dist REAL elements
//      apart;  MODE 3: [batch][M + 1] complex, rows in_dist complex elements apart;  MODE 4: as MODE 1
// out: MODE 0/1: [batch][M] complex;  MODE 2: [batch][M + 1] complex;  MODE 3: [batch][2 M] real (out_dist in real elements);
//      MODE 4: [batch][M] real -- Re of the inverse transform (out_dist in real elements): what the phase shift keeps of its
//      inverse over the wavenumbers (mig_python.py:282) without a pass of its own
// tw:  e^{-2 pi i k / NT}, k < NT, NT = M (MODE 0/1) or 2 M (MODE 2/3)
// wcol, wrow (MODE 2, both or neither): float64 weights -- sample j of row r enters as (T)((double) x * (wcol[j] * wrow[r])): the
// phase shift's taper (mig_python.py:258) applied on the way into the transform over the traces; wfirst: (T)(((double) x * wcol[j]) *
// wrow[r]), Stolt's order (:157)
template <typename T, int MODE>
__global__ __launch_bounds__(1024) void own_fft_rows(const void *__restrict__ in_, void *__restrict__ out_, int M, int logm, size_t in_dist,
                                                    size_t out_dist, T scale, const OCp<T> *__restrict__ tw, const double *__restrict__ wcol,
                                                    const double *__restrict__ wrow, int wfirst)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char own_lds[];
    OCp<T> *s = reinterpret_cast<OCp<T> *>(own_lds);
    const int tid = threadIdx.x, nth = blockDim.x;
    const size_t row = blockIdx.x;
    constexpr bool INV = MODE == OWN_C2C_INV || MODE == OWN_C2R || MODE == OWN_C2C_INV_RE;
    constexpr int TWS = (MODE == OWN_R2C || MODE == OWN_C2R) ? 2 : 1;           // stride of the length-M twiddles in the table
    // ---- load
    if (MODE == OWN_C2R) {
        const OCp<T> *X = reinterpret_cast<const OCp<T> *>(in_) + row * in_dist;
        // Z[k] = (X[k] + conj X[M - k]) + i conj(w_k) (X[k] - conj X[M - k]),  w_k = e^{-2 pi i k / 2M}: the spectrum of
        // x[2n] + i x[2n+1], times 2 (so that the result is 2 M x, the unnormalised inverse real transform)
        for (int k = tid; k < M; k += nth) {
            const OCp<T> a = X[k], b = own_conj(X[M - k]);
            const OCp<T> e = own_add(a, b), d = own_sub(a, b);
            const OCp<T> o = own_mul(own_conj(tw[k]), d);
            s[own_pad(k)] = OCp<T>{e.x - o.y, e.y + o.x};      // e + i o
        }
    } else {
        // (MODE 2: the 2 M reals of a row ARE M complex numbers x[2n] + i x[2n+1])
        const OCp<T> *X = MODE == OWN_R2C ? reinterpret_cast<const OCp<T> *>(reinterpret_cast<const T *>(in_) + row * in_dist)
                                          : reinterpret_cast<const OCp<T> *>(in_) + row * in_dist;
        if (MODE == OWN_R2C && wcol) {
            const double wr = wrow[row];
            for (int i = tid; i < M; i += nth) {
                const OCp<T> v = X[i];
                s[own_pad(i)] = wfirst ? OCp<T>{(T)(((double)v.x * wcol[2 * i]) * wr), (T)(((double)v.y * wcol[2 * i + 1]) * wr)}
                                       : OCp<T>{(T)((double)v.x * (wcol[2 * i] * wr)), (T)((double)v.y * (wcol[2 * i + 1] * wr))};
            }
        } else {
            for (int i = tid; i < M; i += nth) s[own_pad(i)] = X[i];
        }
    }
    __syncthreads();
    own_fft_passes<T, INV, true>(s, M, logm, tid, nth, tw, TWS);
    // ---- store (the digit reversal is undone here)
    if (MODE == OWN_R2C) {
        OCp<T> *Y = reinterpret_cast<OCp<T> *>(out_) + row * out_dist;
        // X[k] = (Z[k] + conj Z[M-k]) / 2 - i w_k (Z[k] - conj Z[M-k]) / 2,  k = 0 .. M  (Z[M] = Z[0])
        for (int k = tid; k <= M; k += nth) {
            const OCp<T> zk = s[own_pad(own_rev(k & (M - 1), M, logm))], zm = own_conj(s[own_pad(own_rev((M - k) & (M - 1), M, logm))]);
            const OCp<T> e = own_add(zk, zm), d = own_sub(zk, zm);
            const OCp<T> o = own_mul(tw[k == M ? 0 : k], d);           // w_M = -1, below
            const T sg = k == M ? (T)-1 : (T)1;
            // e/2 - i (w d)/2 = (e.x + sg o.y, e.y - sg o.x) / 2
            Y[k] = OCp<T>{(T)0.5 * (e.x + sg * o.y) * scale, (T)0.5 * (e.y - sg * o.x) * scale};
        }
    } else if (MODE == OWN_C2R) {
        OCp<T> *Y = reinterpret_cast<OCp<T> *>(reinterpret_cast<T *>(out_) + row * out_dist);
        for (int n = tid; n < M; n += nth) {
            const OCp<T> z = s[own_pad(own_rev(n, M, logm))];
            Y[n] = OCp<T>{z.x * scale, z.y * scale};
        }
    } else if (MODE == OWN_C2C_INV_RE) {
        T *Y = reinterpret_cast<T *>(out_) + row * out_dist;
        for (int k = tid; k < M; k += nth) Y[k] = s[own_pad(own_rev(k, M, logm))].x * scale;
    } else {
        OCp<T> *Y = reinterpret_cast<OCp<T> *>(out_) + row * out_dist;
        for (int k = tid; k < M; k += nth) {
            const OCp<T> z = s[own_pad(own_rev(k, M, logm))];
            Y[k] = OCp<T>{z.x * scale, z.y * scale};
        }
    }
}

// complex (rows x cols) -> (cols x rows) through an LDS tile (phaseshift.hip's ps_transpose_c, shared with stolt.hip)
template <typename T, int TS>
__global__ __launch_bounds__(256) void own_transpose_c(const OCp<T> *__restrict__ in, OCp<T> *__restrict__ out, int rows, int cols)
{
    __shared__ OCp<T> tile[TS][TS + 1];
    const int c0 = blockIdx.x * TS, r0 = blockIdx.y * TS;
    const int tx = threadIdx.x % TS, ty = threadIdx.x / TS;
    for (int r = ty; r < TS; r += 256 / TS)
        if (r0 + r < rows && c0 + tx < cols) tile[r][tx] = in[(size_t)(r0 + r) * cols + c0 + tx];
    __syncthreads();
    for (int c = ty; c < TS; c += 256 / TS)
        if (c0 + c < cols && r0 + tx < rows) out[(size_t)(c0 + c) * rows + r0 + tx] = tile[tx][c];
}
template <typename T> static void own_launch_transpose(const void *in, void *out, int rows, int cols, hipStream_t st)
{
    constexpr int TS = sizeof(T) == 4 ? 64 : 32;
    hipLaunchKernelGGL((own_transpose_c<T, TS>), dim3((cols + TS - 1) / TS, (rows + TS - 1) / TS), dim3(256), 0, st,
                       reinterpret_cast<const OCp<T> *>(in), reinterpret_cast<OCp<T> *>(out), rows, cols);
}

// the twiddle table e^{-2 pi i k / nt}, k < nt, of one (length, precision), on the device
struct OwnTwiddles {
    DevBuf buf;
    int nt = 0;
    bool dbl = false;
    template <typename T> int ensure(int n, hipStream_t st)
    {
        if (nt == n && dbl == (sizeof(T) == 8) && buf.p) return IMPDAR_OK;
        std::vector<OCp<T>> h((size_t)n);
        for (int k = 0; k < n; ++k) {
            // exact octant symmetry keeps the table accurate to the last bit where it matters (k = n/4, n/2, ...)
            const long double a = -2.0L * 3.141592653589793238462643383279502884L * (long double)k / (long double)n;
            h[(size_t)k] = OCp<T>{(T)cosl(a), (T)sinl(a)};
        }
        IMPDAR_HIP_CHECK(buf.ensure(h.size() * sizeof(OCp<T>)));
        IMPDAR_HIP_CHECK(hipMemcpyAsync(buf.p, h.data(), h.size() * sizeof(OCp<T>), hipMemcpyHostToDevice, st));
        IMPDAR_HIP_CHECK(hipStreamSynchronize(st));         // (the host vector goes out of scope)
        nt = n;
        dbl = sizeof(T) == 8;
        return IMPDAR_OK;
    }
};

// mode: OWN_*; n = the transform length (complex length for C2C, real length for R2C / C2R); dists in elements of the
// respective side (complex for complex rows, real for real rows)
template <typename T>
static int own_fft_launch(int mode, int n, size_t batch, const void *in, void *out, size_t in_dist, size_t out_dist, double scale,
                          const OwnTwiddles &tw, hipStream_t st, const double *wcol = nullptr, const double *wrow = nullptr, int wfirst = 0)
{
    const int M = (mode == OWN_R2C || mode == OWN_C2R) ? n / 2 : n;
    int logm = 0;
    while ((1 << logm) < M) ++logm;
    if (!own_fft_len_ok(M) || (1 << logm) != M || tw.nt != n || tw.dbl != (sizeof(T) == 8)) {
        impdar_set_error("own_fft_launch: unsupported length %d (mode %d)", n, mode);
        return IMPDAR_ERR_UNSUPPORTED;
    }
    const size_t lds = (size_t)(own_pad(M) + 1) * sizeof(OCp<T>);
#ifndef OWN_T_BIG
#define OWN_T_BIG 1024
#endif
    // (a butterfly waits for its twiddle from the table in global memory: rows of 4096+ points with 256 threads ran 8 dependent
    // butterflies per thread and pass, 3x rocFFT's time at 8192 points -- C2C over 8192 x 8192: 656 us, 1024 threads: 376 us,
    // rocFFT 234 us; the twiddles from two small LDS tables instead (one more rounding) changed nothing more: 355 us)
    const int threads = M >= 4096 ? OWN_T_BIG : (M >= 2048 ? (OWN_T_BIG > 512 ? 512 : OWN_T_BIG) : (M >= 1024 ? 256 : 64));
    const OCp<T> *t = tw.buf.as<OCp<T>>();
#define OWN_LAUNCH(MODE)                                                                                                          \
    do {                                                                                                                          \
        auto k = own_fft_rows<T, MODE>;                                                                                           \
        IMPDAR_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));             \
        hipLaunchKernelGGL(k, dim3((unsigned)batch), dim3(threads), lds, st, in, out, M, logm, in_dist, out_dist, (T)scale, t, wcol, wrow, wfirst); \
    } while (0)
    switch (mode) {
    case OWN_C2C_FWD: OWN_LAUNCH(OWN_C2C_FWD); break;
    case OWN_C2C_INV: OWN_LAUNCH(OWN_C2C_INV); break;
    case OWN_R2C: OWN_LAUNCH(OWN_R2C); break;
    case OWN_C2C_INV_RE: OWN_LAUNCH(OWN_C2C_INV_RE); break;
    default: OWN_LAUNCH(OWN_C2R); break;
    }
#undef OWN_LAUNCH
    IMPDAR_HIP_CHECK(hipGetLastError());
    return IMPDAR_OK;
}
