// Phase-shift (Gazdag) migration on gfx950 and the taper-only "T-K" stub.
//
// Behaviour restated from src/impdar/lib/migrationlib/mig_python.py:
//   :211-287  migrationPhaseShift: taper (in place, d*(H*V)), zero-pad time to
//             nt = 2^ceil(log2 snum), FK = fft2(data,(nt,tnum)), phaseShift,
//             image = ifft_k(TK).real
//   :396-420  constant velocity:  TK[tau,k] = sum_w 1[(v kx/2)^2 < w^2] FK[w,k] e^{i (tau+1) phi},
//             phi = w dt sqrt(1-(v kx/2w)^2)   (recurrence FFK *= e^{i phi} per tau)
//   :438-487  1-D v(z): per tau  FK[w,k] *= e^{i w dt Re sqrt(coss)},
//             coss = 1-(v_tau kx/2w)^2;  FK[w,k] = 0 for good once coss <= thr_tau;
//             TK[tau,k] = sum_w FK[w,k]
//   :490-492  TK /= snum
//   :290-355  migrationTimeWavenumber = taper only
//
// Kernel mapping: one workgroup per wavenumber k.  Each lane keeps the
// complex state of M = nt/BLOCK frequencies in registers and walks tau in
// tiles of TT; the per-tau sum over frequencies is a wavefront
// reduce-scatter (DPP/shuffle butterfly, 2*TT values in ~2*TT shuffles)
// followed by one LDS pass across the waves, so each workgroup emits 128-byte
// contiguous runs of TK_T[k][tau].
#include "fft.h"
#include "own_fft.h"
#include "ps_series_plan.h"
#include <algorithm>
#include <mutex>
#include <type_traits>

template <typename T> struct Cp { T x, y; };


// (snum,tnum) real -> tapered, zero-padded, transposed complex X[tnum][nt]
template <typename T>
__global__ __launch_bounds__(256) void ps_taper_pad_transpose(const T *__restrict__ in, Cp<T> *__restrict__ X, int snum,
                                                              int tnum, int nt, double htaper, double vtaper)
{
    __shared__ T tile[64][65];
    const int k0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int k = k0 + r, j = j0 + tx;
        T v = 0;
        if (k < snum && j < tnum) {
            const double hv = impdar_taper_w(j, tnum, htaper) * impdar_taper_w(k, snum, vtaper);
            v = (T)((double)in[(size_t)k * tnum + j] * hv);        // data *= H*V, :258
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int j = j0 + r, k = k0 + tx;
        if (j < tnum && k < nt) {
            Cp<T> c;
            c.x = tile[tx][r];
            c.y = 0;
            X[(size_t)j * nt + k] = c;
        }
    }
}

// the same, real: X[tnum][nt] for the real-to-complex transform along time of the Hermitian walk
template <typename T>
__global__ __launch_bounds__(256) void ps_taper_pad_transpose_real(const T *__restrict__ in, T *__restrict__ X, int snum, int tnum,
                                                                   int nt, double htaper, double vtaper)
{
    __shared__ T tile[64][65];
    __shared__ double vw[64];
    const int k0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    // (the weights are float64 divisions: the column's once per thread, the tile's 64 rows' once per tile -- two per element
    // until round 6, 141 us at 8192^2 for a pass that moves 0.54 GB)
    const double hw = j0 + tx < tnum ? impdar_taper_w(j0 + tx, tnum, htaper) : 0.0;
    if (threadIdx.x < 64) vw[threadIdx.x] = k0 + (int)threadIdx.x < snum ? impdar_taper_w(k0 + (int)threadIdx.x, snum, vtaper) : 0.0;
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int k = k0 + r, j = j0 + tx;
        T v = 0;
        if (k < snum && j < tnum) {
            const double hv = hw * vw[r];
            v = (T)((double)in[(size_t)k * tnum + j] * hv);        // data *= H*V, :258
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int j = j0 + r, k = k0 + tx;
        if (j < tnum && k < nt) X[(size_t)j * nt + k] = tile[tx][r];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void ps_taper_inplace(T *__restrict__ d, int snum, int tnum, double htaper,
                                                        double vtaper)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)snum * tnum) return;
    const int k = (int)(i / tnum), j = (int)(i % tnum);
    const double hv = impdar_taper_w(j, tnum, htaper) * impdar_taper_w(k, snum, vtaper);
    d[i] = (T)((double)d[i] * hv);
}

// Z[tnum][snum] complex -> out (snum,tnum) real part
// out[c][r] = in[r][c] for a (rows x cols) complex array: the transforms over the traces / wavenumbers run on
// contiguous rows of the transposed array.  rocFFT's own strided plans for the same transform (stride = row length,
// distance 1) take 1.77 ms at 8192 x 8192 complex64 where this transpose + a contiguous plan take 0.25 + 0.23 ms
// (profiles/r03_fft_strided_probe.txt).
template <typename T, int TS>
__global__ __launch_bounds__(256) void ps_transpose_c(const Cp<T> *__restrict__ in, Cp<T> *__restrict__ out, int rows, int cols)
{
    __shared__ Cp<T> tile[TS][TS + 1];
    const int c0 = blockIdx.x * TS, r0 = blockIdx.y * TS;
    const int tx = threadIdx.x % TS, ty = threadIdx.x / TS;
    for (int r = ty; r < TS; r += 256 / TS)
        if (r0 + r < rows && c0 + tx < cols) tile[r][tx] = in[(size_t)(r0 + r) * cols + c0 + tx];
    __syncthreads();
    for (int c = ty; c < TS; c += 256 / TS)
        if (c0 + c < cols && r0 + tx < rows) out[(size_t)(c0 + c) * rows + r0 + tx] = tile[tx][c];
}

template <typename T> static void ps_launch_transpose(const void *in, void *out, int rows, int cols, hipStream_t st)
{
    constexpr int TS = sizeof(T) == 4 ? 64 : 32;
    hipLaunchKernelGGL((ps_transpose_c<T, TS>), dim3((cols + TS - 1) / TS, (rows + TS - 1) / TS), dim3(256), 0, st,
                       reinterpret_cast<const Cp<T> *>(in), reinterpret_cast<Cp<T> *>(out), rows, cols);
}

// Only the REAL part of the inverse transform over the wavenumbers is kept (mig_python.py:282), and Re sum_k TK[k] e^{+i k x} =
// sum_k G[k] e^{+i k x} with G[k] = (TK[k] + conj TK[-k]) / 2 Hermitian: the transpose forms G for k = 0 .. tnum / 2 on its way
// ([k][tau] -> [tau][k], rows hs apart), and a real inverse transform of HALF the complex length does the rest (round 6: the
// transposed array written is half as large, the transform's row in LDS too).
template <typename T, int TS>
__global__ __launch_bounds__(256) void ps_transpose_herm(const Cp<T> *__restrict__ in, Cp<T> *__restrict__ out, int tnum, int snum, int hs)
{
    __shared__ Cp<T> tile[TS][TS + 1];
    const int c0 = blockIdx.x * TS, r0 = blockIdx.y * TS;          // c: tau, r: k
    const int tx = threadIdx.x % TS, ty = threadIdx.x / TS, nh = tnum / 2;
    for (int r = ty; r < TS; r += 256 / TS) {
        const int k = r0 + r;
        if (k <= nh && c0 + tx < snum) {
            const Cp<T> a = in[(size_t)k * snum + c0 + tx], b = in[(size_t)(k ? tnum - k : 0) * snum + c0 + tx];
            tile[r][tx] = Cp<T>{(T)0.5 * (a.x + b.x), (T)0.5 * (a.y - b.y)};
        }
    }
    __syncthreads();
    for (int c = ty; c < TS; c += 256 / TS)
        if (c0 + c < snum && r0 + tx <= nh) out[(size_t)(c0 + c) * hs + r0 + tx] = tile[tx][c];
}

template <typename T> static void ps_launch_transpose_herm(const void *in, void *out, int tnum, int snum, int hs, hipStream_t st)
{
    constexpr int TS = sizeof(T) == 4 ? 64 : 32;
    hipLaunchKernelGGL((ps_transpose_herm<T, TS>), dim3((snum + TS - 1) / TS, (tnum / 2 + TS) / TS), dim3(256), 0, st,
                       reinterpret_cast<const Cp<T> *>(in), reinterpret_cast<Cp<T> *>(out), tnum, snum, hs);
}

// out[c][r] = in[r][c] for r < rows, 0 for rows <= r < rows_out: [snum][tnum/2 + 1] -> [tnum/2 + 1][nt], the time axis zero-padded
// (the spectrum over the traces on its way to the transform over time, P.fhalf)
template <typename T, int TS>
__global__ __launch_bounds__(256) void ps_transpose_pad(const Cp<T> *__restrict__ in, Cp<T> *__restrict__ out, int rows, int cols, int rows_out)
{
    __shared__ Cp<T> tile[TS][TS + 1];
    const int c0 = blockIdx.x * TS, r0 = blockIdx.y * TS;
    const int tx = threadIdx.x % TS, ty = threadIdx.x / TS;
    for (int r = ty; r < TS; r += 256 / TS)
        tile[r][tx] = (r0 + r < rows && c0 + tx < cols) ? in[(size_t)(r0 + r) * cols + c0 + tx] : Cp<T>{(T)0, (T)0};
    __syncthreads();
    for (int c = ty; c < TS; c += 256 / TS)
        if (c0 + c < cols && r0 + tx < rows_out) out[(size_t)(c0 + c) * rows_out + r0 + tx] = tile[tx][c];
}

// out[i] = Re Z[i] (mig_python.py:282 keeps the real part of the inverse transform)
template <typename T>
__global__ __launch_bounds__(256) void ps_real_part(const Cp<T> *__restrict__ Z, T *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = Z[i].x;
}

template <typename T>
__global__ __launch_bounds__(256) void ps_real_transpose(const Cp<T> *__restrict__ Z, T *__restrict__ out, int snum,
                                                         int tnum)
{
    __shared__ T tile[64][65];
    const int k0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int j = j0 + r, k = k0 + tx;
        tile[r][tx] = (j < tnum && k < snum) ? Z[(size_t)j * snum + k].x : (T)0;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int k = k0 + r, j = j0 + tx;
        if (k < snum && j < tnum) out[(size_t)k * tnum + j] = tile[tx][r];
    }
}

template <typename T> __device__ inline void sincos_t(T x, T *s, T *c);
// |x| <= pi here (x = w dt sqrt(coss), |w dt| <= pi): Cody-Waite reduction by
// pi/2 and the classic degree-7/8 minimax polynomials; ~1e-7 absolute error,
// about 20 VALU instructions, no slow path.
template <> __device__ inline void sincos_t<float>(float x, float *s, float *c)
{
    const float q = rintf(x * 0.636619772f);
    float r = fmaf(q, -1.5707963705062866f, x);
    r = fmaf(q, 4.37113900018624283e-8f, r);
    const float r2 = r * r;
    const float sp = fmaf(r * r2, fmaf(r2, fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
    const float cp = fmaf(r2 * r2, fmaf(r2, fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f),
                                        4.166664568298827e-2f), fmaf(r2, -0.5f, 1.0f));
    const int qi = (int)q;
    const float ss = (qi & 1) ? cp : sp;
    const float cc = (qi & 1) ? sp : cp;
    *s = (qi & 2) ? -ss : ss;
    *c = ((qi + 1) & 2) ? -cc : cc;
}
template <> __device__ inline void sincos_t<double>(double x, double *s, double *c) { sincos(x, s, c); }

// float32 recurrences are re-anchored to the original spectrum and the fp64 phase every PS_ANCHOR depth steps.
// Measured at config 5 (8192^2, same box): 64 -> 65.8 / 84.0 ms (const / v(z)) with a spot-wavenumber error of
// 0.8e-6 / 2.1e-6 against the fp64 oracle; 128 -> 63.0 / 80.5 ms, 1.6e-6 / 4.2e-6; 256 -> 62.0 / 78.9 ms,
// 3.1e-6 / 8.2e-6 (the drift is systematic: it doubles with the interval).  128 keeps the error at the scale of
// the fp32 FFTs around the kernel, 50x inside the stated 2e-4.
#ifndef IMPDAR_PS_ANCHOR
#define IMPDAR_PS_ANCHOR 128
#endif
constexpr int PS_ANCHOR = IMPDAR_PS_ANCHOR;

#define PS_STAMP(point)

struct PsParams {
    const void *F;          // [tnum][nt] complex
    void *TK;               // [tnum][snum] complex
    const double *kx;       // [tnum]
    const double *w;        // [nt]   (zero frequency already replaced by 1e-10/dt)
    const double *vz;       // [snum] (v(z) mode)
    const double *thr;      // [snum] (v(z) mode): (tau/tt[-1]/1e6)^2
    double vconst, dt;
    double vtol;            // float32 v(z): relative velocity change below which the phase increments are reused
    const int *sched;       // float32 v(z), ps_vz32_kernel: [snum] 1 where a new constant-velocity run starts
    const int *tsched;      // ... [ceil(ntile/32)] bit t of word t/32 set where 16-step tile t holds such a step
    const int *rowmap;      // ... [tnum] wavenumber of workgroup b (rows holding boundary frequencies first), or null
    const double *sm_step, *sm_tile;   // float32, ps_smooth32_kernel: the tables of ps_smooth_tables
    void *sm_part;          // ps_smooth kernels: [sm_nchunks][nk][snum] complex partial images (more than one chunk)
    int sm_nchunks;
    int sm_m;               // ... and the frequencies per lane of a wave (chunks of 64 sm_m)
    const double *eps;      // float64 v(z), ps_vz64_kernel: [ceil(snum/16)] sum over the tile's steps of v / v_run - 1
    int snum, tnum, nt, vz_mode;
    // Frequency slots a workgroup walks.  Full walk: nf = nt, slot i = row i of F.  Hermitian walk (herm = 1, real
    // radargram): nf = nt/2; slot 0 = the Nyquist row nt/2 (weight 1), slots 1..nt/2-1 = rows 1..nt/2-1 with weight 2
    // (each stands for itself and for its mirror image (-w, -k)); the zero-frequency row is added by ps_dc_kernel.
    // P.w is indexed by SLOT (the host uploads it in slot order).
    int nf, herm;
    int fstride;            // complex elements between the rows of F (nt; nt/2 + 1 when the time transform is real-to-complex)
    int nk;                 // wavenumbers in this launch (tnum, or a rank's slab)
    int k0;                 // first wavenumber of this launch (a rank's slab of a kx-sharded run; 0 otherwise): workgroup b
                            // works on wavenumber k0 + b and writes row b of TK
    int fhalf;              // 1: F is [tnum/2 + 1][nt] -- the wavenumbers k >= 0 with ALL frequencies (rows fstride = nt apart, the
                            // transform over the traces taken first, on the radargram's own rows): FK[tnum - k][w] = conj FK[k][-w].
                            // ps_nufft_kernel's pairs read both of their rows out of one; ps_edge_kernel, ps_dc_kernel know it;
                            // no other kernel is launched on it (ps_run)
};

// Why half of the frequencies are enough for a real radargram (mig_python.py:268-270, 282, 396-420, 438-487):
// FK = fft2(real data) is Hermitian, FK[-w,-k] = conj FK[w,k]; the phase w dt sqrt(1 - (v kx / 2w)^2) is odd in w
// and even in kx, the propagating / evanescent masks are even in both; so the negative-frequency half of
// TK[tau,k] = sum_w alive FK e^{i Phi} equals conj(TK+[tau,-k]) of the positive half TK+, whose inverse transform
// over k is the complex conjugate of ifft_k(TK+).  Only ifft_k(TK).real is kept (:282), hence
//   image = Re ifft_k( 2 TK+  +  TK[w = 0]  +  TK[w = Nyquist] ).
// The w = 0 row (replaced by 1e-10/dt, :400-402) propagates only where kx = 0, with coss = 1: ps_dc_kernel adds
// FK[0,k0] e^{i (tau+1) 1e-10} there.  The host takes this walk only when the axes are exactly antisymmetric
// (kx[-k] = -kx[k], ws[-i] = -ws[i], ws[0] = 0) and the replaced zero frequency is evanescent for every kx != 0.
template <typename T> __device__ __forceinline__ Cp<T> ps_load_slot(const Cp<T> *__restrict__ Frow, const PsParams &P, int slot)
{
    if (!P.herm) return Frow[slot];
    Cp<T> f = Frow[slot == 0 ? (P.nt >> 1) : slot];
    if (slot != 0) {
        f.x *= (T)2;
        f.y *= (T)2;
    }
    return f;
}
// the same for wavenumber k in either layout of F (P.fhalf: the Hermitian walk's slots of row k >= tnum/2 + 1 are the mirrored
// frequencies of row tnum - k, conjugated)
template <typename T> __device__ __forceinline__ Cp<T> ps_load_slot_k(const PsParams &P, int k, int slot)
{
    const Cp<T> *F = reinterpret_cast<const Cp<T> *>(P.F);
    if (!P.fhalf) return ps_load_slot<T>(F + (size_t)k * P.fstride, P, slot);
    const int idx = slot == 0 ? (P.nt >> 1) : slot;
    const bool mirror = 2 * k > P.tnum;
    Cp<T> f = mirror ? F[(size_t)(P.tnum - k) * P.fstride + (P.nt - idx)] : F[(size_t)k * P.fstride + idx];
    if (mirror) f.y = -f.y;
    if (slot != 0) {
        f.x *= (T)2;
        f.y *= (T)2;
    }
    return f;
}

// wave-level reduce-scatter of NV (power of two <= 32) values over the 64 lanes: each
// halving step sends half of the remaining values to the partner lane; once one value is
// left the remaining lane bits are folded with plain butterflies.  On return v[0] of lane
// L holds the total of value index (L >> (6 - log2 NV)) & (NV - 1).
// value of the lane MASK away (lane ^ MASK).  float32: DPP moves inside a 16-lane row (no LDS crossbar round trip:
// ds_bpermute costs a wave ~100 cycles of latency per dependent step, and all waves of the one resident
// workgroup reduce at the same moment, so nothing hides it)
template <typename T, int MASK> __device__ __forceinline__ T lane_xor(T v)
{
    constexpr bool DPP = true;
    auto dpp32 = [](unsigned u) {
        unsigned x;
        if constexpr (MASK == 1) x = __builtin_amdgcn_update_dpp(0u, u, 0xB1, 0xf, 0xf, false);        // quad_perm:[1,0,3,2]
        else if constexpr (MASK == 2) x = __builtin_amdgcn_update_dpp(0u, u, 0x4E, 0xf, 0xf, false);   // quad_perm:[2,3,0,1]
        else if constexpr (MASK == 4) {
            x = __builtin_amdgcn_update_dpp(0u, u, 0x104, 0xf, 0x5, false);    // row_shl:4 into banks 0 and 2 (lane <- lane + 4)
            x = __builtin_amdgcn_update_dpp(x, u, 0x114, 0xf, 0xa, false);     // row_shr:4 into banks 1 and 3 (lane <- lane - 4)
        } else x = __builtin_amdgcn_update_dpp(0u, u, 0x128, 0xf, 0xf, false);                         // row_ror:8
        return x;
    };
    if constexpr (DPP && sizeof(T) == 4 && MASK <= 8) {
        return __uint_as_float(dpp32(__float_as_uint(v)));
    } else if constexpr (DPP && sizeof(T) == 8 && MASK <= 8) {
        // a double moves as its two words
        return __hiloint2double((int)dpp32((unsigned)__double2hiint(v)), (int)dpp32((unsigned)__double2loint(v)));
    } else {
        return __shfl_xor(v, MASK, 64);
    }
}

template <typename T, int HALF, int MASK, int NV>
__device__ __forceinline__ void wrs_halve(T (&v)[NV], int lane)
{
    constexpr bool SWAP = true;
    if constexpr (SWAP && sizeof(T) == 4 && (MASK == 32 || MASK == 16)) {
        // gfx950 v_permlane32_swap / v_permlane16_swap: the upper half (odd rows) of the first operand trades places
        // with the lower half (even rows) of the second -- afterwards every lane holds the value it keeps and the one
        // its partner sent, in the two registers
#pragma unroll
        for (int i = 0; i < HALF; ++i) {
            const unsigned a = __float_as_uint(v[i]), b = __float_as_uint(v[i + HALF]);
            if constexpr (MASK == 32) {
                const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
                v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            } else {
                const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
                v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            }
        }
    } else if constexpr (SWAP && sizeof(T) == 8 && (MASK == 32 || MASK == 16)) {
        // float64: the same swap on the low and the high words
#pragma unroll
        for (int i = 0; i < HALF; ++i) {
            const unsigned alo = (unsigned)__double2loint(v[i]), ahi = (unsigned)__double2hiint(v[i]);
            const unsigned blo = (unsigned)__double2loint(v[i + HALF]), bhi = (unsigned)__double2hiint(v[i + HALF]);
            if constexpr (MASK == 32) {
                const auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
                const auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
                v[i] = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
            } else {
                const auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
                const auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
                v[i] = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
            }
        }
    } else {
        // keep HALF of the 2*HALF live values, hand the other half to the lane MASK away
        const bool up = (lane & MASK) != 0;
#pragma unroll
        for (int i = 0; i < HALF; ++i) {
            const T send = up ? v[i] : v[i + HALF];
            const T keep = up ? v[i + HALF] : v[i];
            v[i] = keep + lane_xor<T, MASK>(send);
        }
    }
}

template <typename T, int NV>
__device__ __forceinline__ void wave_reduce_scatter(T (&v)[NV], int lane)
{
    static_assert(NV == 8 || NV == 32, "instantiated for 8 and 32 values");
    // every index below is a compile-time constant (a runtime-indexed register array is
    // lowered to compare/select chains: ~2000 instructions per call when this was a loop)
    if constexpr (NV == 32) {
        wrs_halve<T, 16, 32>(v, lane);
        wrs_halve<T, 8, 16>(v, lane);
        wrs_halve<T, 4, 8>(v, lane);
        wrs_halve<T, 2, 4>(v, lane);
        wrs_halve<T, 1, 2>(v, lane);
        v[0] += lane_xor<T, 1>(v[0]);
    } else {
        wrs_halve<T, 4, 32>(v, lane);
        wrs_halve<T, 2, 16>(v, lane);
        wrs_halve<T, 1, 8>(v, lane);
        v[0] += lane_xor<T, 4>(v[0]);
        v[0] += lane_xor<T, 2>(v[0]);
        v[0] += lane_xor<T, 1>(v[0]);
    }
}

// Frequency slot of (thread, m) in the vector kernels: PAIRS of owned frequencies are 64 slots apart (a wave's pair covers
// 128 consecutive slots, the waves of a workgroup interleave at that grain).  The kernels skip a pair that is zero in all
// 64 lanes, and the evanescent frequencies are a contiguous band of slots: with the pair's two frequencies a whole
// workgroup width apart (tid + m BLOCK, rounds 1-3) a pair went out only 512 slots after its first half did.
template <int BLOCK, int M> __device__ __forceinline__ int ps_slot(int tid, int m)
{
    if constexpr (M >= 2 && M % 2 == 0) return (m >> 1) * (2 * BLOCK) + ((tid >> 6) << 7) + ((m & 1) << 6) + (tid & 63);
    else return tid + m * BLOCK;
}

template <typename T, int BLOCK, int M, bool VZ>
__global__ __launch_bounds__(BLOCK) void ps_kernel(PsParams P)
{
    // tau per tile: the v(z) update is ~40 instructions per (tau, frequency) and is fully
    // unrolled, so its tile is kept short to bound the code size
    constexpr int PS_TT = VZ ? 4 : 16;
    constexpr int NW = BLOCK / 64;
    constexpr int SH = VZ ? 3 : 1;            // lane >> SH = value index held after the reduce-scatter
    __shared__ T red[2][NW][2 * PS_TT];
    // float32 v(z): the fp64 phase increment of every owned frequency at the current velocity lives in LDS
    // ([m][thread], each thread reads only what it wrote): 2M registers too many, and recomputing it (fp64
    // divide + square root) at every anchor cost as much as the steps in between
    __shared__ double phd_lds[(VZ && sizeof(T) == 4) ? M * BLOCK : 1];
    const int k = P.k0 + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Cp<T> *F = reinterpret_cast<const Cp<T> *>(P.F) + (size_t)k * P.fstride;
    Cp<T> *TK = reinterpret_cast<Cp<T> *>(P.TK) + (size_t)(k - P.k0) * P.snum;
    const double kxk = P.kx[k];

    // float32 data: a multiplicative recurrence in fp32 drifts systematically (the rounded rotation is the
    // same at every depth step of a constant-velocity run: 8192 steps x 6e-8 = 5e-4 .. 2e-3 at config 5),
    // so the accumulated phase is kept in fp64 and the spectrum is rotated from its ORIGINAL value:
    //   constant v: recurrence, re-anchored to FK0 * exp(i tau phi) every PS_ANCHOR depth steps
    //   v(z):       Phi += w dt sqrt(coss) in fp64 (the increment is recomputed in fp64 only when the
    //               velocity changes); F = FK0 * exp(i Phi) at every velocity change and every PS_ANCHOR-th step,
    //               a fixed fp32 rotation per step in between (as for constant v)
    // float64 data keeps the reference's recurrence (rounding 1e-16 per step).
    constexpr bool F32 = sizeof(T) == 4;
    T fr[M], fi[M];          // complex state per owned frequency (F32 v(z): the original spectrum FK0)
    T pa[M], pb[M];          // const-v: (cos phi, sin phi); v(z) fp64: (g = (kx/2w)^2, w*dt); v(z) fp32: (coss, -)
    double phd[F32 && !VZ ? M : 1]; // F32 const v: phase increment per depth step
    constexpr bool FZ = F32 && VZ;
    double Phi[FZ ? M : 1];          // F32 v(z): accumulated phase at the last anchor, kept in [-pi, pi]
    T gr[FZ ? M : 1], gi[FZ ? M : 1];   // F32 v(z): rotated state FK0 * exp(i Phi_tau)
    T pc[FZ ? M : 1], ps[FZ ? M : 1];   // F32 v(z): exp(i increment) of the current velocity
    T f0r[F32 && !VZ ? M : 1], f0i[F32 && !VZ ? M : 1];   // F32 const v: original spectrum for re-anchoring
    constexpr bool FD = VZ && !F32;
    T pw[FD ? M : 1];                 // fp64 v(z): w (pa holds 1/w)
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int iw = ps_slot<BLOCK, M>(tid, m);
        fr[m] = fi[m] = pa[m] = pb[m] = 0;
        if (FD) pw[FD ? m : 0] = 0;
        if (F32 && !VZ) phd[F32 && !VZ ? m : 0] = 0.0;
        if (FZ) {
            Phi[FZ ? m : 0] = 0.0;
            gr[FZ ? m : 0] = gi[FZ ? m : 0] = pc[FZ ? m : 0] = ps[FZ ? m : 0] = 0;
        }
        if (F32 && !VZ) f0r[F32 && !VZ ? m : 0] = f0i[F32 && !VZ ? m : 0] = 0;
        if (iw < P.nf) {
            const Cp<T> f = ps_load_slot<T>(F, P, iw);
            const double w = P.w[iw];
            if (!VZ) {
                const double vk = P.vconst * kxk / 2.0;
                const double vkx2 = vk * vk;                           // :411
                if (vkx2 < w * w) {                                    // :412 propagating only
                    const double ph = w * P.dt * sqrt(1.0 - vkx2 / (w * w));   // :415
                    double s, c;
                    sincos(ph, &s, &c);
                    pa[m] = (T)c;
                    pb[m] = (T)s;
                    fr[m] = f.x;
                    fi[m] = f.y;
                    if (F32) {
                        phd[F32 && !VZ ? m : 0] = ph;
                        f0r[F32 && !VZ ? m : 0] = f.x;
                        f0i[F32 && !VZ ? m : 0] = f.y;
                    }
                }
            } else {
                // coss = 1 - ((0.5 v) kx / w)^2 in the reference's own rounding (:460): a frequency that sits on the
                // evanescent boundary (coss = 0 up to an ulp or two; round-number geometries produce whole families)
                // is kept or dropped by the sign of that rounding.  pa = rn(1/w) and w itself turn the per-step
                // division into three operations with the correctly rounded quotient (below).
                pa[m] = (T)(1.0 / w);
                pb[m] = (T)(w * P.dt);
                pw[FD ? m : 0] = (T)w;
                fr[m] = f.x;
                fi[m] = f.y;
            }
        }
        // one frequency's fp64 set-up at a time (interleaved, the M chains spill)
        asm volatile("" : "+v"(fr[m]), "+v"(fi[m]), "+v"(pa[m]), "+v"(pb[m]));
    }
    // float32 helpers: exp(i x) for a phase kept in fp64, |x| <= pi after the wrap
    auto rot32 = [](double ph, T *s, T *c) { sincos_t<T>((T)ph, s, c); };
    // F32 v(z): fp64 phase increment w dt sqrt(coss) of owned frequency m at velocity v (0 past the end of the axis)
    auto incr = [&](int m, double v, double *cs_out) -> double {
        const int iw = ps_slot<BLOCK, M>(tid, m);
        double inc = 0.0, cs = 1.0;
        if (iw < P.nf) {
            const double w = P.w[iw];
            const double a = 0.5 * v * kxk / w;                 // :456
            cs = 1.0 - a * a;
            inc = w * P.dt * (cs > 0.0 ? sqrt(cs) : 0.0);       // :458-460
        }
        *cs_out = cs;
        return inc;
    };
    double v_prev = -1.0;    // F32 v(z): velocity of the last depth step
    unsigned edge = 0;       // F32 v(z): bit m = frequency m sits on the evanescent boundary (|coss| < 1e-8)
    double vz_next = (FZ && P.vz) ? P.vz[0] : 0.0, thr_next = (FZ && P.thr) ? P.thr[0] : 0.0;   // F32 v(z): prefetched
    int nrec = 0;            // F32 v(z): depth steps since the last anchor (uniform)
    bool rot_valid = false;  // F32 v(z): pc/ps hold exp(i phd) of the current velocity (uniform)

    const int ntile = (P.snum + PS_TT - 1) / PS_TT;
    for (int tile = 0; tile < ntile; ++tile) {
        PS_STAMP(0)
        T acc[2 * PS_TT];
#pragma unroll
        for (int i = 0; i < 2 * PS_TT; ++i) acc[i] = 0;
        const int tau0 = tile * PS_TT;
        // The chains carry no memory dependence, so instruction selection is free to interleave
        // all M x TT of them and the live temporaries spill; threading the state through an empty
        // volatile asm after every (tau, frequency) update pins the order without adding code.
        if (!VZ) {
            if (F32 && tile > 0 && (tile & (PS_ANCHOR / PS_TT - 1)) == 0) {
                // re-anchor: state after tau0 steps = FK0 * exp(i tau0 phi), phase reduced in fp64
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const double x = (double)tau0 * phd[F32 && !VZ ? m : 0];
                    const double r = x - 6.283185307179586 * rint(x * 0.15915494309189535);
                    T sn, cs;
                    rot32(r, &sn, &cs);
                    const T a = f0r[F32 && !VZ ? m : 0], b = f0i[F32 && !VZ ? m : 0];
                    fr[m] = fma(a, cs, -(b * sn));
                    fi[m] = fma(a, sn, b * cs);
                    asm volatile("" : "+v"(fr[m]), "+v"(fi[m]));
                }
            }
            // (pairs of frequencies that are zero in all 64 lanes -- the evanescent band, masked once at the start,
            // :404-412 -- are skipped: see ps_vz32_kernel; exact zeros, the order of the other terms kept)
            constexpr int QG = M >= 2 ? 2 : 1;
#pragma unroll
            for (int m0 = 0; m0 < M; m0 += QG) {
                bool nz = false;
#pragma unroll
                for (int j = 0; j < QG; ++j) nz = nz || fr[m0 + j] != (T)0 || fi[m0 + j] != (T)0;
                if (__builtin_amdgcn_ballot_w64(nz) == 0) continue;     // uniform
#pragma unroll
                for (int t = 0; t < PS_TT; ++t) {
#pragma unroll
                    for (int j = 0; j < QG; ++j) {
                        const int m = m0 + j;
                        const T nr = fma(fr[m], pa[m], -(fi[m] * pb[m]));   // FFK *= cp, :418
                        const T ni = fma(fr[m], pb[m], fi[m] * pa[m]);
                        fr[m] = nr;
                        fi[m] = ni;
                        acc[2 * t] += nr;                                   // TK[itau] += FFK, :420
                        acc[2 * t + 1] += ni;
                        asm volatile("" : "+v"(fr[m]), "+v"(fi[m]), "+v"(acc[2 * t]), "+v"(acc[2 * t + 1]));
                    }
                }
            }
        } else if (F32) {
            // not unrolled: the rare blocks (velocity change, anchor, boundary frequencies) appear once in the
            // code and the per-step sums are filed into acc[] by a uniform index (unrolled four times the body
            // exceeded the unroller's budget and acc[2t] turned into select chains: 2x slower than before)
#pragma unroll 1
            for (int t = 0; t < PS_TT; ++t) {
                const int tau = tau0 + t;
                if (tau < P.snum) {   // uniform
                    // this step's velocity and threshold were requested a step ago (a load-and-wait per step
                    // cost as much as the step)
                    const double vd = vz_next;
                    const T thr = (T)thr_next;
                    const int tn = tau + 1 < P.snum ? tau + 1 : tau;
                    vz_next = P.vz[tn];
                    thr_next = P.thr[tn];
                    const bool changed = fabs(vd - v_prev) > P.vtol * fabs(vd);   // uniform
                    if (changed) {
                        // new velocity (layered profiles get here a few times).  Velocities within 1e-10 of
                        // the last one count as the same: 2*gradient(z(t)) of a layered table is constant
                        // inside a layer up to ~4e-13 of rounding noise, and a 1e-10 velocity error moves the
                        // phase by < 3e-6 rad over 8192 steps (float32 path only).
                        unsigned new_edge = 0;
#pragma unroll
                        for (int m = 0; m < M; ++m) {
                            if (nrec > 0) {   // uniform: the steps since the last anchor turned by the OLD increment
                                const double ph = Phi[FZ ? m : 0] + (double)nrec * phd_lds[FZ ? m * BLOCK + tid : 0];
                                Phi[FZ ? m : 0] = ph - 6.283185307179586 * rint(ph * 0.15915494309189535);
                            }
                            // A frequency on the evanescent boundary (coss = 0 to rounding; with round-number
                            // geometries whole families of (kx, w) sit exactly there) is kept or dropped for
                            // good by the sign of coss, which the reference re-evaluates with every step's
                            // velocity: those (a handful per radargram) are advanced step by step below, in
                            // fp64, and take no part in the shared-increment scheme (increment 0 for them).
                            double cs;
                            const double inc = incr(m, vd, &cs);
                            const bool on_edge = fabs(cs) < 1e-8;
                            phd_lds[FZ ? m * BLOCK + tid : 0] = on_edge ? 0.0 : inc;
                            new_edge |= (on_edge ? 1u : 0u) << m;
                            pa[m] = (T)cs;
                            // evanescent at this velocity: zero from here on (:484-485).  The threshold the
                            // reference compares coss with is (tau/tt_end/1e6)^2 <= 1e-12, so away from the
                            // boundary the test is the sign of coss and is settled once per velocity; the
                            // boundary frequencies are tested step by step below.
                            if (!on_edge && cs <= 0.0) {
                                fr[m] = 0;
                                fi[m] = 0;
                            }
                            asm volatile("" : "+v"(pa[m]), "+v"(fr[m]), "+v"(fi[m]));
                        }
                        v_prev = vd;
                        edge = new_edge;
                        rot_valid = false;
                        nrec = 0;
                    }
                    nrec += 1;
                    if (changed || nrec == PS_ANCHOR) {
                        // anchor: the state from the ORIGINAL spectrum and the fp64 phase
#pragma unroll
                        for (int m = 0; m < M; ++m) {
                            double ph = Phi[FZ ? m : 0] + (double)nrec * phd_lds[FZ ? m * BLOCK + tid : 0];
                            ph -= 6.283185307179586 * rint(ph * 0.15915494309189535);
                            Phi[FZ ? m : 0] = ph;
                            T sn, cs;
                            rot32(ph, &sn, &cs);
                            gr[FZ ? m : 0] = fma(fr[m], cs, -(fi[m] * sn));     // FK0 * exp(i Phi), :464 cumulated
                            gi[FZ ? m : 0] = fma(fr[m], sn, fi[m] * cs);
                            asm volatile("" : "+v"(gr[FZ ? m : 0]), "+v"(gi[FZ ? m : 0]));
                        }
                        nrec = 0;
                    } else {
                        // same velocity as the last step: every frequency turns by its fixed increment; the
                        // fp32 recurrence runs for fewer than PS_ANCHOR steps between anchors (8e-6 rad of drift)
                        if (!rot_valid) {
#pragma unroll
                            for (int m = 0; m < M; ++m) {
                                T sn, cs;
                                rot32(phd_lds[FZ ? m * BLOCK + tid : 0], &sn, &cs);   // |increment| <= |w| dt <= pi
                                pc[FZ ? m : 0] = cs;
                                ps[FZ ? m : 0] = sn;
                                asm volatile("" : "+v"(pc[FZ ? m : 0]), "+v"(ps[FZ ? m : 0]));
                            }
                            rot_valid = true;
                        }
#pragma unroll
                        for (int m = 0; m < M; ++m) {
                            const T a = gr[FZ ? m : 0], b = gi[FZ ? m : 0];
                            gr[FZ ? m : 0] = fma(a, pc[FZ ? m : 0], -(b * ps[FZ ? m : 0]));
                            gi[FZ ? m : 0] = fma(a, ps[FZ ? m : 0], b * pc[FZ ? m : 0]);
                        }
                    }
                    if (edge) {
                        // boundary frequencies: this step's increment from this step's velocity, in fp64
#pragma unroll
                        for (int m = 0; m < M; ++m)
                            if ((edge >> m) & 1u) {
                                const double w = P.w[ps_slot<BLOCK, M>(tid, m)];
                                const double a = 0.5 * vd * kxk / w;
                                const double cs = 1.0 - a * a;
                                pa[m] = (T)cs;
                                double ph = Phi[FZ ? m : 0] + w * P.dt * (cs > 0.0 ? sqrt(cs) : 0.0);
                                ph -= 6.283185307179586 * rint(ph * 0.15915494309189535);
                                Phi[FZ ? m : 0] = ph;
                                T sn, c2;
                                rot32(ph, &sn, &c2);
                                gr[FZ ? m : 0] = fma(fr[m], c2, -(fi[m] * sn));
                                gi[FZ ? m : 0] = fma(fr[m], sn, fi[m] * c2);
                                if (pa[m] <= thr) {                         // :484-485, stays zero afterwards
                                    fr[m] = 0;
                                    fi[m] = 0;
                                    gr[FZ ? m : 0] = 0;
                                    gi[FZ ? m : 0] = 0;
                                }
                            }
                    }
                    // :487, four partial sums per component (one 16-deep dependent chain of adds stalls a
                    // lone wavefront)
                    T pr[4] = {0, 0, 0, 0}, pi[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        pr[m & 3] += gr[FZ ? m : 0];
                        pi[m & 3] += gi[FZ ? m : 0];
                    }
                    const T sr = (pr[0] + pr[1]) + (pr[2] + pr[3]), si = (pi[0] + pi[1]) + (pi[2] + pi[3]);
#pragma unroll
                    for (int u = 0; u < PS_TT; ++u)
                        if (t == u) {   // uniform
                            acc[2 * u] = sr;
                            acc[2 * u + 1] = si;
                        }
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < PS_TT; ++t) {
                const int tau = min(tau0 + t, P.snum - 1);
                const T num = ((T)0.5 * (T)P.vz[tau]) * (T)kxk;         // (0.5 vbg) kx, :460
                const T thr = (T)P.thr[tau];
                const bool live_tau = (tau0 + t) < P.snum;
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    // a = rn(num / w): q = rn(num rn(1/w)), exact remainder by fma, one correction (Markstein);
                    // a padded lane has pa = pw = 0 and gets a = 0
                    const T q = num * pa[m];
                    const T a = fma(fma(-q, pw[FD ? m : 0], num), pa[m], q);
                    const T coss = (T)1 - a * a;                        // :460
                    const T ph = pb[m] * sqrt(coss > 0 ? coss : (T)0);  // :460 (real part of the complex sqrt)
                    T s, c;
                    sincos_t<T>(ph, &s, &c);
                    T nr = fma(fr[m], c, -(fi[m] * s));                 // :464
                    T ni = fma(fr[m], s, fi[m] * c);
                    if (coss <= thr) {                                  // :484-485, stays zero afterwards
                        nr = 0;
                        ni = 0;
                    }
                    if (live_tau) {
                        fr[m] = nr;
                        fi[m] = ni;
                        acc[2 * t] += nr;                               // :487
                        acc[2 * t + 1] += ni;
                    }
                    asm volatile("" : "+v"(fr[m]), "+v"(fi[m]), "+v"(acc[2 * t]), "+v"(acc[2 * t + 1]));
                }
            }
        }
        // ---- sum over frequencies: wave butterfly, then across waves via LDS
        PS_STAMP(1)
        wave_reduce_scatter<T, 2 * PS_TT>(acc, lane);
        T (*buf)[2 * PS_TT] = red[tile & 1];
        if ((lane & ((1 << SH) - 1)) == 0) buf[wave][lane >> SH] = acc[0];
        PS_STAMP(2)
        __syncthreads();
        PS_STAMP(3)
        if (tid < 2 * PS_TT) {
            T s = 0;
#pragma unroll
            for (int q = 0; q < NW; ++q) s += buf[q][tid];
            const int tau = tau0 + (tid >> 1);
            if (tau < P.snum) {
                T *dst = reinterpret_cast<T *>(TK + tau) + (tid & 1);
                *dst = s / (T)P.snum;                                   // TK /= snum, :492
            }
        }
        PS_STAMP(4)
        // red[] is double-buffered: the next tile writes the other buffer and the
        // barrier of that tile orders it against these reads
    }
}

// loads through the scalar cache (uniform address, data read-only for the kernel), load and wait in one statement:
// a split request / wait pair is not safe in C++ -- the compiler is free to copy the destination register between
// the two statements, i.e. before the data has arrived
__device__ __forceinline__ int scalar_load_i32(const int *p)
{
    int v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
__device__ __forceinline__ void scalar_load_2f64(const double *p, const double *q, double *a, double *b)
{
    double x, y;
    asm volatile("s_load_dwordx2 %0, %2, 0x0\n\ts_load_dwordx2 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(x), "=&s"(y)
                 : "s"(p), "s"(q)
                 : "memory");
    *a = x;
    *b = y;
}

// ---------------------------------------------------------------------------
// float32 v(z) (Gazdag), layered profiles: the depth axis is cut into runs of constant velocity (the host
// marks the steps where the velocity moves by more than P.vtol, the same rule ps_kernel applies on the fly),
// and inside a run every frequency turns by a fixed rotation -- the constant-velocity inner loop.  A 16-step
// tile without a run boundary and inside the axis ("quiet") takes the fully unrolled rotate-accumulate body of
// the constant-velocity kernel (6 instructions per (tau, frequency)); every other tile walks its steps one by
// one with the general update (velocity change: fold the pending steps into the fp64 phase, new fp64 increment,
// evanescence settled for the run, mig_python.py:456-485; anchors).  Frequencies on the evanescent boundary
// (|coss| < 1e-8: kept or dropped by the sign of coss at every step's own velocity) take no part in the shared
// rotation: they are walked in fp64, step by step -- inline in the per-step tiles, as a correction to the
// step sums after the unrolled body in quiet tiles.  The state is anchored to FK0 * exp(i Phi) with the fp64
// phase at every run start and at least every PS_ANCHOR steps.
// Registers per owned frequency: the rotating state, the run's rotation and the fp64 phase at the last
// anchor; the original spectrum FK0 and the fp64 increment live in LDS ([m][thread], each thread reads only
// what it wrote) -- 128 KB at 8192 frequencies, one workgroup per CU.
// ---------------------------------------------------------------------------
#ifndef IMPDAR_PS_VZ32_M8_WAVES
#define IMPDAR_PS_VZ32_M8_WAVES 1      // min waves per SIMD asked of the 512 x 8 instantiation (4: two workgroups per CU at 128 VGPRs, with spills)
#endif
template <int BLOCK, int M>
__global__ __launch_bounds__(BLOCK, (BLOCK == 512 && M == 8) ? IMPDAR_PS_VZ32_M8_WAVES : 1) void ps_vz32_kernel(PsParams P)
{
    constexpr int TT = 16;
    constexpr int NW = BLOCK / 64;
    extern __shared__ __attribute__((aligned(16))) char ps_smem[];
    double *phd_lds = reinterpret_cast<double *>(ps_smem);                              // [M][BLOCK] fp64 increment of the run
    float2 *f0_lds = reinterpret_cast<float2 *>(ps_smem + (size_t)M * BLOCK * 8);       // [M][BLOCK] original spectrum
    float(*red)[NW][2 * TT] = reinterpret_cast<float(*)[NW][2 * TT]>(ps_smem + (size_t)M * BLOCK * 16);
    // boundary frequencies in quiet tiles (see below): per-step corrections to the frequency sum, double-buffered over tiles
    float *corr = reinterpret_cast<float *>(ps_smem + (size_t)M * BLOCK * 16 + 2 * NW * 2 * TT * sizeof(float));   // [2][2 * TT]
    const int k = P.rowmap ? P.rowmap[blockIdx.x] : P.k0 + (int)blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Cp<float> *F = reinterpret_cast<const Cp<float> *>(P.F) + (size_t)k * P.fstride;
    Cp<float> *TK = reinterpret_cast<Cp<float> *>(P.TK) + (size_t)(k - P.k0) * P.snum;
    const double kxk = P.kx[k];
    if (tid < 4 * TT) corr[tid] = 0.f;

    float gr[M], gi[M];      // FK0 * exp(i Phi_tau): the rotating state
    float pc[M], ps[M];      // exp(i increment) of the current run
    double Phi[M];           // accumulated phase at the last anchor, kept in [-pi, pi]
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int iw = ps_slot<BLOCK, M>(tid, m);
        float2 f = make_float2(0.f, 0.f);
        if (iw < P.nf) {
            const Cp<float> c = ps_load_slot<float>(F, P, iw);
            f = make_float2(c.x, c.y);
        }
        f0_lds[m * BLOCK + tid] = f;
        phd_lds[m * BLOCK + tid] = 0.0;
        gr[m] = gi[m] = ps[m] = 0.f;
        pc[m] = 1.f;
        Phi[m] = 0.0;
    }
    auto rot32 = [](double ph, float *s, float *c) { sincos_t<float>((float)ph, s, c); };
    auto anchor = [&](int nrec) {
        // the state from the ORIGINAL spectrum and the fp64 phase: nrec steps at the run's increment since the last anchor
#pragma unroll
        for (int m = 0; m < M; ++m) {
            double ph = Phi[m] + (double)nrec * phd_lds[m * BLOCK + tid];
            ph -= 6.283185307179586 * rint(ph * 0.15915494309189535);
            Phi[m] = ph;
            float sn, cs;
            rot32(ph, &sn, &cs);
            const float2 f = f0_lds[m * BLOCK + tid];
            gr[m] = fmaf(f.x, cs, -(f.y * sn));          // FK0 * exp(i Phi), :464 cumulated
            gi[m] = fmaf(f.x, sn, f.y * cs);
            asm volatile("" : "+v"(gr[m]), "+v"(gi[m]));
        }
    };
    unsigned edge = 0;       // bit m = frequency m sits on the evanescent boundary of the current run (|coss| < 1e-8)
    bool wg_edge = false;    // ... some thread of the workgroup has one (uniform)
    int nrec = 0;            // depth steps since the last anchor (uniform)

    const int ntile = (P.snum + TT - 1) / TT;
    // One 16-step tile, instantiated twice: quiet tiles and per-step tiles each get their own loop below.  With both
    // paths in one loop body (the per-step path is ~40 KB of code between the unrolled body and the tile's tail)
    // every part of a quiet tile ran 8-40 % slower -- same instructions, stamps in profiles/r02_ps_stamps.txt:
    // 9140 cycles per tile against 7870 with the per-step code out of the loop (12 % of the kernel at config 5).
    auto do_tile = [&](const int tile, auto quiet_tag) __attribute__((always_inline)) {
        constexpr bool quiet = decltype(quiet_tag)::value;
        float acc[2 * TT];
#pragma unroll
        for (int i = 0; i < 2 * TT; ++i) acc[i] = 0.f;
        const int tau0 = tile * TT;
        PS_STAMP(0)
        if constexpr (quiet) {
            if (nrec + TT > PS_ANCHOR) {
                anchor(nrec);
                nrec = 0;
            }
            // Pairs of frequencies whose state is zero in all 64 lanes of the wave are skipped (round 4): once evanescent,
            // zero for good (:484-485), and the frequencies below v kx / 2 are a contiguous band -- 42 % of the (kx, w)
            // plane at config 5, whole waves' worth of slots.  A skipped term is an exact zero and the order of the
            // others is kept (for every step: frequencies in rising order): the sums are bit for bit what they were.
            constexpr int QG = M >= 2 ? 2 : 1;
#pragma unroll
            for (int m0 = 0; m0 < M; m0 += QG) {
                bool nz = false;
#pragma unroll
                for (int j = 0; j < QG; ++j) nz = nz || gr[m0 + j] != 0.f || gi[m0 + j] != 0.f;
                if (__builtin_amdgcn_ballot_w64(nz) == 0) continue;             // uniform
#pragma unroll
                for (int t = 0; t < TT; ++t) {
#pragma unroll
                    for (int j = 0; j < QG; ++j) {
                        const int m = m0 + j;
                        const float nr = fmaf(gr[m], pc[m], -(gi[m] * ps[m]));     // FK *= exp(i w dt sqrt(coss)), :464
                        const float ni = fmaf(gr[m], ps[m], gi[m] * pc[m]);
                        gr[m] = nr;
                        gi[m] = ni;
                        acc[2 * t] += nr;                                           // TK[itau] += FK, :487
                        acc[2 * t + 1] += ni;
                        asm volatile("" : "+v"(gr[m]), "+v"(gi[m]), "+v"(acc[2 * t]), "+v"(acc[2 * t + 1]));
                    }
                }
            }
            nrec += TT;
        } else {
#pragma unroll 1
            for (int t = 0; t < TT; ++t) {
                const int tau = tau0 + t;
                if (tau >= P.snum) break;      // uniform
                const double vd = P.vz[tau];
                const float thr = (float)P.thr[tau];
                const bool changed = P.sched[tau] != 0;      // uniform
                if (changed) {
                    unsigned new_edge = 0;
                    // opaque copy of the thread index: otherwise the M 64-bit addresses of P.w[...] below are hoisted
                    // out of the tile loop and held in 2M registers for the whole kernel (spills in the tile tail)
                    int tid_here = tid;
                    asm volatile("" : "+v"(tid_here));
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        const int iw = ps_slot<BLOCK, M>(tid_here, m);
                        {
                            // (a frequency that is out -- original spectrum zeroed -- in all 64 lanes has nothing to turn:
                            // the divide, square root and sincos of a run's start are skipped for it, round 4)
                            const float2 f0c = f0_lds[m * BLOCK + tid];
                            if (__builtin_amdgcn_ballot_w64(f0c.x != 0.f || f0c.y != 0.f) == 0) {      // uniform
                                pc[m] = 1.f;
                                ps[m] = 0.f;
                                phd_lds[m * BLOCK + tid] = 0.0;
                                continue;
                            }
                        }
                        if (nrec > 0) {   // the steps since the last anchor turned by the OLD increment
                            const double ph = Phi[m] + (double)nrec * phd_lds[m * BLOCK + tid];
                            Phi[m] = ph - 6.283185307179586 * rint(ph * 0.15915494309189535);
                        }
                        double inc = 0.0, cs = 1.0;
                        if (iw < P.nf) {
                            const double w = P.w[iw];
                            const double a = 0.5 * vd * kxk / w;                 // :456
                            cs = 1.0 - a * a;
                            inc = w * P.dt * (cs > 0.0 ? sqrt(cs) : 0.0);       // :458-460
                        }
                        // a frequency on the evanescent boundary (coss = 0 to rounding) is kept or dropped by the
                        // sign of coss at every step's own velocity: advanced step by step below, in fp64
                        const bool on_edge = fabs(cs) < 1e-8;
                        const double use = on_edge ? 0.0 : inc;
                        phd_lds[m * BLOCK + tid] = use;
                        new_edge |= (on_edge ? 1u : 0u) << m;
                        // evanescent at this velocity: zero from here on (:484-485; away from the boundary the
                        // reference's threshold (tau/tt_end/1e6)^2 <= 1e-12 is the sign of coss)
                        if (!on_edge && cs <= 0.0) f0_lds[m * BLOCK + tid] = make_float2(0.f, 0.f);
                        float sn, c2;
                        rot32(use, &sn, &c2);                                   // |increment| <= |w| dt <= pi
                        pc[m] = c2;
                        ps[m] = sn;
                        asm volatile("" : "+v"(pc[m]), "+v"(ps[m]));
                    }
                    edge = new_edge;
                    wg_edge = __syncthreads_or(new_edge != 0) != 0;
                    nrec = 0;
                }
                nrec += 1;
                if (changed || nrec == PS_ANCHOR) {
                    anchor(nrec);
                    nrec = 0;
                } else {
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        const float a = gr[m], b = gi[m];
                        gr[m] = fmaf(a, pc[m], -(b * ps[m]));
                        gi[m] = fmaf(a, ps[m], b * pc[m]);
                    }
                }
                if (edge) {
                    // boundary frequencies: this step's increment from this step's velocity, in fp64
                    int tid_here = tid;
                    asm volatile("" : "+v"(tid_here));
#pragma unroll
                    for (int m = 0; m < M; ++m)
                        if ((edge >> m) & 1u) {
                            const double w = P.w[ps_slot<BLOCK, M>(tid_here, m)];
                            const double a = 0.5 * vd * kxk / w;
                            const double cs = 1.0 - a * a;
                            double ph = Phi[m] + w * P.dt * (cs > 0.0 ? sqrt(cs) : 0.0);
                            ph -= 6.283185307179586 * rint(ph * 0.15915494309189535);
                            Phi[m] = ph;
                            float sn, c2;
                            rot32(ph, &sn, &c2);
                            float2 f = f0_lds[m * BLOCK + tid];
                            if ((float)cs <= thr) {                             // :484-485, stays zero afterwards
                                f = make_float2(0.f, 0.f);
                                f0_lds[m * BLOCK + tid] = f;
                            }
                            gr[m] = fmaf(f.x, c2, -(f.y * sn));
                            gi[m] = fmaf(f.x, sn, f.y * c2);
                        }
                }
                float pr[4] = {0, 0, 0, 0}, pi[4] = {0, 0, 0, 0};               // :487
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    pr[m & 3] += gr[m];
                    pi[m & 3] += gi[m];
                }
                const float sr = (pr[0] + pr[1]) + (pr[2] + pr[3]), si = (pi[0] + pi[1]) + (pi[2] + pi[3]);
#pragma unroll
                for (int u = 0; u < TT; ++u)
                    if (t == u) {   // uniform
                        acc[2 * u] = sr;
                        acc[2 * u + 1] = si;
                    }
            }
        }
        // ---- sum over frequencies: wave butterfly, then across waves via LDS
        PS_STAMP(1)
        wave_reduce_scatter<float, 2 * TT>(acc, lane);
        float(*buf)[2 * TT] = red[tile & 1];
        if ((lane & 1) == 0) buf[wave][lane >> 1] = acc[0];
        PS_STAMP(2)
        if (quiet && wg_edge && edge) {
            // Boundary frequencies of this thread (a handful per radargram; whole families of (kx, w) with
            // round-number geometries: dx 1 m, dt 10 ns, v 1.68e8 puts one on every 25th wavenumber).  Their
            // shared increment is zero, so the unrolled body carried their state through the tile unchanged and
            // added it to every step's sum; here each is walked through the tile's 16 steps in fp64 at every
            // step's own velocity (kept or dropped by the sign of coss, mig_python.py:456-485, as in the
            // per-step path below) and the difference to the state the body added goes to the step's
            // correction, which the tile's tail adds to the frequency sum.  Placed here, after the reduce-scatter,
            // because the 32 per-step sums are then out of the registers (before it: 35 spilled).  (Sending such workgroups through
            // the per-step path for the whole run cost them 12x: 28 ms of tail at config 5.)
            int tid_here = tid;
            asm volatile("" : "+v"(tid_here));
            // the steps' velocities and thresholds come through the scalar cache (uniform addresses; as vector loads
            // each would stall the walk for a microsecond)
            const double *svz = P.vz + tau0, *sthr = P.thr + tau0;
            float *cr = corr + (tile & 1) * 2 * TT;
            // one copy of the walk, the owned frequency picked by compare-and-select (an unrolled copy per m kept
            // 16 sets of fp64 temporaries alive: 28 spilled registers)
            for (unsigned e = edge; e; e &= e - 1) {
                const int m = __builtin_ctz(e);
                double ph = 0.0;
                float hr = 0.f, hi = 0.f;
#pragma unroll
                for (int q = 0; q < M; ++q)
                    if (q == m) {
                        ph = Phi[q];
                        hr = gr[q];
                        hi = gi[q];
                    }
                const double w = P.w[ps_slot<BLOCK, M>(tid_here, m)];
                float2 f = f0_lds[m * BLOCK + tid];
                float sr = hr, si = hi;
#pragma unroll 1
                for (int t = 0; t < TT; ++t) {
                    double vd, thr_d;
                    scalar_load_2f64(svz + t, sthr + t, &vd, &thr_d);
                    const float thr = (float)thr_d;
                    const double a = 0.5 * vd * kxk / w;
                    const double cs = 1.0 - a * a;
                    ph += w * P.dt * (cs > 0.0 ? sqrt(cs) : 0.0);
                    ph -= 6.283185307179586 * rint(ph * 0.15915494309189535);
                    float sn, c2;
                    rot32(ph, &sn, &c2);
                    if ((float)cs <= thr) f = make_float2(0.f, 0.f);      // :484-485, stays zero afterwards
                    sr = fmaf(f.x, c2, -(f.y * sn));
                    si = fmaf(f.x, sn, f.y * c2);
                    atomicAdd(&cr[2 * t], sr - hr);
                    atomicAdd(&cr[2 * t + 1], si - hi);
                }
                f0_lds[m * BLOCK + tid] = f;
#pragma unroll
                for (int q = 0; q < M; ++q)
                    if (q == m) {
                        Phi[q] = ph;
                        gr[q] = sr;
                        gi[q] = si;
                    }
            }
        }
        __syncthreads();
        PS_STAMP(3)
        if (tid < 2 * TT) {
            float sum = corr[(tile & 1) * 2 * TT + tid];     // boundary frequencies of a quiet tile (zero otherwise)
            corr[(tile & 1) * 2 * TT + tid] = 0.f;           // next written two tiles on, a barrier in between
#pragma unroll
            for (int q = 0; q < NW; ++q) sum += buf[q][tid];
            const int tau = tau0 + (tid >> 1);
            if (tau < P.snum) {
                float *dst = reinterpret_cast<float *>(TK + tau) + (tid & 1);
                *dst = sum / (float)P.snum;                                     // TK /= snum, :492
            }
        }
        PS_STAMP(4)
        // red[] is double-buffered: the next tile writes the other buffer and the barrier of that tile
        // orders it against these reads
    };
    // the tiles' schedule flags come through the scalar cache, 32 tiles per word (as plain loads they are vector
    // loads -- the compiler cannot prove the table read-only -- whose s_waitcnt vmcnt(0) also waits for the write
    // acknowledgement of the sums wave 0 stored a few instructions earlier)
    unsigned dirty_bits = 0;
    auto is_quiet = [&](const int tile) __attribute__((always_inline)) {
        if ((tile & 31) == 0) dirty_bits = (unsigned)scalar_load_i32(P.tsched + (tile >> 5));
        return ((dirty_bits >> (tile & 31)) & 1u) == 0 && tile * TT + TT <= P.snum;      // uniform
    };
    int tile = 0;
    while (tile < ntile) {
        for (; tile < ntile && !is_quiet(tile); ++tile) do_tile(tile, std::false_type());
        for (; tile < ntile && is_quiet(tile); ++tile) do_tile(tile, std::true_type());
    }
}

// ---------------------------------------------------------------------------
// float64 v(z), layered profiles: the float64 counterpart of ps_vz32_kernel (what a float64 .mat file gets).  The
// per-step kernel (ps_kernel<double, ..., true>) evaluates a square root and a sincos for every (tau, frequency):
// 232 ms at 4096^2 where the constant-velocity float64 kernel takes 19 ms.  Here, inside a run of constant velocity
// every frequency turns by the run's rotation (float64: no anchors needed, the recurrence loses ~1e-16 per step
// like the reference's own product), 6 float64 instructions per (tau, frequency) in fully unrolled 16-step tiles.
//  * Velocity noise inside a run (2*gradient(z(t)) of a layered table is constant only to ~4e-13) is not
//    dropped: the host sums eps = v/v_run - 1 over every tile, and at the end of the tile each frequency is turned
//    by exp(i coef eps_sum) (first order: the angle stays below 5e-7), coef = d(increment)/d(ln v) = -w dt a^2 / sqrt(coss) kept in LDS
//    ([m][thread]); the per-step tiles apply it step by step.
//  * Frequencies near the evanescent boundary (|coss| < 1e-6) take no part in the shared rotation: they are walked
//    step by step at every step's own velocity with the reference's rounding of coss (kept or dropped by its sign,
//    mig_python.py:456-485) -- inline in the per-step tiles, as a correction to the step sums in quiet tiles
//    (float64 LDS atomics), exactly as in ps_vz32_kernel.
// ---------------------------------------------------------------------------
template <int BLOCK, int M>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(2))) void ps_vz64_kernel(PsParams P)
{
    constexpr int TT = 16;
    constexpr int NW = BLOCK / 64;
    extern __shared__ __attribute__((aligned(16))) char ps_smem[];
    double *coef = reinterpret_cast<double *>(ps_smem);                                  // [M][BLOCK]
    double(*red)[NW][2 * TT] = reinterpret_cast<double(*)[NW][2 * TT]>(ps_smem + (size_t)M * BLOCK * 8);
    double *corr = reinterpret_cast<double *>(ps_smem + (size_t)M * BLOCK * 8 + 2 * NW * 2 * TT * sizeof(double));   // [2][2 * TT]
    const int k = P.rowmap ? P.rowmap[blockIdx.x] : P.k0 + (int)blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Cp<double> *F = reinterpret_cast<const Cp<double> *>(P.F) + (size_t)k * P.fstride;
    Cp<double> *TK = reinterpret_cast<Cp<double> *>(P.TK) + (size_t)(k - P.k0) * P.snum;
    const double kxk = P.kx[k];
    if (tid < 4 * TT) corr[tid] = 0.0;

    double gr[M], gi[M];     // the field of every owned frequency
    double pc[M], ps[M];     // exp(i increment) of the current run (identity for boundary and dead frequencies)
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int iw = ps_slot<BLOCK, M>(tid, m);
        gr[m] = gi[m] = ps[m] = 0.0;
        pc[m] = 1.0;
        if (iw < P.nf) {
            const Cp<double> c = ps_load_slot<double>(F, P, iw);
            gr[m] = c.x;
            gi[m] = c.y;
        }
        coef[m * BLOCK + tid] = 0.0;
    }
    unsigned edge = 0;       // bit m = frequency m sits near the evanescent boundary of the current run
    bool wg_edge = false;    // ... some thread of the workgroup has one (uniform)
    double v_run = 1.0;      // the current run's velocity (uniform)

    const int ntile = (P.snum + TT - 1) / TT;
    // one boundary frequency, one depth step, the reference's way: FK *= exp(i w dt sqrt(coss)), zero for good once
    // coss <= threshold
    auto edge_step = [&](double w, double vd, double thr, double &sr, double &si) {
        const double a = 0.5 * vd * kxk / w;                                    // :456
        const double cs = 1.0 - a * a;
        const double inc = w * P.dt * (cs > 0.0 ? sqrt(cs) : 0.0);              // :458-460
        double sn, c2;
        sincos(inc, &sn, &c2);
        const double nr = sr * c2 - si * sn, ni = sr * sn + si * c2;           // :464
        sr = nr;
        si = ni;
        if (cs <= thr) {                                                        // :484-485
            sr = 0.0;
            si = 0.0;
        }
    };
    auto do_tile = [&](const int tile, auto quiet_tag) __attribute__((always_inline)) {
        constexpr bool quiet = decltype(quiet_tag)::value;
        double acc[2 * TT];
#pragma unroll
        for (int i = 0; i < 2 * TT; ++i) acc[i] = 0.0;
        const int tau0 = tile * TT;
        if constexpr (quiet) {
            // (pairs of frequencies that are zero in all 64 lanes are skipped: see ps_vz32_kernel)
            constexpr int QG = M >= 2 ? 2 : 1;
#pragma unroll
            for (int m0 = 0; m0 < M; m0 += QG) {
                bool nz = false;
#pragma unroll
                for (int j = 0; j < QG; ++j) nz = nz || gr[m0 + j] != 0.0 || gi[m0 + j] != 0.0;
                if (__builtin_amdgcn_ballot_w64(nz) == 0) continue;             // uniform
#pragma unroll
                for (int t = 0; t < TT; ++t) {
#pragma unroll
                    for (int j = 0; j < QG; ++j) {
                        const int m = m0 + j;
                        const double nr = fma(gr[m], pc[m], -(gi[m] * ps[m]));      // FK *= exp(i w dt sqrt(coss)), :464
                        const double ni = fma(gr[m], ps[m], gi[m] * pc[m]);
                        gr[m] = nr;
                        gi[m] = ni;
                        acc[2 * t] += nr;                                           // TK[itau] += FK, :487
                        acc[2 * t + 1] += ni;
                        asm volatile("" : "+v"(gr[m]), "+v"(gi[m]), "+v"(acc[2 * t]), "+v"(acc[2 * t + 1]));
                    }
                }
            }
        } else {
#pragma unroll 1
            for (int t = 0; t < TT; ++t) {
                const int tau = tau0 + t;
                if (tau >= P.snum) break;      // uniform
                const double vd = P.vz[tau];
                const double thr = P.thr[tau];
                const bool changed = P.sched[tau] != 0;      // uniform
                int tid_here = tid;
                asm volatile("" : "+v"(tid_here));
                if (changed) {
                    unsigned new_edge = 0;
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        const int iw = ps_slot<BLOCK, M>(tid_here, m);
                        double c2 = 1.0, sn = 0.0, cf = 0.0;
                        // (a frequency whose state is zero in all 64 lanes is out for good: nothing to compute for it)
                        const bool wave_out = __builtin_amdgcn_ballot_w64(gr[m] != 0.0 || gi[m] != 0.0) == 0;      // uniform
                        if (iw < P.nf && !wave_out) {
                            const double w = P.w[iw];
                            const double a = 0.5 * vd * kxk / w;                 // :456
                            const double cs = 1.0 - a * a;
                            const bool on_edge = fabs(cs) < 1e-6;
                            new_edge |= (on_edge ? 1u : 0u) << m;
                            if (!on_edge) {
                                if (cs <= 0.0) {
                                    // evanescent at this velocity: zero from here on (:484-485; away from the
                                    // boundary the threshold (tau/tt_end/1e6)^2 <= 1e-12 is the sign of coss)
                                    gr[m] = 0.0;
                                    gi[m] = 0.0;
                                } else {
                                    const double root = sqrt(cs);
                                    sincos(w * P.dt * root, &sn, &c2);          // :458-464
                                    cf = -(w * P.dt) * (a * a) / root;
                                }
                            }
                        }
                        pc[m] = c2;
                        ps[m] = sn;
                        coef[m * BLOCK + tid] = cf;
                        asm volatile("" : "+v"(pc[m]), "+v"(ps[m]), "+v"(gr[m]), "+v"(gi[m]));
                    }
                    edge = new_edge;
                    wg_edge = __syncthreads_or(new_edge != 0) != 0;
                    v_run = vd;
                }
                const double eps = vd / v_run - 1.0;         // this step's velocity against the run's (uniform)
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    double a = gr[m], b = gi[m];
                    const double nr = fma(a, pc[m], -(b * ps[m]));
                    const double ni = fma(a, ps[m], b * pc[m]);
                    a = nr;
                    b = ni;
                    if (eps != 0.0) {                        // uniform
                        const double d = coef[m * BLOCK + tid] * eps;
                        a = fma(-ni, d, nr);
                        b = fma(nr, d, ni);
                    }
                    gr[m] = a;
                    gi[m] = b;
                }
                if (edge) {
                    // boundary frequencies: this step's increment from this step's velocity (their pc/ps are the identity)
#pragma unroll
                    for (int m = 0; m < M; ++m)
                        if ((edge >> m) & 1u) edge_step(P.w[ps_slot<BLOCK, M>(tid_here, m)], vd, thr, gr[m], gi[m]);
                }
                double pr[4] = {0, 0, 0, 0}, pi[4] = {0, 0, 0, 0};              // :487
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    pr[m & 3] += gr[m];
                    pi[m & 3] += gi[m];
                }
                const double sr = (pr[0] + pr[1]) + (pr[2] + pr[3]), si = (pi[0] + pi[1]) + (pi[2] + pi[3]);
#pragma unroll
                for (int u = 0; u < TT; ++u)
                    if (t == u) {   // uniform
                        acc[2 * u] = sr;
                        acc[2 * u + 1] = si;
                    }
            }
        }
        // ---- sum over frequencies: wave butterfly, then across waves via LDS
        wave_reduce_scatter<double, 2 * TT>(acc, lane);
        double(*buf)[2 * TT] = red[tile & 1];
        if ((lane & 1) == 0) buf[wave][lane >> 1] = acc[0];
        if constexpr (quiet) {
            // velocity noise of the tile: F *= exp(i coef eps_sum) = 1 + i d up to d^2 / 2, and d = coef eps_sum stays
            // below 5e-7 even for a frequency next to the boundary band (coef <= 3e3 |w dt|) under a drift that uses
            // the whole run tolerance (eps_sum <= 16 x 1e-11): the dropped term is < 2e-13
            double es, unused;
            scalar_load_2f64(P.eps + tile, P.eps + tile, &es, &unused);
            if (es != 0.0) {     // uniform
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const double d = coef[m * BLOCK + tid] * es;
                    const double a = gr[m], b = gi[m];
                    gr[m] = fma(-b, d, a);
                    gi[m] = fma(a, d, b);
                }
            }
            if (wg_edge && edge) {
                // boundary frequencies of this thread: the unrolled body carried their state through the tile
                // unchanged (identity rotation) and added it to every step's sum; walk the 16 steps and file the
                // difference (see ps_vz32_kernel)
                int tid_here = tid;
                asm volatile("" : "+v"(tid_here));
                const double *svz = P.vz + tau0, *sthr = P.thr + tau0;
                double *cr = corr + (tile & 1) * 2 * TT;
                for (unsigned e = edge; e; e &= e - 1) {
                    const int m = __builtin_ctz(e);
                    double hr = 0.0, hi = 0.0;
#pragma unroll
                    for (int q = 0; q < M; ++q)
                        if (q == m) {
                            hr = gr[q];
                            hi = gi[q];
                        }
                    const double w = P.w[ps_slot<BLOCK, M>(tid_here, m)];
                    double sr = hr, si = hi;
#pragma unroll 1
                    for (int t = 0; t < TT; ++t) {
                        double vd, thr;
                        scalar_load_2f64(svz + t, sthr + t, &vd, &thr);
                        edge_step(w, vd, thr, sr, si);
                        atomicAdd(&cr[2 * t], sr - hr);
                        atomicAdd(&cr[2 * t + 1], si - hi);
                    }
#pragma unroll
                    for (int q = 0; q < M; ++q)
                        if (q == m) {
                            gr[q] = sr;
                            gi[q] = si;
                        }
                }
            }
        }
        __syncthreads();
        if (tid < 2 * TT) {
            double sum = corr[(tile & 1) * 2 * TT + tid];    // boundary frequencies of a quiet tile (zero otherwise)
            corr[(tile & 1) * 2 * TT + tid] = 0.0;           // next written two tiles on, a barrier in between
#pragma unroll
            for (int q = 0; q < NW; ++q) sum += buf[q][tid];
            const int tau = tau0 + (tid >> 1);
            if (tau < P.snum) {
                double *dst = reinterpret_cast<double *>(TK + tau) + (tid & 1);
                *dst = sum / (double)P.snum;                                    // TK /= snum, :492
            }
        }
    };
    unsigned dirty_bits = 0;
    auto is_quiet = [&](const int tile) __attribute__((always_inline)) {
        if ((tile & 31) == 0) dirty_bits = (unsigned)scalar_load_i32(P.tsched + (tile >> 5));
        return ((dirty_bits >> (tile & 31)) & 1u) == 0 && tile * TT + TT <= P.snum;      // uniform
    };
    int tile = 0;
    while (tile < ntile) {
        for (; tile < ntile && !is_quiet(tile); ++tile) do_tile(tile, std::false_type());
        for (; tile < ntile && is_quiet(tile); ++tile) do_tile(tile, std::true_type());
    }
}

// Hermitian walk: the zero-frequency row.  The reference replaces w = 0 by 1e-10/dt (:400-402); that frequency
// propagates only where kx = 0, and there coss = 1 at every velocity, so both branches turn it by w0 dt per depth
// step: TK[tau, k0] += FK[0, k0] e^{i (tau + 1) w0 dt} / snum (:415-420, :456-487, :492).  One thread per depth step,
// after the frequency kernel has stored its sums.
template <typename T>
__global__ __launch_bounds__(256) void ps_dc_kernel(const Cp<T> *__restrict__ F, Cp<T> *__restrict__ TK, int k0, int tk_row,
                                                    int fstride, int snum, double w0dt)
{
    const int tau = blockIdx.x * 256 + threadIdx.x;
    if (tau >= snum) return;
    const Cp<T> f = F[(size_t)k0 * fstride];
    double sn, cs;
    sincos((double)(tau + 1) * w0dt, &sn, &cs);
    const double re = (double)f.x * cs - (double)f.y * sn, im = (double)f.x * sn + (double)f.y * cs;
    Cp<T> *dst = TK + (size_t)tk_row * snum + tau;
    dst->x += (T)(re / (double)snum);
    dst->y += (T)(im / (double)snum);
}

#include "ps_mfma.h"

struct PsPlan {
    OwnTwiddles tw_time, tw_trace;       // the library's own row transforms (own_fft.h): what the first call of a size runs on
    DevBuf d_taper;                      // [tnum + snum] float64 taper weights (the transform over the traces taken first: P.fhalf)
    int own_calls = 0;                   // ... calls of this size that did
    int dtype = -1, snum = 0, tnum = 0, nt = 0;
    const impdar_ctx *owner = nullptr;   // plans and buffers live on this context's device and stream
    FftPlan f_time, f_trace, b_trace;
    // launch order of the runs kernels (order_rows in ps_run) and what it was made from: kept for the next call
    std::vector<double> rm_kx, rm_ws, rm_v;
    int rm_state = -1;                   // -1 none, 0 natural order, 1 d_rowmap holds the order
    bool rows_form = true;               // b_trace / r_trace / b_slab run on contiguous rows of transposed arrays (else rocFFT's strided plans)
    FftPlan r_time, r_trace;             // Hermitian walk: real-to-complex along time (nt/2 + 1 rows), then over the traces
    bool r_ready = false, c_ready = false;
    DevBuf Xr;                           // ... its real input [tnum][nt]
    FftPlan b_slab;                      // kx-sharded run: inverse transform over k of this rank's depth rows
    int slab_key[3] = {-1, -1, -1};
    const impdar_ctx *slab_owner = nullptr;
    DevBuf d_sendbuf, d_slab;            // ... packed blocks of the all-to-all; the transposed slab
    DevBuf d_blocks, d_edge, d_runtab;   // matrix-core path: row-block table; boundary-frequency counts + lists; per-run phases
    DevBuf d_pr_runs, d_pr_stages, d_rw; // many-runs matrix-core path (ps_runs.h): runs, stages, 1 / w
    DevBuf d_mcount;                     // matrix-core paths: MFMA instructions the kernel issued (one 64-bit counter)
    DevBuf d_pn_pieces, d_pn_corr, d_pn_e1;    // transform path (ps_nufft.h): pieces, the window's correction tables, the first-order sums
    SrHostPlan sr_plan;                  // series path (ps_series.h): the pieces of the last velocity profile, and on the device
    DevBuf d_sr_pieces, d_sr_ev;
    bool sr_dev = false;                 // ... the device copies are those of sr_plan
    OwnTwiddles pn_tw[14];               // ... twiddles of the grid lengths 2^l
    std::vector<float> h_pn_corr;        // ... the tables on the host (made once per padded length)
    std::vector<double> h_pn_corr64;
    int pn_corr_off[2][13] = {{-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1}, {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1}};     // [float32, float64]
    int pn_corr_dev = -1;                // which of the two tables the device buffer holds
    double mfma_instructions = -1.0;     // ... of the last call (-1: not a matrix-core call)
    // a (kx, runs) geometry whose boundary-frequency lists overflowed in a matrix-core path: not tried again
    std::vector<double> ovf_kx;
    std::vector<PsMfmaRun> ovf_runs;
    DevBuf X, TK, d_kx, d_w, d_vz, d_thr, d_sched, d_rowmap, d_eps, d_sm, d_part;
    std::vector<double> h_sm_step, h_sm_tile;
    bool b_ready = false;
};
static std::mutex g_ps_mu;
static PsPlan *g_ps_plan = nullptr;

// called by impdar_ctx_destroy: a cached plan must not outlive the stream it was created on
void impdar_ps_forget(const impdar_ctx *ctx)
{
    std::lock_guard<std::mutex> lk(g_ps_mu);
    if (g_ps_plan && (g_ps_plan->owner == ctx || g_ps_plan->slab_owner == ctx)) {      // (the sharded calls' plan and buffers too)
        delete g_ps_plan;
        g_ps_plan = nullptr;
    }
}

// impdar_release_caches: out of device memory somewhere -- drop the cached plan unless a phase-shift call is using it
static thread_local bool t_ps_busy = false;
void impdar_ps_trim()
{
    if (t_ps_busy) return;
    std::unique_lock<std::mutex> lk(g_ps_mu, std::try_to_lock);
    if (lk.owns_lock() && g_ps_plan) {
        delete g_ps_plan;
        g_ps_plan = nullptr;
    }
}

static thread_local const char *t_ps_kernel = "";       // the frequency-sum kernel of the call in progress (metrics)

template <typename T, int BLOCK, int M>
static void ps_launch(const PsParams &P, hipStream_t st)
{
    constexpr size_t vz32_lds = (size_t)M * BLOCK * 16 + 2 * (BLOCK / 64) * 32 * sizeof(float) + 2 * 32 * sizeof(float);
    if constexpr (sizeof(T) == 4 && vz32_lds <= 160 * 1024) {
        if (P.vz_mode && P.sched) {
            auto k = ps_vz32_kernel<BLOCK, M>;
            (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)vz32_lds);
            hipLaunchKernelGGL(k, dim3(P.nk), dim3(BLOCK), vz32_lds, st, P);
            t_ps_kernel = "ps_vz32_kernel";
            return;
        }
    }
    constexpr size_t vz64_lds = (size_t)M * BLOCK * 8 + 2 * (BLOCK / 64) * 32 * sizeof(double) + 2 * 32 * sizeof(double);
    if constexpr (sizeof(T) == 8 && vz64_lds <= 160 * 1024 && M <= 16) {
        if (P.vz_mode && P.sched && P.eps) {
            auto k = ps_vz64_kernel<BLOCK, M>;
            (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)vz64_lds);
            hipLaunchKernelGGL(k, dim3(P.nk), dim3(BLOCK), vz64_lds, st, P);
            t_ps_kernel = "ps_vz64_kernel";
            return;
        }
    }
    t_ps_kernel = P.vz_mode ? "ps_kernel (per step)" : "ps_kernel (constant velocity)";
    if (P.vz_mode)
        hipLaunchKernelGGL((ps_kernel<T, BLOCK, M, true>), dim3(P.nk), dim3(BLOCK), 0, st, P);
    else
        hipLaunchKernelGGL((ps_kernel<T, BLOCK, M, false>), dim3(P.nk), dim3(BLOCK), 0, st, P);
}

#include "ps_smooth.h"      // v(z) that changes at every step: carried square roots and rotations

template <typename T>
static int ps_dispatch(const PsParams &P, hipStream_t st)
{
    const int nt = P.nf;                // frequency slots per wavenumber (half of the padded samples in a Hermitian walk)
    // Workgroup shape.  "deep": 16 frequencies per lane and as few waves as that takes (the reduce-scatter and the
    // tile barrier are a fixed cost per wave and tile, the rotate-accumulate body scales with the frequencies per
    // lane; smaller workgroups also put two or more on a CU, so one's barrier wait overlaps another's body).
    // "wide": 512 threads as soon as there are 512 frequencies (round 1-2 shape).
    // Measured at 8192^2 with the half walk (4096 frequencies per wavenumber; profiles/r03_ps_shapes.txt): float32 constant
    // v 35.0 (wide) / 41.6 (deep) ms, v(z) 43.4 / 44.9; float64 constant v 63.6 / 56.8, v(z) 72.6 / 70.7.
    const bool deep = sizeof(T) == 8;
    if (nt <= 64) ps_launch<T, 64, 1>(P, st);
    else if (nt <= 128) ps_launch<T, 128, 1>(P, st);
    else if (nt <= 256) ps_launch<T, 256, 1>(P, st);
    else if (nt <= 512) ps_launch<T, 512, 1>(P, st);
    else if (nt <= 1024) { if (deep) ps_launch<T, 64, 16>(P, st); else ps_launch<T, 512, 2>(P, st); }
    else if (nt <= 2048) { if (deep) ps_launch<T, 128, 16>(P, st); else ps_launch<T, 512, 4>(P, st); }
    else if (nt <= 4096) { if (deep) ps_launch<T, 256, 16>(P, st); else ps_launch<T, 512, 8>(P, st); }
    else if (nt <= 8192) ps_launch<T, 512, 16>(P, st);     // (1024 x 8 for the float32 v(z) runs kernel: 33 spilled VGPRs, 12 % slower)
    else if (nt <= 16384) ps_launch<T, 512, 32>(P, st);
    else {
        impdar_set_error("phase-shift kernel supports up to 16384 frequencies per wavenumber (got %d)", nt);
        return IMPDAR_ERR_UNSUPPORTED;
    }
    IMPDAR_HIP_CHECK(hipGetLastError());
    return IMPDAR_OK;
}

#include "ps_nufft.h"
#include "ps_series.h"      // a velocity that changes inside a piece: a few transforms with shared nodes + direct sums at the boundary
#include "ps_runs.h"        // many runs of constant velocity: float32 MFMA, phases generated in the kernel

// did a matrix-core path find more boundary frequencies than it lists on this (kx, runs) geometry before?  (ADVICE r4:
// such inputs paid set-up + sums + a blocking read-back on EVERY call, discarded, before the vector kernels ran)
static bool ps_known_overflow(const PsPlan &pl, const double *kx, int tnum, const std::vector<PsMfmaRun> &runs)
{
    if (pl.ovf_kx.size() != (size_t)tnum || pl.ovf_runs.size() != runs.size() || runs.empty()) return false;
    if (memcmp(pl.ovf_kx.data(), kx, (size_t)tnum * 8) != 0) return false;
    for (size_t i = 0; i < runs.size(); ++i)
        if (runs[i].v != pl.ovf_runs[i].v || runs[i].start != pl.ovf_runs[i].start || runs[i].len != pl.ovf_runs[i].len) return false;
    return true;
}
static void ps_note_overflow(PsPlan &pl, const double *kx, int tnum, const std::vector<PsMfmaRun> &runs)
{
    pl.ovf_kx.assign(kx, kx + tnum);
    pl.ovf_runs = runs;
}

// ---- many-runs matrix-core path (ps_runs.h): stages, launch.  Same contract as ps_mfma_run.
static int ps_runs_run(PsPlan &pl, PsParams P, const std::vector<PsMfmaRun> &runs, const double *kx_host, const double *w_host,
                       const double *thr, hipStream_t st, bool *done)
{
    *done = false;
    const int snum = P.snum, tnum = P.tnum;
    if (P.nf % 32 != 0 || P.nf < 256 || snum < 64 || runs.empty()) return IMPDAR_OK;
    for (int i = 0; i < snum; ++i)
        if (!(thr[i] < 1e-10)) return IMPDAR_OK;           // the evanescence test must be the sign of coss off the boundary band
    if (ps_known_overflow(pl, kx_host, tnum, runs)) return IMPDAR_OK;
    // runs -> pieces: long (<= 512 steps, tiles of 8) and single steps
    std::vector<PrRun> pr;
    int nshort_total = 0, nblk_total = 0;
    for (const PsMfmaRun &r : runs) {
        if (r.len <= PR_SHORT_LEN) {
            pr.push_back(PrRun{r.v, r.start, r.len, 1, 0});
            nshort_total += r.len;
            continue;
        }
        const int npiece = (r.len + PR_LONG_MAX - 1) / PR_LONG_MAX;
        for (int i = 0, at = 0; i < npiece; ++i) {
            // pieces of (nearly) equal length, whole tiles except the last
            int len = ((r.len - at) / (npiece - i) + PR_TT - 1) / PR_TT * PR_TT;
            len = std::min(len, r.len - at);
            pr.push_back(PrRun{r.v, r.start + at, len, 0, 0});
            nblk_total += ((len + PR_TT - 1) / PR_TT + PR_ROWS - 1) / PR_ROWS;
            at += len;
        }
    }
    // a record that is mostly single steps (a velocity that changes at nearly every step) is ps_smooth's
    if (nshort_total > snum / 4 || nblk_total == 0) return IMPDAR_OK;
    std::vector<PrStage> stages;
    {
        PrStage cur;
        auto open = [&](int run0) {
            cur = PrStage{};
            cur.run0 = run0;
            for (int i = 0; i < PR_LONGS; ++i) cur.long_run[i] = -1;
        };
        auto close = [&]() {
            if (cur.nruns == 0) return;
            // the single steps go to the wave with the fewest blocks
            int best = 0;
            for (int i = 1; i < PR_LONGS; ++i)
                if (cur.long_nblk[i] < cur.long_nblk[best]) best = i;
            cur.short_wave = best;
            stages.push_back(cur);
        };
        open(0);
        int nlong = 0;
        for (int i = 0; i < (int)pr.size(); ++i) {
            const bool is_long = pr[i].kind == 0;
            // (single steps lead the long run that follows them: the set-up chains their states into its anchor -- a stage
            // that has its four long runs is closed before the next group)
            const bool fits = cur.nruns < PR_STAGE_RUNS && (is_long ? nlong < PR_LONGS : (nlong < PR_LONGS && cur.nshort + pr[i].len <= PR_SROWS));
            if (!fits) {
                close();
                open(i);
                nlong = 0;
            }
            if (is_long) {
                pr[i].slot = nlong;
                cur.long_run[nlong] = i;
                cur.long_nblk[nlong] = ((pr[i].len + PR_TT - 1) / PR_TT + PR_ROWS - 1) / PR_ROWS;
                ++nlong;
            } else {
                pr[i].slot = cur.nshort;
                for (int s_ = 0; s_ < pr[i].len; ++s_) cur.short_tau[cur.nshort++] = pr[i].start + s_;
            }
            ++cur.nruns;
        }
        close();
    }
    impdar_trace("ps_runs: %zu pieces in %zu stages", pr.size(), stages.size());
    const int nparts = (P.nf + PR_PART - 1) / PR_PART;
    const size_t part_bytes = nparts > 1 ? (size_t)nparts * P.nk * snum * sizeof(Cp<float>) : 0;
    std::vector<double> rw((size_t)P.nf);
    for (int i = 0; i < P.nf; ++i) rw[i] = 1.0 / w_host[i];
    if (pl.d_pr_runs.ensure(pr.size() * sizeof(PrRun)) != hipSuccess || pl.d_pr_stages.ensure(stages.size() * sizeof(PrStage)) != hipSuccess ||
        pl.d_rw.ensure(rw.size() * 8) != hipSuccess || pl.d_edge.ensure((size_t)tnum * (1 + PM_EMAX) * sizeof(int)) != hipSuccess ||
        (part_bytes && pl.d_part.ensure(part_bytes) != hipSuccess) || pl.d_mcount.ensure(8) != hipSuccess) {
        (void)hipGetLastError();
        return IMPDAR_OK;                      // no room: the other paths take the call
    }
    IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_pr_runs.p, pr.data(), pr.size() * sizeof(PrRun), hipMemcpyHostToDevice, st));
    IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_pr_stages.p, stages.data(), stages.size() * sizeof(PrStage), hipMemcpyHostToDevice, st));
    IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_rw.p, rw.data(), rw.size() * 8, hipMemcpyHostToDevice, st));
    PrParams Q;
    Q.P = P;
    Q.runs = pl.d_pr_runs.as<PrRun>();
    Q.stages = pl.d_pr_stages.as<PrStage>();
    Q.rw = pl.d_rw.as<double>();
    Q.nstages = (int)stages.size();
    Q.nruns = (int)pr.size();
    Q.nparts = nparts;
    Q.part = pl.d_part.p;
    Q.edge_cnt = pl.d_edge.as<int>();
    Q.edge_list = Q.edge_cnt + tnum;
    Q.mfma_count = pl.d_mcount.as<unsigned long long>();
    IMPDAR_HIP_CHECK(hipMemsetAsync(Q.edge_cnt, 0, (size_t)tnum * sizeof(int), st));
    IMPDAR_HIP_CHECK(hipMemsetAsync(Q.mfma_count, 0, 8, st));
    // the whole wavenumber axis with kx[tnum - k] = -kx[k]: rows k and tnum - k in one workgroup (ps_runs.h); a slab of a
    // kx-sharded run has no mirror rows
    bool pairs = P.k0 == 0 && P.nk == tnum && tnum >= 2;
    for (int k = 1; 2 * k < tnum && pairs; ++k) pairs = kx_host[k] == -kx_host[tnum - k];      // (the Nyquist row of an even axis is its own partner)
    const bool tr = impdar_trace_on();           // tracing: drain the stream between the launches, a line each
    if (tr) {
        (void)hipStreamSynchronize(st);
        impdar_trace("ps_runs: stream idle before the frequency sums");
    }
    if (pairs) {
        IMPDAR_HIP_CHECK(hipFuncSetAttribute((const void *)ps_runs_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pr_lds_bytes(2)));
        hipLaunchKernelGGL(ps_runs_kernel<2>, dim3((unsigned)(tnum / 2 + 1) * nparts), dim3(256), pr_lds_bytes(2), st, Q);
    } else {
        IMPDAR_HIP_CHECK(hipFuncSetAttribute((const void *)ps_runs_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pr_lds_bytes(1)));
        hipLaunchKernelGGL(ps_runs_kernel<1>, dim3((unsigned)P.nk * nparts), dim3(256), pr_lds_bytes(1), st, Q);
    }
    if (tr) {
        (void)hipStreamSynchronize(st);
        impdar_trace("ps_runs: ps_runs_kernel done");
    }
    if (nparts > 1) {
        const size_t n = (size_t)P.nk * snum;
        hipLaunchKernelGGL((ps_smooth_sum_kernel<float>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                           reinterpret_cast<const Cp<float> *>(pl.d_part.p), reinterpret_cast<Cp<float> *>(P.TK), nparts, n, snum);
    }
    if (tr) {
        (void)hipStreamSynchronize(st);
        impdar_trace("ps_runs: partial images summed");
    }
    {
        PsMfmaParams E;                        // (ps_edge_kernel reads P and the lists only)
        E.mfma_count = nullptr;
        E.P = P;
        E.nruns = 0;
        E.pairs = 0;
        E.edge_cnt = Q.edge_cnt;
        E.edge_list = Q.edge_list;
        hipLaunchKernelGGL(ps_edge_kernel_t<float>, dim3(P.nk), dim3(256), 0, st, E);
    }
    IMPDAR_HIP_CHECK(hipGetLastError());
    impdar_trace("ps_runs: buffers ready, tables copied, kernels enqueued");
    IMPDAR_HIP_CHECK(hipStreamSynchronize(st));            // (the host tables must outlive their copies)
    impdar_trace("ps_runs: stream drained");
    std::vector<int> cnt((size_t)P.nk);
    IMPDAR_HIP_CHECK(hipMemcpy(cnt.data(), Q.edge_cnt + P.k0, cnt.size() * sizeof(int), hipMemcpyDeviceToHost));
    int worst = 0;
    for (int c : cnt) worst = std::max(worst, c);
    // every part of a wavenumber counts its own boundary frequencies into the same list: the first PM_EMAX are walked
    if (worst > PM_EMAX) {
        ps_note_overflow(pl, kx_host, tnum, runs);
        return IMPDAR_OK;                      // contributions missing from TK: discarded, another path produces the result
    }
    {
        unsigned long long n = 0;
        IMPDAR_HIP_CHECK(hipMemcpy(&n, Q.mfma_count, 8, hipMemcpyDeviceToHost));
        pl.mfma_instructions = (double)n;
    }
    impdar_trace("ps_runs: counters read back: %.0f MFMA instructions, worst edge count %d, %d parts, pairs %d", pl.mfma_instructions, worst, nparts, (int)pairs);
    *done = true;
    return IMPDAR_OK;
}

// 1 / psihat(n), n = 0 .. Lp/2, of every padded length in `need`: psihat(n) = int psi(x) cos(2 pi n x / G) dx over |x| < W/2
// (Simpson, float64; the integrand ends at e^{-beta} = 1e-8 of its maximum).  Made once per length and plan; *grew: the device
// copy is stale.  (Shared by the transform paths ps_nufft.h and ps_series.h: same window, same tables.)
template <typename T>
static std::vector<T> &pn_corr_tables(PsPlan &pl, const bool (&need)[13], int (&off_out)[13], bool *grew)
{
    constexpr int PN_W = PnCfg<T>::W;
    std::vector<T> &corr = sizeof(T) == 4 ? reinterpret_cast<std::vector<T> &>(pl.h_pn_corr) : reinterpret_cast<std::vector<T> &>(pl.h_pn_corr64);
    int *corr_off = pl.pn_corr_off[sizeof(T) == 8 ? 1 : 0];
    for (int l = 0; l < 13; ++l) {
        if (!need[l] || corr_off[l] >= 0) continue;
        corr_off[l] = (int)corr.size();
        *grew = true;
        const int Lp = 1 << l, G = 2 * Lp, NS = 512;
        const double beta = 2.30 * PN_W, h = (double)PN_W / NS;
        std::vector<double> psi((size_t)NS + 1);
        for (int q = 0; q <= NS; ++q) {
            const double x = -0.5 * PN_W + q * h, z = 1.0 - (2.0 * x / PN_W) * (2.0 * x / PN_W);
            psi[q] = std::exp(beta * (std::sqrt(z > 0.0 ? z : 0.0) - 1.0)) * ((q == 0 || q == NS) ? 1.0 : ((q & 1) ? 4.0 : 2.0));
        }
        // (sum_q psi_q cos(n theta_q), theta_q = 2 pi x_q / G, for all n at once: per node a rotation by theta_q from n to n + 1,
        // re-seeded with the library's cos / sin every 64 -- a cos() per (n, q) was ~20 ms of a process's first call at 8192^2)
        std::vector<double> acc((size_t)Lp / 2 + 1, 0.0);
        for (int q = 0; q <= NS; ++q) {
            const double th = 6.283185307179586 * (-0.5 * PN_W + q * h) / G, ct = std::cos(th), st_ = std::sin(th), pq = psi[q];
            double c = 1.0, sn = 0.0;
            for (int n = 0; n <= Lp / 2; ++n) {
                if ((n & 63) == 0) {
                    c = std::cos(th * n);
                    sn = std::sin(th * n);
                }
                acc[(size_t)n] += pq * c;
                const double c2 = c * ct - sn * st_;
                sn = sn * ct + c * st_;
                c = c2;
            }
        }
        for (int n = 0; n <= Lp / 2; ++n) corr.push_back((T)(1.0 / (acc[(size_t)n] * h / 3.0)));
    }
    for (int l = 0; l < 13; ++l) off_out[l] = corr_off[l] < 0 ? 0 : corr_off[l];
    return corr;
}

// ---- transform path (ps_nufft.h): pieces, correction tables, launch.  Same contract as ps_mfma_run.
template <typename T>
static int ps_nufft_run(PsPlan &pl, PsParams P, const std::vector<PsMfmaRun> &runs, bool vz, const double *kx_host, const double *w_host,
                        const double *thr, hipStream_t st, bool *done, const double *vmig = nullptr, bool allow_pairs = false)
{
    // vmig (float64 data, v(z)): the per-step velocities -- the runs' rounding noise enters as a first-order term (ps_nufft.h)
    // allow_pairs: the sums go on into the inverse transform (not to a caller who asked for the rows TK themselves): rows k and
    // tnum - k may be written as their Hermitian combination and its conjugate (ps_nufft_kernel<float, true>)
    const bool first_order = sizeof(T) == 8 && vz && vmig != nullptr;
    *done = false;
    const int snum = P.snum, tnum = P.tnum, nf = P.nf;
    if (!P.herm || nf < 64 || nf > PN_NFMAX || snum < 64 || runs.empty()) return IMPDAR_OK;
    if (vz && ps_known_overflow(pl, kx_host, tnum, runs)) return IMPDAR_OK;
    if (vz)
        for (int i = 0; i < snum; ++i)
            if (!(thr[i] < 1e-10)) return IMPDAR_OK;       // the evanescence test must be the sign of coss off the boundary band
    // the gather inverts the dispersion relation on a uniform frequency axis: slot i + 1 at (i + 1) dw, the Nyquist row (slot 0) at nf dw
    const double dw = w_host[1];
    if (!(dw > 0.0)) return IMPDAR_OK;
    for (int i = 1; i < nf; ++i)
        if (std::fabs(w_host[i] - (double)i * dw) > 1e-9 * (double)i * dw) return IMPDAR_OK;
    if (std::fabs(std::fabs(w_host[0]) - (double)nf * dw) > 1e-9 * (double)nf * dw) return IMPDAR_OK;
    // the whole wavenumber axis with kx[tnum - k] = -kx[k]: a pair of rows per workgroup, as ps_runs_kernel's
    bool pairs = allow_pairs && P.k0 == 0 && P.nk == tnum && tnum >= 2 && tnum % 2 == 0;
    for (int k = 1; 2 * k < tnum && pairs; ++k) pairs = kx_host[k] == -kx_host[tnum - k];
    if (P.fhalf && !pairs) return IMPDAR_OK;           // (the half layout is read by the pairs only; ps_run repeats the transforms)
    const int lmax_steps = pn_lmax<T>(pairs);
    std::vector<PnPiece> pc;
    int nshort_steps = 0;
    bool need[13] = {};
    for (const PsMfmaRun &r : runs) {
        if (vz && r.len <= PN_SHORT) {
            // consecutive short runs (the single steps of one smeared boundary) share a piece: one pass over the frequencies
            nshort_steps += r.len;
            if (!pc.empty() && pc.back().kind == 1 && pc.back().start + pc.back().len == r.start && pc.back().len + r.len <= PN_SHORT) {
                for (int q = 0; q < r.len; ++q) pc.back().vs[pc.back().len + q] = r.v;
                pc.back().len += r.len;
            } else {
                PnPiece p1{r.v, r.start, r.len, 1, 0, {}};
                for (int q = 0; q < PN_SHORT; ++q) p1.vs[q] = r.v;
                pc.push_back(p1);
            }
            continue;
        }
        const int npiece = (r.len + lmax_steps - 1) / lmax_steps;
        for (int i = 0, at = 0; i < npiece; ++i) {
            const int len = (r.len - at) / (npiece - i);
            int l = 4;
            while ((1 << l) < len) ++l;
            PnPiece p0{r.v, r.start + at, len, 0, l, {}};
            pc.push_back(p0);
            need[l] = true;
            at += len;
        }
    }
    // a direct step costs as much as a tenth of a piece: tables of many layers stay with ps_runs_kernel
    if (nshort_steps > 128 || pc.size() > 256) return IMPDAR_OK;
    std::vector<double> e1;
    if (first_order) {
        // per piece: the reference velocity sqrt(mean v^2) and E(n) / cbar^2 = sum_{t <= n} (v_t^2 / mean - 1)
        e1.assign((size_t)snum, 0.0);
        for (PnPiece &p0 : pc) {
            if (p0.kind != 0) continue;
            long double acc = 0.0L;
            for (int t = 0; t < p0.len; ++t) acc += (long double)vmig[p0.start + t] * vmig[p0.start + t];
            const double vb2 = (double)(acc / p0.len);
            p0.v = std::sqrt(vb2);
            long double run = 0.0L;
            for (int t = 0; t < p0.len; ++t) {
                run += ((long double)vmig[p0.start + t] * vmig[p0.start + t] - (long double)vb2) / (long double)vb2;
                e1[(size_t)p0.start + t] = (double)run;
            }
        }
        if (pl.d_pn_e1.ensure(e1.size() * 8) != hipSuccess) {
            (void)hipGetLastError();
            return IMPDAR_OK;
        }
    }
    PnParams Q;
    bool grew = false;
    std::vector<T> &corr = pn_corr_tables<T>(pl, need, Q.corr_off, &grew);
    std::vector<double> rw((size_t)nf);
    for (int i = 0; i < nf; ++i) rw[i] = 1.0 / w_host[i];
    if (pl.d_pn_pieces.ensure(pc.size() * sizeof(PnPiece)) != hipSuccess || pl.d_pn_corr.ensure(corr.size() * sizeof(T) + 16) != hipSuccess ||
        pl.d_rw.ensure(rw.size() * 8) != hipSuccess || pl.d_edge.ensure((size_t)tnum * (1 + PM_EMAX) * sizeof(int)) != hipSuccess) {
        (void)hipGetLastError();
        return IMPDAR_OK;
    }
    for (int l = 5; l <= 13; ++l)
        if (need[l - 1]) {
            int rc = pl.pn_tw[l].ensure<T>(1 << l, st);
            if (rc) return rc;
        }
    for (int l = 0; l < 14; ++l) Q.tw[l] = pl.pn_tw[l].buf.p;
    IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_pn_pieces.p, pc.data(), pc.size() * sizeof(PnPiece), hipMemcpyHostToDevice, st));
    if (grew || pl.pn_corr_dev != (int)sizeof(T)) {
        IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_pn_corr.p, corr.data(), corr.size() * sizeof(T), hipMemcpyHostToDevice, st));
        pl.pn_corr_dev = (int)sizeof(T);
    }
    IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_rw.p, rw.data(), rw.size() * 8, hipMemcpyHostToDevice, st));
    Q.e1 = nullptr;
    if (first_order) {
        IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_pn_e1.p, e1.data(), e1.size() * 8, hipMemcpyHostToDevice, st));
        Q.e1 = pl.d_pn_e1.as<double>();
    }
    Q.P = P;
    Q.pieces = pl.d_pn_pieces.as<PnPiece>();
    Q.npieces = (int)pc.size();
    Q.rw = pl.d_rw.as<double>();
    Q.corr = pl.d_pn_corr.p;
    Q.edge_cnt = pl.d_edge.as<int>();
    Q.edge_list = Q.edge_cnt + tnum;
    Q.vz = vz ? 1 : 0;
    if (vz) IMPDAR_HIP_CHECK(hipMemsetAsync(Q.edge_cnt, 0, (size_t)tnum * sizeof(int), st));
    {
        int lmax = 4;
        for (int l = 0; l < 13; ++l)
            if (need[l]) lmax = l;
        Q.gmax = 2 << lmax;
    }
    if (pairs) {
        IMPDAR_HIP_CHECK(hipFuncSetAttribute((const void *)ps_nufft_kernel<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)pn_lds_bytes<T>(2 * pn_lmax<T>(true), sizeof(T) == 8, true)));
        hipLaunchKernelGGL((ps_nufft_kernel<T, true>), dim3((unsigned)(tnum / 2 + 1)), dim3(PnCfg<T>::NTH), pn_lds_bytes<T>(Q.gmax, first_order, true), st, Q);
    } else {
        IMPDAR_HIP_CHECK(hipFuncSetAttribute((const void *)ps_nufft_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pn_lds_bytes<T>(2 * PnCfg<T>::LMAX, sizeof(T) == 8)));
        hipLaunchKernelGGL(ps_nufft_kernel<T>, dim3((unsigned)P.nk), dim3(PnCfg<T>::NTH), pn_lds_bytes<T>(Q.gmax, first_order), st, Q);
    }
    if (vz) {
        PsMfmaParams E;                        // (ps_edge_kernel reads P and the lists only)
        E.mfma_count = nullptr;
        E.P = P;
        E.nruns = 0;
        E.pairs = 0;
        E.edge_cnt = Q.edge_cnt;
        E.edge_list = Q.edge_list;
        hipLaunchKernelGGL(ps_edge_kernel_t<T>, dim3(P.nk), dim3(256), 0, st, E);
    }
    IMPDAR_HIP_CHECK(hipGetLastError());
    IMPDAR_HIP_CHECK(hipStreamSynchronize(st));            // (the host tables must outlive their copies)
    if (vz) {
        std::vector<int> cnt((size_t)P.nk);
        IMPDAR_HIP_CHECK(hipMemcpy(cnt.data(), Q.edge_cnt + P.k0, cnt.size() * sizeof(int), hipMemcpyDeviceToHost));
        int worst = 0;
        for (int c : cnt) worst = std::max(worst, c);
        if (worst > PM_EMAX) {
            ps_note_overflow(pl, kx_host, tnum, runs);
            return IMPDAR_OK;                  // contributions missing from TK: discarded, another path produces the result
        }
    }
    *done = true;
    return IMPDAR_OK;
}

// ---- series path (ps_series.h): any v(z) profile -- pieces from the planner (ps_series_plan.h, cached per profile), tables, launch.
// Same contract as ps_mfma_run: *done = false and IMPDAR_OK when the call is not for this path.
template <typename T>
static int ps_series_run(PsPlan &pl, PsParams P, const double *vmig, const double *kx_host, const double *w_host, const double *thr,
                         hipStream_t st, bool *done, double alt_ms_8192 = 0.0, bool allow_pairs = false)
{
    // allow_pairs: as ps_nufft_run's -- the sums go on into the inverse transform: rows k and tnum - k as their Hermitian combination
    // alt_ms_8192 > 0: what the kernel that would take the call otherwise is expected to need (per 8192 wavenumbers); the call is
    // taken only where the estimate from the planner.s model is SR_MARGIN (0.9) of that or less.  0: taken regardless (IMPDAR_PS_MFMA=7)
    *done = false;
    const int snum = P.snum, tnum = P.tnum, nf = P.nf;
    constexpr bool dbl = sizeof(T) == 8;
    if (!P.herm || nf < 64 || nf > SR_NFMAX || snum < 32) return IMPDAR_OK;
    for (int i = 0; i < snum; ++i)
        if (!(thr[i] < 1e-10) || !std::isfinite(vmig[i]) || vmig[i] == 0.0) return IMPDAR_OK;   // the evanescence test must be the sign of coss off the boundary band
    // the gather inverts the dispersion relation on a uniform frequency axis: slot i + 1 at (i + 1) dw, the Nyquist row (slot 0) at nf dw
    // (the kernel's coefficients take w = (i + 1) dw: the axis must be that to rounding -- 2 pi fftfreq is)
    const double dw = w_host[1];
    if (!(dw > 0.0)) return IMPDAR_OK;
    for (int i = 1; i < nf; ++i)
        if (std::fabs(w_host[i] - (double)i * dw) > 2e-15 * (double)i * dw) return IMPDAR_OK;
    if (std::fabs(std::fabs(w_host[0]) - (double)nf * dw) > 2e-15 * (double)nf * dw) return IMPDAR_OK;
    double kxh_max = 0.0;
    for (int k = 0; k < tnum; ++k) kxh_max = std::max(kxh_max, 0.5 * std::fabs(kx_host[k]));
    if (!(kxh_max > 0.0) || !std::isfinite(kxh_max)) return IMPDAR_OK;
    SrHostPlan &hp = pl.sr_plan;
    const bool same = hp.v.size() == (size_t)snum && memcmp(hp.v.data(), vmig, (size_t)snum * 8) == 0 && hp.dt == P.dt && hp.dw == dw &&
                      hp.nf == nf && hp.kxh_max == kxh_max && hp.dbl == dbl && !hp.pieces.empty();
    if (!same) {
        pl.sr_dev = false;
        impdar_trace("ps_series: planning the pieces of a %d-step profile", snum);
        if (!sr_make_plan(hp, vmig, snum, P.dt, dw, nf, kxh_max, dbl)) {
            hp.pieces.clear();
            return IMPDAR_OK;
        }
        impdar_trace("ps_series: %zu pieces", hp.pieces.size());
    }
    if (hp.pieces.empty() || hp.pieces.size() > 4096) return IMPDAR_OK;
    // the whole wavenumber axis with kx[tnum - k] = -kx[k]: a pair of rows per workgroup (ps_series_kernel<T, true>)
    // float32 only: on float64 data the pair's records take the frequencies through LDS in four rounds of two gather passes each and the
    // kernel spills twice as much -- 8192^2: rising gradient 368 -> 365 ms, falling 274 -> 312, firn column 66 -> 72 (float32: 110 -> 77.5,
    // 52 -> 41, 14.7 -> 11.5; profiles/r06_series.txt section 7)
    bool pairs = !dbl && allow_pairs && P.k0 == 0 && P.nk == tnum && tnum >= 2 && tnum % 2 == 0 && !P.fhalf;
    for (int k = 1; 2 * k < tnum && pairs; ++k) pairs = kx_host[k] == -kx_host[tnum - k];
    if (alt_ms_8192 != 0.0) {
        // (alt < 0: per alive pair -- the per-step kernels)
        const double alt = alt_ms_8192 > 0.0 ? alt_ms_8192 : -alt_ms_8192 * hp.alive_pairs;
        const double rest = hp.model_cost - hp.model_gather;
        const double est = dbl ? SR_MS_GATHER_F64 * hp.model_gather + SR_MS_REST_F64 * rest
                               : SR_MS_GATHER_F32 * hp.model_gather + (pairs ? SR_MS_REST_PAIR_F32 : SR_MS_REST_F32) * rest;
        impdar_trace("ps_series: estimate %.1f ms per 8192 wavenumbers against %.1f", est, alt);
        if (est > SR_MARGIN * alt) return IMPDAR_OK;
    }
    bool need[13] = {};
    for (const SrPiece &pc : hp.pieces) need[pc.loglp] = true;
    SrParams Q;
    bool grew = false;
    std::vector<T> &corr = pn_corr_tables<T>(pl, need, Q.corr_off, &grew);
    std::vector<double> rw((size_t)nf);
    for (int i = 0; i < nf; ++i) rw[i] = 1.0 / w_host[i];
    std::vector<T> ev(hp.ev.begin(), hp.ev.end());
    if (pl.d_sr_pieces.ensure(hp.pieces.size() * sizeof(SrPiece)) != hipSuccess || pl.d_sr_ev.ensure(ev.size() * sizeof(T) + 16) != hipSuccess ||
        pl.d_pn_corr.ensure(corr.size() * sizeof(T) + 16) != hipSuccess || pl.d_rw.ensure(rw.size() * 8) != hipSuccess) {
        (void)hipGetLastError();
        return IMPDAR_OK;
    }
    for (int l = 5; l <= 13; ++l)
        if (need[l - 1]) {
            int rc = pl.pn_tw[l].ensure<T>(1 << l, st);
            if (rc) return rc;
        }
    {
        int rc = pl.pn_tw[9].ensure<T>(SR_TWLDS, st);       // (the kernel keeps this one in LDS)
        if (rc) return rc;
    }
    for (int l = 0; l < 14; ++l) Q.tw[l] = pl.pn_tw[l].buf.p;
    if (!pl.sr_dev) {
        IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_sr_pieces.p, hp.pieces.data(), hp.pieces.size() * sizeof(SrPiece), hipMemcpyHostToDevice, st));
        if (!ev.empty()) IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_sr_ev.p, ev.data(), ev.size() * sizeof(T), hipMemcpyHostToDevice, st));
    }
    if (grew || pl.pn_corr_dev != (int)sizeof(T)) {
        IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_pn_corr.p, corr.data(), corr.size() * sizeof(T), hipMemcpyHostToDevice, st));
        pl.pn_corr_dev = (int)sizeof(T);
    }
    IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_rw.p, rw.data(), rw.size() * 8, hipMemcpyHostToDevice, st));
    Q.P = P;
    Q.pieces = pl.d_sr_pieces.as<SrPiece>();
    Q.npieces = (int)hp.pieces.size();
    Q.ev = pl.d_sr_ev.p;
    Q.rw = pl.d_rw.as<double>();
    Q.corr = pl.d_pn_corr.p;
    Q.kxh_max = kxh_max;
    {
        // the grids' LDS doubles as the scratch of the direct sums: the chunk sums of every wave and lane + one group's [L] partial sums
        int lmax = 0;
        for (const SrPiece &pc : hp.pieces) lmax = std::max(lmax, pc.len);
        const int direct_scratch = (SrCfg<T>::NTH / 64) * 64 * 12 + lmax * (int)sizeof(OCp<T>);
        Q.grid_bytes = (std::max(hp.grid_bytes, direct_scratch) + 15) & ~15;
    }
    if constexpr (!dbl) {
        if (pairs) {
            IMPDAR_HIP_CHECK(hipFuncSetAttribute((const void *)ps_series_kernel<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)sr_lds_bytes<float>(SR_GRID_BYTES + 16, true)));
            hipLaunchKernelGGL((ps_series_kernel<float, true>), dim3((unsigned)(tnum / 2 + 1)), dim3(SrCfg<float>::NTH), sr_lds_bytes<float>(Q.grid_bytes, true), st, Q);
        }
    }
    if (!pairs) {
        IMPDAR_HIP_CHECK(hipFuncSetAttribute((const void *)ps_series_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sr_lds_bytes<T>(SR_GRID_BYTES + 16)));
        hipLaunchKernelGGL(ps_series_kernel<T>, dim3((unsigned)P.nk), dim3(SrCfg<T>::NTH), sr_lds_bytes<T>(Q.grid_bytes), st, Q);
    }
    IMPDAR_HIP_CHECK(hipGetLastError());
    IMPDAR_HIP_CHECK(hipStreamSynchronize(st));            // (the host tables must outlive their copies)
    pl.sr_dev = true;
    *done = true;
    return IMPDAR_OK;
}

// ---- matrix-core path (ps_mfma.h): eligibility, row-block table, launch ---------------------------------------
// runs: (velocity, first step, length) of every constant-velocity run (one run for a constant velocity).  Returns
// IMPDAR_OK with *done = true when the frequency sums were produced here; *done = false: not eligible, the vector
// kernels take the call.
static int ps_mfma_run(PsPlan &pl, PsParams P, const std::vector<PsMfmaRun> &runs, bool vz, const double *kx_host, const double *thr,
                       hipStream_t st, bool *done, const char **kernel_name)
{
    *done = false;
    *kernel_name = "ps_mfma_kernel";
    const int snum = P.snum, tnum = P.tnum;
    if (vz && ps_known_overflow(pl, kx_host, tnum, runs)) return IMPDAR_OK;
    if (P.nf % (PM_CH * PM_NQ) != 0 || P.nf < 256 || P.nf > 4096 + 2048 || snum < 256 || runs.empty() || (int)runs.size() > PM_MAX_RUNS) return IMPDAR_OK;
    if (vz)
        for (int i = 0; i < snum; ++i)
            if (!(thr[i] < 1e-10)) return IMPDAR_OK;       // the evanescence test must be the sign of coss off the boundary band
    // row blocks: 32 tiles of 64 steps of one run; runs of a few steps (a layer boundary smeared over 3-4 steps by
    // 2 * gradient(z(t))) get none: ps_trans_kernel sums their steps directly
    std::vector<int2> blocks;
    int nshort_steps = 0;
    for (size_t r = 0; r < runs.size(); ++r) {
        if (vz && runs[r].len <= PM_SHORT) {
            nshort_steps += runs[r].len;
            continue;
        }
        const int ntile = (runs[r].len + PM_TT - 1) / PM_TT;
        for (int a0 = 0; a0 < ntile; a0 += 32) blocks.push_back(make_int2((int)r, a0));
    }
    const int nb = (int)blocks.size();
    // limits of the matrix-core path: long runs, row padding (blocks x 2048 steps against the record), steps in short
    // runs.  A row block costs ~2.4 ms at 8192^2 and the vector runs kernels ~40-50 ms for the whole record: 16 runs /
    // 3 x padding is where the two meet (a 21-row table of equal layers: 44.5 -> 38.1 ms; profiles/r03_ps_layers.txt)
    constexpr int max_long = 16, max_short = 200;
    constexpr double max_pad = 3.0;
    if (nb == 0 || nshort_steps > max_short || (int)runs.size() - (vz ? (int)std::count_if(runs.begin(), runs.end(), [](const PsMfmaRun &r) { return r.len <= PM_SHORT; }) : 0) > max_long) return IMPDAR_OK;
    if ((double)nb * 32 * PM_TT > (double)snum * max_pad + 32 * PM_TT) return IMPDAR_OK;      // many medium runs: rows mostly padding
    // groups of up to PM_NRB row blocks (the state tiles a workgroup keeps in LDS), consecutive blocks together
    int ngroups = (nb + PM_NRB - 1) / PM_NRB;
    const int per_group = (nb + ngroups - 1) / ngroups;
    std::vector<int2> table((size_t)ngroups * PM_NRB, make_int2(-1, 0));
    for (int g = 0, at = 0; g < ngroups; ++g) {
        const int n = std::min(per_group, nb - at);
        for (int i = 0; i < n; ++i, ++at) table[(size_t)g * PM_NRB + i] = blocks[at];
    }
    PsMfmaParams Q;
    Q.P = P;
    Q.nruns = (int)runs.size();
    for (int r = 0; r < Q.nruns; ++r) Q.runs[r] = runs[r];
    for (int r = Q.nruns; r < PM_MAX_RUNS; ++r) Q.runs[r] = PsMfmaRun{0.0, 0, 0};
    Q.ngroups = ngroups;
    // The path's own buffers (runtab: 16 bytes per wavenumber, frequency and long run -- 8.6 GB for 16 runs at
    // 8192 x 4096).  Not getting them is not an error: the vector kernels need none of it and take the call.
    Q.nlong = 0;
    for (int r = 0; r < PM_MAX_RUNS; ++r) Q.long_of[r] = -1;
    for (int r = 0; r < Q.nruns; ++r)
        if (!(vz && runs[r].len <= PM_SHORT)) Q.long_of[r] = Q.nlong++;
    {
        const size_t rt_bytes = (size_t)P.nk * P.nf * Q.nlong * sizeof(double2);
        size_t free_b = 0, total_b = 0;
        if (rt_bytes > pl.d_runtab.bytes && hipMemGetInfo(&free_b, &total_b) == hipSuccess && rt_bytes > free_b / 2) {
            (void)hipGetLastError();
            return IMPDAR_OK;                  // would take more than half of what is free: leave it to the vector kernels
        }
        if (pl.d_blocks.ensure(table.size() * sizeof(int2)) != hipSuccess ||
            pl.d_edge.ensure((size_t)tnum * (1 + PM_EMAX) * sizeof(int)) != hipSuccess || pl.d_runtab.ensure(rt_bytes) != hipSuccess ||
            pl.d_mcount.ensure(8) != hipSuccess) {
            (void)hipGetLastError();
            return IMPDAR_OK;
        }
    }
    IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_blocks.p, table.data(), table.size() * sizeof(int2), hipMemcpyHostToDevice, st));
    Q.blocks = pl.d_blocks.as<int2>();
    Q.edge_cnt = pl.d_edge.as<int>();
    Q.edge_list = Q.edge_cnt + tnum;
    if (vz) IMPDAR_HIP_CHECK(hipMemsetAsync(Q.edge_cnt, 0, (size_t)tnum * sizeof(int), st));
    Q.mfma_count = pl.d_mcount.as<unsigned long long>();
    IMPDAR_HIP_CHECK(hipMemsetAsync(Q.mfma_count, 0, 8, st));
    Q.vz = vz ? 1 : 0;
    Q.runtab = pl.d_runtab.as<double2>();
    // the whole wavenumber axis with kx[tnum - k] = -kx[k]: rows k and tnum - k turn by the same angles -- one set-up walk and
    // one table row for both; a slab of a kx-sharded run has no mirror rows
    bool sym = P.k0 == 0 && P.nk == tnum && tnum >= 2;
    for (int k = 1; 2 * k < tnum && sym; ++k) sym = kx_host[k] == -kx_host[tnum - k];      // (the Nyquist row of an even axis is its own partner)
    Q.pairs = sym ? 1 : 0;
    {
        // set-up pass: per-run phases of every (wavenumber, frequency), boundary frequencies, the steps of the short runs
        const size_t lds = 2 * 8 * 2 * PM_SHORT * sizeof(float);
        const dim3 grid(sym ? tnum / 2 + 1 : P.nk);
        if (P.nf <= 2048) hipLaunchKernelGGL(ps_setup_kernel<4>, grid, dim3(512), lds, st, Q);
        else if (P.nf <= 4096) hipLaunchKernelGGL(ps_setup_kernel<8>, grid, dim3(512), lds, st, Q);
        else hipLaunchKernelGGL(ps_setup_kernel<12>, grid, dim3(512), lds, st, Q);
    }
    IMPDAR_HIP_CHECK(hipFuncSetAttribute((const void *)ps_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PM_LDS_BYTES));
    hipLaunchKernelGGL(ps_mfma_kernel, dim3((unsigned)P.nk * Q.ngroups), dim3(PM_WAVES * 64), PM_LDS_BYTES, st, Q);
    if (vz) hipLaunchKernelGGL(ps_edge_kernel_t<float>, dim3(P.nk), dim3(256), 0, st, Q);
    IMPDAR_HIP_CHECK(hipGetLastError());
    // the host table must outlive its async copy
    IMPDAR_HIP_CHECK(hipStreamSynchronize(st));
    if (vz) {
        // ps_setup_kernel takes every boundary frequency out of the matrix-core sums but lists only the first PM_EMAX per
        // wavenumber for ps_edge_kernel (more needs a table of many runs on a rational grid).  The counter keeps
        // counting: an overflow means contributions are missing from TK -- the result is discarded and the vector
        // kernels, which walk every boundary frequency themselves, produce it.
        std::vector<int> cnt((size_t)P.nk);
        IMPDAR_HIP_CHECK(hipMemcpy(cnt.data(), Q.edge_cnt + P.k0, cnt.size() * sizeof(int), hipMemcpyDeviceToHost));
        int worst = 0;
        for (int c : cnt) worst = std::max(worst, c);
        const bool force_overflow = getenv("IMPDAR_PS_TEST_EDGE_OVERFLOW") != nullptr;      // test hook (read per call)
        if (worst > PM_EMAX) ps_note_overflow(pl, kx_host, tnum, runs);    // (remembered: not tried again on this geometry)
        if (worst > PM_EMAX || force_overflow) return IMPDAR_OK;           // *done stays false
    }
    {
        unsigned long long n = 0;
        IMPDAR_HIP_CHECK(hipMemcpy(&n, Q.mfma_count, 8, hipMemcpyDeviceToHost));
        pl.mfma_instructions = (double)n;
    }
    *done = true;
    return IMPDAR_OK;
}

template <typename T>
static int ps_run(impdar_ctx *ctx, PsPlan &pl, const void *d_data, int snum, int tnum, int nt, const double *kx,
                  const double *ws, double dt, const double *tt_us, double vconst, const double *vmig, int vlen,
                  double htaper, double vtaper, void *d_out, int k0 = 0, int nk = -1, void *tk_out = nullptr)
{
    // tk_out != null: a rank of a kx-sharded run -- the transforms (replicated on every rank) and the frequency sums of
    // wavenumbers [k0, k0 + nk) only, written to tk_out [nk][snum]; no inverse transform (impdar_phaseshift_finish_dev)
    if (nk < 0) nk = tnum;
    hipStream_t st = ctx->stream;
    const bool dbl = sizeof(T) == 8;
    // IMPDAR_PS_FFT=strided: the transforms over the traces / wavenumbers as rocFFT's strided plans on the arrays as
    // they lie (rounds 1-3a); default: transpose, contiguous plan, transpose (see ps_transpose_c)
    // rocfft: rocFFT's plans also where the library's own row transforms apply; own: that default spelled out (tests that pin one)
    const char *fe = getenv("IMPDAR_PS_FFT");
    const bool rows_form = !(fe && strcmp(fe, "strided") == 0);
    const bool no_own = fe && strcmp(fe, "rocfft") == 0;
    if (pl.owner != ctx || pl.dtype != (dbl ? IMPDAR_F64 : IMPDAR_F32) || pl.snum != snum || pl.tnum != tnum || pl.nt != nt ||
        pl.rows_form != rows_form) {
        pl.own_calls = 0;
        pl.dtype = -1;
        pl.rm_state = -1;
        pl.r_ready = pl.c_ready = false;
        if (pl.owner != ctx) {               // another device / stream: drop everything bound to the old one
            pl.Xr.release();
            pl.d_taper.release();
            pl.d_blocks.release();
            pl.d_edge.release();
            pl.d_runtab.release();
            pl.d_pr_runs.release();
            pl.d_pn_pieces.release();
            pl.d_pn_corr.release();
            pl.d_pn_e1.release();
            pl.pn_corr_dev = -1;
            pl.d_sr_pieces.release();
            pl.d_sr_ev.release();
            pl.sr_dev = false;
            for (OwnTwiddles &t : pl.pn_tw) {
                t.buf.release();
                t.nt = 0;
            }
            pl.d_mcount.release();
            pl.d_pr_stages.release();
            pl.d_rw.release();
            pl.d_sendbuf.release();
            pl.d_slab.release();
            pl.X.release(); pl.TK.release(); pl.d_kx.release(); pl.d_w.release(); pl.d_vz.release(); pl.d_thr.release(); pl.d_sm.release(); pl.d_part.release();
            pl.d_sched.release();
            pl.d_rowmap.release();
            pl.d_eps.release();
            pl.owner = ctx;
        }
        pl.c_ready = false;              // the forward transforms are made on first use (below): which pair depends on the walk
        pl.rows_form = rows_form;
        pl.b_ready = false;              // the inverse transform over the traces: made below, beside the forward pair
        IMPDAR_HIP_CHECK(pl.X.ensure((size_t)tnum * nt * 2 * sizeof(T)));
        IMPDAR_HIP_CHECK(pl.d_kx.ensure((size_t)tnum * 8));
        IMPDAR_HIP_CHECK(pl.d_w.ensure((size_t)nt * 8));
        IMPDAR_HIP_CHECK(pl.d_vz.ensure((size_t)snum * 8));
        IMPDAR_HIP_CHECK(pl.d_thr.ensure((size_t)snum * 8));
        pl.dtype = dbl ? IMPDAR_F64 : IMPDAR_F32;
        pl.snum = snum;
        pl.tnum = tnum;
        pl.nt = nt;
    }
    std::vector<double> w(ws, ws + nt), thr(snum, 0.0);
    for (int i = 0; i < nt; ++i)
        if (w[i] == 0.0) w[i] = 1e-10 / dt;                            // :400-402
    // Hermitian walk (see ps_load_slot): the radargram is real, so frequencies 1..nt/2-1 also stand for their mirror
    // images.  Taken only when the axes are exactly antisymmetric and the replaced zero frequency is evanescent
    // wherever kx != 0 (it always is for a physical geometry: |v kx / 2| >= v pi / (tnum dx) against 1e-10/dt);
    // anything else -- and IMPDAR_PS_HERMITIAN=0 -- keeps the reference's walk over all nt frequencies.
    const double w0 = 1e-10 / dt;
    std::vector<int> k_zero;                // wavenumbers with kx = 0 (the zero-frequency row propagates there)
    bool herm = nt >= 4 && (nt & (nt - 1)) == 0 && ws[0] == 0.0;
    {
        const char *he = getenv("IMPDAR_PS_HERMITIAN");      // read per call: the tests compare the two walks in one process
        if (he && atoi(he) == 0) herm = false;
    }
    if (herm) {
        for (int i = 1; i < nt / 2 && herm; ++i) herm = std::isfinite(ws[i]) && ws[i] != 0.0 && ws[nt - i] == -ws[i];
        herm = herm && std::isfinite(ws[nt / 2]) && ws[nt / 2] != 0.0;
        double vmin = std::fabs(vconst);
        if (vlen) {
            vmin = std::fabs(vmig[0]);
            for (int i = 0; i < snum; ++i) {
                herm = herm && std::isfinite(vmig[i]);
                vmin = std::min(vmin, std::fabs(vmig[i]));
            }
        }
        herm = herm && std::isfinite(vmin) && std::isfinite(w0);
        for (int k = 0; k < tnum && herm; ++k) {
            // (the wavenumber Nyquist row of an even trace count is its own mirror image: only kx^2 enters)
            const int km = (tnum - k) % tnum;
            herm = std::isfinite(kx[k]) && (km == k || kx[km] == -kx[k]);
            if (kx[k] == 0.0) k_zero.push_back(k);
            else herm = herm && std::fabs(0.5 * vmin * kx[k]) > 2.0 * w0;     // w0 evanescent with a wide margin
        }
        herm = herm && k_zero.size() <= 4;
    }
    const int nf = herm ? nt / 2 : nt;
    const int fstride = herm ? nt / 2 + 1 : nt;
    if (herm) {
        w[0] = ws[nt / 2];                  // slot order: Nyquist first, then rows 1..nt/2-1 (already in place)
    }
    {
        // TK: the frequency sums [tnum][snum] of an unsharded call, and the scratch of the transposed forward transform
        // (nt / 2 + 1 rows of tnum -- more than snum rows when the caller pads the time axis beyond the next power of
        // two).  A rank of a kx-sharded run writes its sums to the caller's slab: it only needs the scratch.
        const size_t scratch_rows = (herm && pl.rows_form) ? (size_t)(nt / 2 + 1) : 0;
        const size_t rows = tk_out ? scratch_rows : std::max<size_t>((size_t)snum, scratch_rows);
        if (rows) IMPDAR_HIP_CHECK(pl.TK.ensure((size_t)tnum * rows * 2 * sizeof(T)));
    }
    // Power-of-two sizes run their transforms on the library's own row kernels (own_fft.h), every call: nothing to compile,
    // no plan to make (rocFFT: 0.25-3 s per process, its kernels for lengths above 1024 are compiled at run time), and within
    // 0.1 ms of rocFFT's plans at 8192^2 since the long rows run 1024 threads (round 5; until then the first call of a size
    // only).  (Plans made on a thread during a first call: a process that exits while rocFFT is still compiling on
    // another thread crashes in its teardown -- rc -11 / -6 in 2 of 2 such exits, profiles/r05_first_call.txt.)
    bool use_own = false;
    if (herm && pl.rows_form && !no_own && own_fft_len_ok(nt / 2) && own_fft_len_ok(tnum)) {
        IMPDAR_HIP_CHECK(pl.Xr.ensure((size_t)tnum * nt * sizeof(T)));
        use_own = true;
        pl.own_calls += 1;
        impdar_trace("phaseshift: transforms on the library's own row kernels");
        int rc;
        if ((rc = pl.tw_time.ensure<T>(nt, st)) || (rc = pl.tw_trace.ensure<T>(tnum, st))) return rc;
    }
    if (!use_own) {
        int rc;
        const rocfft_array_type ci = rocfft_array_type_complex_interleaved;
        // the plans this call still lacks, created side by side (run-time compilation: impdar_parallel_plans)
        std::vector<std::function<int()>> makers;
        const bool rows_form_b = pl.rows_form;
        if (!pl.b_ready)
            makers.push_back([&] {
                return rows_form_b ? pl.b_trace.create(rocfft_transform_type_complex_inverse, dbl, true, tnum, snum, ci, ci, 1, tnum, 1, tnum,
                                                       1.0 / tnum, st)
                                   : pl.b_trace.create(rocfft_transform_type_complex_inverse, dbl, true, tnum, snum, ci, ci, snum, 1, snum, 1,
                                                       1.0 / tnum, st);
            });
        if (herm && !pl.r_ready) {
            IMPDAR_HIP_CHECK(pl.Xr.ensure((size_t)tnum * nt * sizeof(T)));
            makers.push_back([&] {
                return pl.r_time.create(rocfft_transform_type_real_forward, dbl, false, nt, tnum, rocfft_array_type_real,
                                        rocfft_array_type_hermitian_interleaved, 1, nt, 1, fstride, 1.0, st);
            });
            makers.push_back([&] {
                return pl.rows_form ? pl.r_trace.create(rocfft_transform_type_complex_forward, dbl, true, tnum, fstride, ci, ci, 1, tnum,
                                                        1, tnum, 1.0, st)
                                    : pl.r_trace.create(rocfft_transform_type_complex_forward, dbl, true, tnum, fstride, ci, ci, fstride, 1,
                                                        fstride, 1, 1.0, st);
            });
        }
        if (!herm && !pl.c_ready) {
            makers.push_back([&] { return pl.f_time.create(rocfft_transform_type_complex_forward, dbl, true, nt, tnum, ci, ci, 1, nt, 1, nt, 1.0, st); });
            makers.push_back([&] { return pl.f_trace.create(rocfft_transform_type_complex_forward, dbl, true, tnum, nt, ci, ci, nt, 1, nt, 1, 1.0, st); });
        }
        if (!makers.empty()) {
            impdar_trace("phaseshift: %zu rocFFT plans to make", makers.size());
            if ((rc = impdar_parallel_plans(ctx->device, makers))) return rc;
            impdar_trace("phaseshift: plans ready");
            pl.b_ready = true;
            if (herm) pl.r_ready = true;
            else pl.c_ready = true;
        }
    }
    if (vlen)
        for (int i = 0; i < snum; ++i) {
            const double tau = tt_us[i] / 1.0e6;                         // :441
            const double r = tau / tt_us[snum - 1] / 1e6;                // :484
            thr[i] = r * r;
        }
    IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_kx.p, kx, (size_t)tnum * 8, hipMemcpyHostToDevice, st));
    IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_w.p, w.data(), (size_t)nf * 8, hipMemcpyHostToDevice, st));
    if (vlen) {
        IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_vz.p, vmig, (size_t)snum * 8, hipMemcpyHostToDevice, st));
        IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_thr.p, thr.data(), (size_t)snum * 8, hipMemcpyHostToDevice, st));
    }
    PsParams P;
    P.F = pl.X.p;
    P.fstride = fstride;
    P.k0 = k0;
    P.nk = nk;
    P.TK = tk_out ? tk_out : pl.TK.p;
    P.kx = pl.d_kx.as<double>();
    P.w = pl.d_w.as<double>();
    P.vz = pl.d_vz.as<double>();
    P.thr = pl.d_thr.as<double>();
    P.vconst = vconst;
    P.dt = dt;
    P.snum = snum;
    P.tnum = tnum;
    P.nt = nt;
    P.nf = nf;
    P.herm = herm ? 1 : 0;
    P.vz_mode = vlen ? 1 : 0;
    P.fhalf = 0;
    P.vtol = dbl ? 1e-11 : 1e-10;
    P.sched = P.tsched = P.rowmap = nullptr;
    P.eps = nullptr;
    P.sm_step = P.sm_tile = nullptr;
    P.sm_part = nullptr;
    P.sm_nchunks = 1;
    P.sm_m = 8;
    std::vector<int> sched, rowmap;
    std::vector<double> epsum;
    if (vlen) {
        // runs of constant velocity (ps_vz32_kernel): a step starts a new run when its velocity differs from the
        // run's first by more than vtol (relative) -- 2*gradient(z(t)) of a layered table is constant inside a
        // layer up to ~4e-13 of rounding noise, and a 1e-10 velocity error moves the phase by < 3e-6 rad over
        // 8192 steps (float32: ignored; float64: vtol 1e-11 and the deviation is carried along as a phase,
        // ps_vz64_kernel).  Profiles that change at (nearly) every step keep the per-step kernels.
        const int ntile = (snum + 15) / 16;
        sched.assign((size_t)snum + (ntile + 31) / 32, 0);
        double vrun = -1.0;
        int ndirty = 0;
        epsum.assign(ntile, 0.0);
        for (int i = 0; i < snum; ++i) {
            if (i == 0 || std::fabs(vmig[i] - vrun) > P.vtol * std::fabs(vmig[i])) {     // a run always starts at step 0
                const int tile = i / 16;
                const unsigned bit = 1u << (tile & 31);
                unsigned &word = reinterpret_cast<unsigned &>(sched[snum + tile / 32]);
                sched[i] = 1;
                ndirty += (word & bit) ? 0 : 1;
                word |= bit;
                vrun = vmig[i];
            }
            epsum[i / 16] += vmig[i] / vrun - 1.0;      // float64 kernel: the run's velocity noise, tile by tile
        }
        // the per-step tiles of the runs kernels cost several quiet tiles each: beyond a share of such tiles the
        // per-step kernel is faster for float32 (2048^2, 40 / 80 / 160 layers: 4.6 / 7.5 / 12.2 ms against
        // 5.5 / 6.8 / 9.5 ms); the float64 runs kernel stays ahead until every tile holds a change (10.1 / 16.6 /
        // 27.5 ms against 26.4 ms throughout: its per-step tiles pay the square root and sincos at the changes
        // only).  profiles/tools/ps_dirty.py.
        const double dirty_max = dbl ? 0.9 : 0.5;
        bool vfinite = true;                // a velocity profile with NaN / inf entries takes the per-step kernel
        for (int i = 0; i < snum; ++i) vfinite = vfinite && std::isfinite(vmig[i]) && vmig[i] != 0.0;
        if (vfinite && ((double)ndirty <= dirty_max * ntile || snum <= 64)) {
            IMPDAR_HIP_CHECK(pl.d_sched.ensure(sched.size() * sizeof(int)));
            IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_sched.p, sched.data(), sched.size() * sizeof(int), hipMemcpyHostToDevice, st));
            P.sched = pl.d_sched.as<int>();
            P.tsched = P.sched + snum;
            if (dbl) {
                IMPDAR_HIP_CHECK(pl.d_eps.ensure(epsum.size() * sizeof(double)));
                IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_eps.p, epsum.data(), epsum.size() * sizeof(double), hipMemcpyHostToDevice, st));
                P.eps = pl.d_eps.as<double>();
            }
        }
    }
    // ---- which layout of the spectrum (P.fhalf).  When ps_nufft_kernel will take the call with a pair of wavenumbers per workgroup
    // (the whole axis, antisymmetric kx; a constant velocity or a table of up to 16 thick layers and 64 runs -- the conditions of the
    // dispatch below that can be told before the transforms), the transform over the TRACES goes first, on the radargram's own rows
    // with the taper applied on their way in, and only the wavenumbers k = 0 .. tnum/2 are kept, with all frequencies:
    //   R2C over x [snum][tnum] -> [snum][tnum/2 + 1];  transpose (zero rows up to nt) -> [tnum/2 + 1][nt];  C2C over t in place
    // instead of taper + transpose, R2C over t, transpose, C2C over x, transpose: two passes over the array less.  The pair (k, tnum - k)
    // reads both of its rows out of row k (FK[tnum - k][w] = conj FK[k][-w]).  Should the kernel hand the call on after all (boundary
    // frequencies beyond its lists), the transforms are repeated in the other layout (front_full below).
    // ---- will the transform path (ps_nufft_kernel) take a v(z) table?  Its cost is per PIECE (a run of constant velocity, cut at
    // 4096 / 2048 steps) and per directly summed step of the short runs between them, and hardly depends on the record's length; the
    // runs kernels' is per (frequency, step) plus a term per run.  Device ms per 8192 wavenumbers, pairs of wavenumbers per transform
    // (profiles/r06_transforms.txt section 4: tables of 4 ... 41 rows at 8192^2, 4096^2, 2048^2):
    //   float32  ps_nufft_kernel (0.135 + 0.06 nf/4096) (pieces + short steps / 3)     ps_runs_kernel 1.2 + 8.5 (nf/4096)(snum/8192) + 0.05 runs nf/4096
    //   float64  up to 24 thick layers from 2048 frequencies on (33.6 against ps_vz64_kernel's 57.2 ms at 41 rows / 21 layers, 8192^2; level at
    //            4096^2), 16 below
    bool nufft_first = false;
    int vz_runs = 0, vz_long = 0;
    if (vlen) {
        int len = 0, pieces = 0, nshort = 0;
        const int lmax = dbl ? 2048 : 4096;
        auto close_run = [&]() {
            if (len > PN_SHORT) pieces += (len + lmax - 1) / lmax;
            else nshort += len;
            vz_long += len > PM_SHORT;
        };
        for (int i = 0; i < snum; ++i) {
            if (sched[i]) {
                if (vz_runs) close_run();
                vz_runs += 1;
                len = 0;
            }
            len += 1;
        }
        close_run();
        const double fq = (double)nf / 4096.0;
        if (dbl)
            nufft_first = vz_long <= (nf >= 2048 ? 24 : 16);
        else
            nufft_first = nshort <= 128 && (0.135 + 0.06 * fq) * ((double)pieces + (double)nshort / 3.0) <=
                                               1.2 + 8.5 * fq * ((double)snum / 8192.0) + 0.05 * (double)vz_runs * fq;
    }
    // ---- which layout of the spectrum (P.fhalf): see above
    bool half_front = false;
    {
        const char *me = getenv("IMPDAR_PS_MFMA");
        const int pref = me ? atoi(me) : 1;
        bool sym = herm && use_own && pl.rows_form && !tk_out && k0 == 0 && nk == tnum && tnum % 2 == 0 && tnum >= 64 && own_fft_len_ok(nt) &&
                   nf >= 64 && nf <= PN_NFMAX && snum >= 64 && (pref == 1 || pref == 6) &&
                   !(k_zero.size() > 1 || (k_zero.size() == 1 && k_zero[0] != 0));
        for (int k = 1; 2 * k < tnum && sym; ++k) sym = kx[k] == -kx[tnum - k];
        if (sym && !vlen) sym = std::isfinite(vconst) && vconst != 0.0;
        if (sym && vlen) sym = P.sched != nullptr && (pref == 6 || nufft_first);
        half_front = sym;
    }
    std::vector<double> tap_h, tap_v;
    if (half_front) {
        tap_h.resize((size_t)tnum);
        tap_v.resize((size_t)snum);
        for (int j = 0; j < tnum; ++j) tap_h[j] = impdar_taper_w(j, tnum, htaper);
        for (int i = 0; i < snum; ++i) tap_v[i] = impdar_taper_w(i, snum, vtaper);
        if (pl.d_taper.ensure(((size_t)tnum + snum) * 8) != hipSuccess) {
            (void)hipGetLastError();
            half_front = false;
        } else {
            IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_taper.p, tap_h.data(), (size_t)tnum * 8, hipMemcpyHostToDevice, st));
            IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_taper.as<double>() + tnum, tap_v.data(), (size_t)snum * 8, hipMemcpyHostToDevice, st));
        }
    }
    dim3 tgrid((tnum + 63) / 64, (nt + 63) / 64);
    int rc;
    // the transforms of the [k][w > 0] layout every kernel reads (Hermitian walk on the library's own or rocFFT's plans)
    auto front_full = [&]() -> int {
        int rc;
        hipLaunchKernelGGL((ps_taper_pad_transpose_real<T>), tgrid, dim3(256), 0, st, (const T *)d_data, pl.Xr.as<T>(), snum,
                           tnum, nt, htaper, vtaper);
        if (use_own) {
            if ((rc = own_fft_launch<T>(OWN_R2C, nt, (size_t)tnum, pl.Xr.p, pl.X.p, (size_t)nt, (size_t)fstride, 1.0, pl.tw_time, st))) return rc;
        } else if ((rc = pl.r_time.exec(pl.Xr.p, pl.X.p))) {
            return rc;
        }
        if (pl.rows_form) {
            // over the traces on contiguous rows: X [x][fstride] -> [fstride][x] (pl.TK is free until the frequency sums
            // write it, and large enough: snum > nt / 2), transform, and back -> X [k][fstride]
            ps_launch_transpose<T>(pl.X.p, pl.TK.p, tnum, fstride, st);
            if (use_own) {
                if ((rc = own_fft_launch<T>(OWN_C2C_FWD, tnum, (size_t)fstride, pl.TK.p, pl.TK.p, (size_t)tnum, (size_t)tnum, 1.0, pl.tw_trace, st))) return rc;
            } else if ((rc = pl.r_trace.exec(pl.TK.p, nullptr))) {
                return rc;
            }
            ps_launch_transpose<T>(pl.TK.p, pl.X.p, fstride, tnum, st);
            IMPDAR_HIP_CHECK(hipGetLastError());
        } else if ((rc = pl.r_trace.exec(pl.X.p, nullptr))) {
            return rc;
        }
        return IMPDAR_OK;
    };
    // the call handed on by ps_nufft_kernel after the transforms were made for its pairs: the other layout after all
    auto leave_half_front = [&]() -> int {
        if (!P.fhalf) return IMPDAR_OK;
        impdar_trace("phaseshift: the transform path handed the call on: transforms repeated in the [k][w > 0] layout");
        P.fhalf = 0;
        P.fstride = fstride;
        return front_full();
    };
    if ((rc = impdar_ctx_tic(ctx))) return rc;
    if (herm && half_front) {
        const int hs = tnum / 2 + 1;
        const double *th = pl.d_taper.as<double>(), *tv = th + tnum;
        // (pl.TK: free until the frequency sums write it, and at least tnum x snum complex)
        if ((rc = own_fft_launch<T>(OWN_R2C, tnum, (size_t)snum, d_data, pl.TK.p, (size_t)tnum, (size_t)hs, 1.0, pl.tw_trace, st, th, tv))) return rc;
        {
            constexpr int TS = sizeof(T) == 4 ? 64 : 32;
            hipLaunchKernelGGL((ps_transpose_pad<T, TS>), dim3((hs + TS - 1) / TS, (nt + TS - 1) / TS), dim3(256), 0, st, pl.TK.as<Cp<T>>(),
                               pl.X.as<Cp<T>>(), snum, hs, nt);
        }
        if ((rc = own_fft_launch<T>(OWN_C2C_FWD, nt, (size_t)hs, pl.X.p, pl.X.p, (size_t)nt, (size_t)nt, 1.0, pl.tw_time, st))) return rc;
        IMPDAR_HIP_CHECK(hipGetLastError());
        P.fhalf = 1;
        P.fstride = nt;
        impdar_trace("phaseshift: the transform over the traces first: wavenumbers k >= 0 with all frequencies, for pairs in ps_nufft_kernel");
    } else if (herm) {
        if ((rc = front_full())) return rc;
    } else {
        hipLaunchKernelGGL((ps_taper_pad_transpose<T>), tgrid, dim3(256), 0, st, (const T *)d_data, pl.X.as<Cp<T>>(), snum,
                           tnum, nt, htaper, vtaper);
        if ((rc = pl.f_time.exec(pl.X.p, nullptr))) return rc;
        if ((rc = pl.f_trace.exec(pl.X.p, nullptr))) return rc;
    }
    // (only the runs kernels of the vector path use it, and it costs ~2 ms of host time at config 5 with the GPU idle:
    // made when they are about to be launched, not when the matrix-core path takes the call)
    auto order_rows = [&]() -> int {
        // Launch order.  A wavenumber that holds a frequency on the evanescent boundary of a run walks it in
        // fp64 in every tile of that run (~2x the tile time); spread over the launch, the last such rows finish
        // alone after everything else (2 ms at config 5).  They go first, longest first.  The test here only
        // orders the launch -- generous tolerance, the kernel decides for itself: |0.5 v kx| within 1e-7 of
        // some |w|.
        if (nk != tnum) return IMPDAR_OK;                          // (a slab of a sharded run keeps the natural order)
        // the order depends on the axes and the profile only: a repeated geometry re-uses the last one
        if (pl.rm_state >= 0 && pl.rm_kx.size() == (size_t)tnum && pl.rm_ws.size() == (size_t)nt && pl.rm_v.size() == (size_t)snum &&
            memcmp(pl.rm_kx.data(), kx, (size_t)tnum * 8) == 0 && memcmp(pl.rm_ws.data(), ws, (size_t)nt * 8) == 0 &&
            memcmp(pl.rm_v.data(), vmig, (size_t)snum * 8) == 0) {
            if (pl.rm_state == 1) P.rowmap = pl.d_rowmap.as<int>();
            return IMPDAR_OK;
        }
        pl.rm_state = -1;
        pl.rm_kx.assign(kx, kx + tnum);
        pl.rm_ws.assign(ws, ws + nt);
        pl.rm_v.assign(vmig, vmig + snum);
        std::vector<std::pair<double, int>> runs;      // (velocity, steps) of every run
        for (int i = 0; i < snum; ++i) {
            if (sched[i]) runs.emplace_back(vmig[i], 0);
            runs.back().second += 1;
        }
        pl.rm_state = 0;
        if (runs.size() <= 64 && tnum >= 512) {
            std::vector<double> aw(nt);
            for (int j = 0; j < nt; ++j) aw[j] = std::fabs(ws[j] == 0.0 ? 1e-10 / dt : ws[j]);
            std::sort(aw.begin(), aw.end());
            std::vector<std::pair<int, int>> score(tnum);      // (-steps spent walking, row)
            int flagged = 0;
            for (int k = 0; k < tnum; ++k) {
                int steps = 0;
                for (const auto &r : runs) {
                    const double target = 0.5 * r.first * std::fabs(kx[k]);
                    const auto it = std::lower_bound(aw.begin(), aw.end(), target);
                    const double hi = it != aw.end() ? *it : aw.back(), lo = it != aw.begin() ? *(it - 1) : aw.front();
                    const double band = dbl ? 2e-6 : 1e-7;
                    if (std::fabs(hi - target) <= band * target || std::fabs(lo - target) <= band * target) steps += r.second;
                }
                score[k] = std::make_pair(-steps, k);
                flagged += steps > 0;
            }
            if (flagged > 0 && flagged < tnum) {
                std::stable_sort(score.begin(), score.end());
                rowmap.resize(tnum);
                for (int b = 0; b < tnum; ++b) rowmap[b] = score[b].second;
                IMPDAR_HIP_CHECK(pl.d_rowmap.ensure((size_t)tnum * sizeof(int)));
                IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_rowmap.p, rowmap.data(), (size_t)tnum * sizeof(int), hipMemcpyHostToDevice, st));
                P.rowmap = pl.d_rowmap.as<int>();
                pl.rm_state = 1;
            }
        }
        return IMPDAR_OK;
    };
    if ((rc = impdar_ctx_ktic(ctx))) return rc;
    impdar_trace("phaseshift: forward transforms enqueued");
    bool mfma_done = false;
    const char *mfma_kernel_name = "";
    pl.mfma_instructions = -1.0;
    int long_runs = 0;                       // runs of constant velocity longer than a smeared layer boundary (metrics)
    if constexpr (sizeof(T) == 8) {
        // float64 data at a constant velocity: the transform path (ps_nufft.h: a 14-point window in float64 arithmetic); a v(z)
        // table keeps the vector kernels (the velocity noise inside its runs: ps_nufft.h)
        const char *me = getenv("IMPDAR_PS_MFMA");
        const int pref = me ? atoi(me) : 1;
        if (!vlen && pref != 0 && std::isfinite(vconst) && vconst != 0.0) {
            std::vector<PsMfmaRun> one{PsMfmaRun{vconst, 0, snum}};
            if ((rc = ps_nufft_run<double>(pl, P, one, false, kx, w.data(), thr.data(), st, &mfma_done, nullptr, tk_out == nullptr))) return rc;
            if (!mfma_done && (rc = leave_half_front())) return rc;
            if (mfma_done) mfma_kernel_name = "ps_nufft_kernel";
        }
        // a v(z) table of up to 16 thick layers: the transform path with the runs' velocity noise (~4e-13, cut at 1e-11) as its
        // first-order term (ps_nufft.h); 6 asks for it at any number of layers
        if (vlen && P.sched && (pref == 1 || pref == 6)) {
            std::vector<PsMfmaRun> mruns;
            bool ok = true;
            for (int i = 0; i < snum && ok; ++i) {
                ok = std::isfinite(vmig[i]) && vmig[i] != 0.0;
                if (sched[i]) mruns.push_back(PsMfmaRun{vmig[i], i, 0});
                if (!mruns.empty()) mruns.back().len += 1;
            }
            int nlong = 0;
            for (const PsMfmaRun &r : mruns) nlong += r.len > PM_SHORT;
            long_runs = nlong;
            if (ok && (pref == 6 || nufft_first)) {
                if ((rc = ps_nufft_run<double>(pl, P, mruns, true, kx, w.data(), thr.data(), st, &mfma_done, vmig, tk_out == nullptr))) return rc;
                if (!mfma_done && (rc = leave_half_front())) return rc;
                if (mfma_done) mfma_kernel_name = "ps_nufft_kernel";
            }
        }
        // any other v(z) profile -- a velocity that changes at every step, many layers: the series path (ps_series.h)
        if (vlen && !mfma_done && (pref == 1 || pref == 7)) {
            // (what would run otherwise: ps_smooth_kernel, 10.8e-6 ms per alive pair; the runs kernel ps_vz64_kernel
            // -- 43 ms for the record + 0.09 per step that starts a run, at 8192^2 (4-row table 43, 41 rows 56, a firn column's
            // 1960 changing steps 226; 4096^2: 14 / 121 per 8192 wavenumbers): profiles/r06_series.txt)
            int nstarts = 0;
            if (P.sched)
                for (int i = 0; i < snum; ++i) nstarts += sched[i] != 0;
            const double alt = pref == 7 ? 0.0 : (P.sched ? 43.0 * ((double)nf / 4096.0) * ((double)snum / 8192.0) + 0.09 * (double)nstarts * ((double)nf / 4096.0)
                                                          : -SR_MS_PER_PAIR_F64);
            if ((rc = ps_series_run<double>(pl, P, vmig, kx, w.data(), thr.data(), st, &mfma_done, alt, tk_out == nullptr))) return rc;
            if (mfma_done) mfma_kernel_name = "ps_series_kernel";
        }
    }
    if constexpr (sizeof(T) == 4) {
        // float32: the frequency sums on the matrix cores when the depth axis is a few long runs of constant velocity
        std::vector<PsMfmaRun> mruns;
        bool ok = true;
        if (vlen) {
            for (int i = 0; i < snum && ok; ++i) {
                ok = std::isfinite(vmig[i]) && vmig[i] != 0.0;
                if (sched[i]) mruns.push_back(PsMfmaRun{vmig[i], i, 0});
                if (!mruns.empty()) mruns.back().len += 1;
            }
        } else {
            mruns.push_back(PsMfmaRun{vconst, 0, snum});
        }
        // IMPDAR_PS_MFMA: 0 the vector kernels only; 2 / 3 only ps_mfma_kernel / only ps_runs_kernel of the matrix-core
        // paths (A/B runs, tests).  By themselves: up to 3 thick layers -> ps_mfma_kernel (64-step tiles, phases from a table);
        // more long runs -> ps_runs_kernel (8-step tiles, phases generated in the kernel); whichever declines
        // (ps_mfma_kernel: rows mostly padding on short records) hands over to the other, then to the vector kernels.
        const char *me = getenv("IMPDAR_PS_MFMA");
        const int pref = me ? atoi(me) : 1;
        int nlong = 0;
        for (const PsMfmaRun &r : mruns) nlong += r.len > PM_SHORT;
        long_runs = nlong;
        const bool force_overflow = getenv("IMPDAR_PS_TEST_EDGE_OVERFLOW") != nullptr;
        // (8192^2 device ms, equal layers, profiles/r05_ps_runs.txt: ps_mfma_kernel 12.9 / 15.7 / 16.0 / 19.7 / 26.8 at 3 / 4 / 5 / 7 /
        // 11 long runs -- and 10.8 on the config-5 table; ps_runs_kernel, two wavenumbers per workgroup, 11.8 / 11.9 / 12.4 /
        // 12.8 / 13.8 at 3 / 4 / 5 / 7 / 11 long runs, 15.3 at 21, 21.3 at 42, 12.0 on the config-5 table)
        const bool runs_first = vlen != 0 && nlong > 3;
        // ... 6: only the transform path (ps_nufft.h) ahead of them.  By itself: the transform path for a constant velocity and
        // tables of up to 16 thick layers (8192^2 device ms at 3 / 5 / 7 / 11 / 16 / 21 long runs: 5.1 / 6.1 / 7.2 / 9.5 / 13.0 /
        // 16.8 against ps_runs_kernel's 11.8 / 12.4 / 12.8 / 13.7 / 14.4 / 15.3; config 5: 4.5 against ps_mfma_kernel's 10.6,
        // constant velocity 3.0 against 6.8 -- profiles/r05_ps_nufft.txt), then the matrix-core paths as before
        // ... 7: only the series path (ps_series.h).  By itself: profiles without runs of constant velocity to live on
        if (ok && vlen && (pref == 7 || (pref == 1 && !(P.sched && nufft_first) && (!P.sched || mruns.size() > 64))) && !force_overflow) {
            // (what would run otherwise, at 8192^2: ps_smooth32_kernel 5.3e-6 ms per alive pair; ps_runs_kernel 8 ms + 0.036 per run,
            // long or single step -- 41 / 81 / 161 table rows = 160 / 320 / 640 runs: 13.8 / 19.8 / 30.5 ms; a firn column's 1470: 70)
            const double alt = pref == 7 ? 0.0 : (!P.sched ? -SR_MS_PER_PAIR_F32
                                                           : 8.0 * ((double)nf / 4096.0) * ((double)snum / 8192.0) + 0.036 * (double)mruns.size() * ((double)nf / 4096.0));
            if ((rc = ps_series_run<float>(pl, P, vmig, kx, w.data(), thr.data(), st, &mfma_done, alt, tk_out == nullptr))) return rc;
            if (mfma_done) mfma_kernel_name = "ps_series_kernel";
        }
        if (ok && !mfma_done && (pref == 6 || (pref == 1 && (!vlen || (P.sched && nufft_first)))) && !force_overflow) {
            if ((rc = ps_nufft_run<float>(pl, P, mruns, vlen != 0, kx, w.data(), thr.data(), st, &mfma_done, nullptr, tk_out == nullptr))) return rc;
            if (!mfma_done && (rc = leave_half_front())) return rc;
            if (mfma_done) mfma_kernel_name = "ps_nufft_kernel";
        }
        for (int turn = 0; turn < 2 && ok && !mfma_done && pref != 0 && pref != 7; ++turn) {
            const bool use_runs = (turn == 0) == runs_first;
            if (use_runs) {
                if (pref == 2 || !vlen || force_overflow) continue;
                if ((rc = ps_runs_run(pl, P, mruns, kx, w.data(), thr.data(), st, &mfma_done))) return rc;
                if (mfma_done) mfma_kernel_name = "ps_runs_kernel";
            } else {
                if (pref == 3) continue;
                const char *name = "";
                if ((rc = ps_mfma_run(pl, P, mruns, vlen != 0, kx, thr.data(), st, &mfma_done, &name))) return rc;
                if (mfma_done) mfma_kernel_name = name;
            }
        }
    }
    if (!mfma_done && (rc = leave_half_front())) return rc;         // (whatever the dispatch did: no other kernel reads P.fhalf)
    // (made only when the runs kernels of the vector path are about to be launched: ~2 ms of host time with the GPU idle)
    if (!mfma_done && P.sched && (rc = order_rows())) return rc;
    bool smooth_done = false;
    if (!mfma_done && vlen && !P.sched) {
        // no runs of constant velocity to live on (the velocity changes in most 16-step tiles): ps_smooth_kernel
        bool ok = true;
        for (int i = 0; i < snum && ok; ++i) ok = std::isfinite(vmig[i]) && vmig[i] != 0.0 && thr[i] < 1e-10;
        if (ok) {
            // float32, the whole wavenumber axis with kx[tnum - k] = -kx[k]: rows k and tnum - k in one wave (ps_smooth.h)
            bool sm_pairs = P.k0 == 0 && nk == tnum && tnum >= 2;
            for (int k = 1; 2 * k < tnum && sm_pairs; ++k) sm_pairs = kx[k] == -kx[tnum - k];
            P.sm_m = ps_smooth_m(dbl, sm_pairs);
            P.sm_nchunks = ps_smooth_chunks(nf, P.sm_m);
            const size_t part_bytes = P.sm_nchunks > 1 ? (size_t)P.sm_nchunks * nk * snum * sizeof(Cp<T>) : 0;
            std::vector<double> &sm_step = pl.h_sm_step, &sm_tile = pl.h_sm_tile;     // (alive until the next call: async copies)
            if (sizeof(T) == 4) ps_smooth_tables(vmig, thr.data(), snum, sm_step, sm_tile);
            // (no room for the partial images: the per-step kernels below)
            if (pl.d_sm.ensure((sm_step.size() + sm_tile.size()) * 8 + 64) == hipSuccess &&
                (!part_bytes || pl.d_part.ensure(part_bytes) == hipSuccess)) {
                if (sizeof(T) == 4) {
                    IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_sm.p, sm_step.data(), sm_step.size() * 8, hipMemcpyHostToDevice, st));
                    IMPDAR_HIP_CHECK(hipMemcpyAsync(pl.d_sm.as<double>() + sm_step.size(), sm_tile.data(), sm_tile.size() * 8,
                                                    hipMemcpyHostToDevice, st));
                    P.sm_step = pl.d_sm.as<double>();
                    P.sm_tile = pl.d_sm.as<double>() + sm_step.size();
                }
                P.sm_part = pl.d_part.p;
                ps_smooth_launch<T>(P, st, sm_pairs);
                IMPDAR_HIP_CHECK(hipGetLastError());
                t_ps_kernel = sizeof(T) == 4 ? "ps_smooth32_kernel" : "ps_smooth_kernel";
                smooth_done = true;
            } else {
                (void)hipGetLastError();
                P.sm_nchunks = 1;
            }
        }
    }
    if (!mfma_done && !smooth_done && (rc = ps_dispatch<T>(P, st))) return rc;
    ctx->m_entry = tk_out ? "impdar_phaseshift_tk_dev" : "impdar_phaseshift";
    ctx->m_kernel = mfma_done ? mfma_kernel_name : t_ps_kernel;
    ctx->m_kernel_ms = -1.f;                 // (bracketed by ktic / ktoc)
    if (mfma_done && pl.mfma_instructions >= 0.0)      // (MFMA instructions the kernel issued, counted by the kernel: rounds and blocks it skips are not in it)
        snprintf(ctx->m_extra, sizeof ctx->m_extra,
                 "\"hermitian_walk\": %s, \"frequencies\": %d, \"transforms\": \"%s\", \"long_runs\": %d, \"mfma_instructions\": %.0f, \"flop_per_mfma\": %d",
                 herm ? "true" : "false", nf, use_own ? "own" : "rocfft", long_runs, pl.mfma_instructions,
                 strcmp(mfma_kernel_name, "ps_runs_kernel") == 0 ? 16384 : 32768);
    else
        snprintf(ctx->m_extra, sizeof ctx->m_extra, "\"hermitian_walk\": %s, \"frequencies\": %d, \"transforms\": \"%s\", \"long_runs\": %d, \"spectrum\": \"%s\"",
                 herm ? "true" : "false", nf, use_own ? "own" : "rocfft", long_runs, P.fhalf ? "k >= 0, all frequencies" : "all k, frequencies walked");
    if (herm)
        for (int kz : k_zero)
            if (kz >= k0 && kz < k0 + nk)
                hipLaunchKernelGGL((ps_dc_kernel<T>), dim3((snum + 255) / 256), dim3(256), 0, st, pl.X.as<Cp<T>>(),
                                   reinterpret_cast<Cp<T> *>(P.TK), kz, kz - k0, P.fstride, snum, w0 * dt);
    if ((rc = impdar_ctx_ktoc(ctx))) return rc;
    if (tk_out) {
        IMPDAR_HIP_CHECK(hipGetLastError());
        if ((rc = impdar_ctx_toc(ctx))) return rc;
        IMPDAR_HIP_CHECK(hipStreamSynchronize(st));
        return IMPDAR_OK;
    }
    if (pl.rows_form) {
        // inverse over k on contiguous rows: TK [k][tau] -> [tau][k] (pl.X: the spectrum is not needed any more, and
        // nt >= snum), transform, real part (:282) -- already (snum, tnum)
        if (use_own && tnum >= 32) {
            // (the real part is all that is kept: the Hermitian half of the sums goes through a real inverse transform of half
            // the length -- ps_transpose_herm; until round 6 a full transpose and a complex transform that stored real parts:
            // 175 + 314 us at 8192^2 float32)
            const int hs = tnum / 2 + 1;
            ps_launch_transpose_herm<T>(pl.TK.p, pl.X.p, tnum, snum, hs, st);
            if ((rc = own_fft_launch<T>(OWN_C2R, tnum, (size_t)snum, pl.X.p, d_out, (size_t)hs, (size_t)tnum, 1.0 / tnum, pl.tw_trace, st))) return rc;
        } else if (use_own) {
            ps_launch_transpose<T>(pl.TK.p, pl.X.p, tnum, snum, st);
            if ((rc = own_fft_launch<T>(OWN_C2C_INV_RE, tnum, (size_t)snum, pl.X.p, d_out, (size_t)tnum, (size_t)tnum, 1.0 / tnum, pl.tw_trace, st))) return rc;
        } else {
            ps_launch_transpose<T>(pl.TK.p, pl.X.p, tnum, snum, st);
            if ((rc = pl.b_trace.exec(pl.X.p, nullptr))) return rc;
            const size_t n = (size_t)snum * tnum;
            hipLaunchKernelGGL((ps_real_part<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pl.X.as<Cp<T>>(), (T *)d_out, n);
        }
    } else {
        if ((rc = pl.b_trace.exec(pl.TK.p, nullptr))) return rc;
        dim3 bgrid((tnum + 63) / 64, (snum + 63) / 64);
        hipLaunchKernelGGL((ps_real_transpose<T>), bgrid, dim3(256), 0, st, pl.TK.as<Cp<T>>(), (T *)d_out, snum, tnum);
    }
    IMPDAR_HIP_CHECK(hipGetLastError());
    if ((rc = impdar_ctx_toc(ctx))) return rc;
    impdar_trace("phaseshift: all kernels enqueued");
    // host staging vectors (w, thr) must outlive the async copies
    IMPDAR_HIP_CHECK(hipStreamSynchronize(st));
    return IMPDAR_OK;
}

extern "C" int impdar_phaseshift(impdar_ctx *ctx, const void *data, int dtype, int snum, int tnum, int nt,
                                 const double *kx, const double *ws, double dt, const double *tt_us, double vconst,
                                 const double *vmig, int vmig_len, double htaper, double vtaper, void *out)
{
    IMPDAR_ARG_CHECK(ctx && data && out && kx && ws && tt_us, "null argument");
    IMPDAR_ARG_CHECK(dtype == IMPDAR_F32 || dtype == IMPDAR_F64, "dtype must be 0 (f32) or 1 (f64)");
    IMPDAR_ARG_CHECK(snum >= 1 && tnum >= 1 && nt >= snum, "bad sizes snum %d tnum %d nt %d", snum, tnum, nt);
    IMPDAR_ARG_CHECK(vmig_len == 0 || (vmig_len == snum && vmig),
                     "Interpolated velocity profile is not the length of the number of samples in a trace.");
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)snum * tnum * impdar_dtype_size(dtype);
    impdar_trace("impdar_phaseshift: enter (%d x %d, nt %d)", snum, tnum, nt);
    impdar_ctx_pinned_prefetch(ctx, std::min(bytes, IMPDAR_STAGE_RING_BYTES));      // the download's staging ring, pinned while the call works
    // (the two device arrays of the call come from the cache of freed ones, api.hip: no hipMalloc / hipFree per call)
    struct Arrays {
        impdar_ctx *ctx;
        void *din = nullptr, *dout = nullptr;
        ~Arrays()
        {
            (void)hipStreamSynchronize(ctx->stream);          // (nothing in flight may still touch them)
            impdar_devcache_free(din);
            impdar_devcache_free(dout);
        }
    } a{ctx};
    int rc = impdar_devcache_alloc(ctx->device, bytes, &a.din);
    if (rc) return rc;
    if ((rc = impdar_devcache_alloc(ctx->device, bytes, &a.dout))) return rc;
    IMPDAR_HIP_CHECK(hipMemcpyAsync(a.din, data, bytes, hipMemcpyHostToDevice, ctx->stream));
    impdar_trace("impdar_phaseshift: upload enqueued");
    std::lock_guard<std::mutex> lk(g_ps_mu);
    ImpdarBusy busy(t_ps_busy);
    if (!g_ps_plan) g_ps_plan = new PsPlan();
    rc = dtype == IMPDAR_F32 ? ps_run<float>(ctx, *g_ps_plan, a.din, snum, tnum, nt, kx, ws, dt, tt_us, vconst, vmig,
                                             vmig_len, htaper, vtaper, a.dout)
                             : ps_run<double>(ctx, *g_ps_plan, a.din, snum, tnum, nt, kx, ws, dt, tt_us, vconst, vmig,
                                              vmig_len, htaper, vtaper, a.dout);
    if (rc) return rc;
    impdar_trace("impdar_phaseshift: device work complete");
    rc = impdar_download(ctx, out, a.dout, bytes, ctx->stream);
    impdar_trace("impdar_phaseshift: downloaded");
    return rc;
}

// resident form: d_data and d_out are device arrays of `dtype` (snum, tnum); runs on the context's compute stream
extern "C" int impdar_phaseshift_dev(impdar_ctx *ctx, const void *d_data, int dtype, int snum, int tnum, int nt,
                                     const double *kx, const double *ws, double dt, const double *tt_us, double vconst,
                                     const double *vmig, int vmig_len, double htaper, double vtaper, void *d_out)
{
    IMPDAR_ARG_CHECK(ctx && d_data && d_out && kx && ws && tt_us, "null argument");
    IMPDAR_ARG_CHECK(dtype == IMPDAR_F32 || dtype == IMPDAR_F64, "dtype must be 0 (f32) or 1 (f64)");
    IMPDAR_ARG_CHECK(snum >= 1 && tnum >= 1 && nt >= snum, "bad sizes snum %d tnum %d nt %d", snum, tnum, nt);
    IMPDAR_ARG_CHECK(vmig_len == 0 || (vmig_len == snum && vmig),
                     "Interpolated velocity profile is not the length of the number of samples in a trace.");
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> lk(g_ps_mu);
    ImpdarBusy busy(t_ps_busy);
    if (!g_ps_plan) g_ps_plan = new PsPlan();
    const int rc = dtype == IMPDAR_F32 ? ps_run<float>(ctx, *g_ps_plan, d_data, snum, tnum, nt, kx, ws, dt, tt_us, vconst, vmig,
                                                       vmig_len, htaper, vtaper, d_out)
                                       : ps_run<double>(ctx, *g_ps_plan, d_data, snum, tnum, nt, kx, ws, dt, tt_us, vconst, vmig,
                                                        vmig_len, htaper, vtaper, d_out);
    return rc ? rc : impdar_ctx_mark_produced(ctx);
}

// ---------------------------------------------------------------------------
// Sharding over the wavenumbers (optional row): every wavenumber is independent in phaseShift
// (mig_python.py:396-487), so rank r computes TK for its slab of k -- the cheap forward transforms are replicated on
// every rank from the whole radargram --, the slabs are redistributed into depth-row slabs by one all-to-all, and
// every rank finishes its rows with the inverse transform over k (:282).  Three calls per rank:
//   impdar_phaseshift_tk_dev      d_tk [nk][snum] complex: frequency sums of wavenumbers [k0, k0 + nk)
//   impdar_ps_alltoall_dev        d_tk -> d_t2 [tnum][tw] complex: ALL wavenumbers, this rank's depth rows [tau0, tau0 + tw)
//   impdar_phaseshift_finish_dev  d_t2 -> d_out (tw, tnum) real: ifft over k, real part
// ---------------------------------------------------------------------------
extern "C" int impdar_phaseshift_tk_dev(impdar_ctx *ctx, const void *d_data, int dtype, int snum, int tnum, int nt,
                                        const double *kx, const double *ws, double dt, const double *tt_us, double vconst,
                                        const double *vmig, int vmig_len, double htaper, double vtaper, int k0, int nk,
                                        void *d_tk)
{
    IMPDAR_ARG_CHECK(ctx && d_data && d_tk && kx && ws && tt_us, "null argument");
    IMPDAR_ARG_CHECK(dtype == IMPDAR_F32 || dtype == IMPDAR_F64, "dtype must be 0 (f32) or 1 (f64)");
    IMPDAR_ARG_CHECK(snum >= 1 && tnum >= 1 && nt >= snum, "bad sizes snum %d tnum %d nt %d", snum, tnum, nt);
    IMPDAR_ARG_CHECK(k0 >= 0 && nk >= 0 && k0 + nk <= tnum, "wavenumber slab [%d, %d) outside [0, %d)", k0, k0 + nk, tnum);
    IMPDAR_ARG_CHECK(vmig_len == 0 || (vmig_len == snum && vmig),
                     "Interpolated velocity profile is not the length of the number of samples in a trace.");
    if (nk == 0) return IMPDAR_OK;
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> lk(g_ps_mu);
    ImpdarBusy busy(t_ps_busy);
    if (!g_ps_plan) g_ps_plan = new PsPlan();
    const int rc = dtype == IMPDAR_F32 ? ps_run<float>(ctx, *g_ps_plan, d_data, snum, tnum, nt, kx, ws, dt, tt_us, vconst, vmig,
                                                       vmig_len, htaper, vtaper, nullptr, k0, nk, d_tk)
                                       : ps_run<double>(ctx, *g_ps_plan, d_data, snum, tnum, nt, kx, ws, dt, tt_us, vconst, vmig,
                                                        vmig_len, htaper, vtaper, nullptr, k0, nk, d_tk);
    return rc ? rc : impdar_ctx_mark_produced(ctx);
}

// rows [0, nk) x columns [tau0, tau0 + tw) of a [nk][snum] complex array -> contiguous [nk][tw]
template <typename T>
__global__ __launch_bounds__(256) void ps_pack_kernel(const Cp<T> *__restrict__ src, Cp<T> *__restrict__ dst, int nk, int snum,
                                                      int tau0, int tw)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)nk * tw) return;
    const size_t r = i / tw, j = i % tw;
    dst[i] = src[r * snum + tau0 + j];
}

// defined in comm.hip
int impdar_exchange_buffers(impdar_ctx *ctx, const void *sendbuf, void *recvbuf, int npeer, const int *peer, const size_t *soff,
                            const size_t *slen, const size_t *roff, const size_t *rlen, hipStream_t stream);

// `tau_edges` / `k_edges`: nranks + 1 edges of the depth-row and wavenumber slabs (the same on every rank).  Rank r
// sends block (its k slab) x (depth slab of s) to every rank s and receives (k slab of s) x (its depth slab), which
// lands at row k_edges[s] of d_t2 [tnum][tw]: the k-major layout the inverse transform wants, no unpacking.
extern "C" int impdar_ps_alltoall_dev(impdar_ctx *ctx, const void *d_tk, int dtype, int snum, int tnum, int nranks, int rank,
                                      const int *tau_edges, const int *k_edges, void *d_t2)
{
    IMPDAR_ARG_CHECK(ctx && d_tk && d_t2 && tau_edges && k_edges, "null argument");
    IMPDAR_ARG_CHECK(dtype == IMPDAR_F32 || dtype == IMPDAR_F64, "dtype must be 0 (f32) or 1 (f64)");
    IMPDAR_ARG_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d of %d", rank, nranks);
    IMPDAR_ARG_CHECK(tau_edges[0] == 0 && tau_edges[nranks] == snum && k_edges[0] == 0 && k_edges[nranks] == tnum,
                     "slab edges must run from 0 to snum / tnum");
    for (int s = 0; s < nranks; ++s)
        IMPDAR_ARG_CHECK(tau_edges[s + 1] >= tau_edges[s] && k_edges[s + 1] >= k_edges[s], "slab edges must not decrease");
    IMPDAR_ARG_CHECK(nranks == 1 || (ctx->comm && ctx->nranks == nranks), "communicator of %d ranks needed", nranks);
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t esz = 2 * impdar_dtype_size(dtype);
    const int nk = k_edges[rank + 1] - k_edges[rank], tw = tau_edges[rank + 1] - tau_edges[rank];
    std::lock_guard<std::mutex> lk(g_ps_mu);
    ImpdarBusy busy(t_ps_busy);
    if (!g_ps_plan) g_ps_plan = new PsPlan();
    PsPlan &pl = *g_ps_plan;
    IMPDAR_HIP_CHECK(pl.d_sendbuf.ensure(std::max<size_t>((size_t)nk * snum * esz, 16)));
    std::vector<int> peer(nranks);
    std::vector<size_t> soff(nranks), slen(nranks), roff(nranks), rlen(nranks);
    size_t at = 0;
    for (int s = 0; s < nranks; ++s) {
        const int tws = tau_edges[s + 1] - tau_edges[s], nks = k_edges[s + 1] - k_edges[s];
        peer[s] = s;
        soff[s] = at;
        slen[s] = (size_t)nk * tws * esz;
        roff[s] = (size_t)k_edges[s] * tw * esz;
        rlen[s] = (size_t)nks * tw * esz;
        if (nk > 0 && tws > 0) {
            const unsigned grid = (unsigned)(((size_t)nk * tws + 255) / 256);
            if (dtype == IMPDAR_F32)
                hipLaunchKernelGGL((ps_pack_kernel<float>), dim3(grid), dim3(256), 0, st, (const Cp<float> *)d_tk,
                                   reinterpret_cast<Cp<float> *>(pl.d_sendbuf.as<char>() + at), nk, snum, tau_edges[s], tws);
            else
                hipLaunchKernelGGL((ps_pack_kernel<double>), dim3(grid), dim3(256), 0, st, (const Cp<double> *)d_tk,
                                   reinterpret_cast<Cp<double> *>(pl.d_sendbuf.as<char>() + at), nk, snum, tau_edges[s], tws);
        }
        at += slen[s];
    }
    IMPDAR_HIP_CHECK(hipGetLastError());
    int rc;
    if (ctx->comm) {
        if ((rc = impdar_exchange_buffers(ctx, pl.d_sendbuf.p, d_t2, nranks, peer.data(), soff.data(), slen.data(), roff.data(),
                                          rlen.data(), st)))
            return rc;
    } else {
        // one rank, no communicator: the only block goes to itself
        IMPDAR_HIP_CHECK(hipMemcpyAsync(reinterpret_cast<char *>(d_t2) + roff[0], pl.d_sendbuf.as<char>() + soff[0], slen[0],
                                        hipMemcpyDeviceToDevice, st));
    }
    IMPDAR_HIP_CHECK(hipStreamSynchronize(st));
    return impdar_ctx_mark_produced(ctx);
}

extern "C" int impdar_phaseshift_finish_dev(impdar_ctx *ctx, void *d_t2, int dtype, int tw, int tnum, void *d_out)
{
    IMPDAR_ARG_CHECK(ctx && d_t2 && d_out, "null argument");
    IMPDAR_ARG_CHECK(dtype == IMPDAR_F32 || dtype == IMPDAR_F64, "dtype must be 0 (f32) or 1 (f64)");
    IMPDAR_ARG_CHECK(tw >= 0 && tnum >= 1, "bad sizes tw %d tnum %d", tw, tnum);
    if (tw == 0) return IMPDAR_OK;
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> lk(g_ps_mu);
    ImpdarBusy busy(t_ps_busy);
    if (!g_ps_plan) g_ps_plan = new PsPlan();
    PsPlan &pl = *g_ps_plan;
    const bool dbl = dtype == IMPDAR_F64;
    hipStream_t st = ctx->stream;
    int rc;
    if (pl.slab_key[0] != dtype || pl.slab_key[1] != tw || pl.slab_key[2] != tnum || pl.slab_owner != ctx) {
        const rocfft_array_type ci = rocfft_array_type_complex_interleaved;
        if ((rc = pl.b_slab.create(rocfft_transform_type_complex_inverse, dbl, true, tnum, tw, ci, ci, 1, tnum, 1, tnum, 1.0 / tnum, st)))
            return rc;
        pl.slab_key[0] = dtype, pl.slab_key[1] = tw, pl.slab_key[2] = tnum;
        pl.slab_owner = ctx;
    }
    // ifft over k (:282) on contiguous rows: [k][tau] -> [tau][k], transform, real part
    const size_t n = (size_t)tw * tnum;
    IMPDAR_HIP_CHECK(pl.d_slab.ensure(n * 2 * impdar_dtype_size(dtype)));
    if (dbl) ps_launch_transpose<double>(d_t2, pl.d_slab.p, tnum, tw, st);
    else ps_launch_transpose<float>(d_t2, pl.d_slab.p, tnum, tw, st);
    if ((rc = pl.b_slab.exec(pl.d_slab.p, nullptr))) return rc;
    if (dbl)
        hipLaunchKernelGGL((ps_real_part<double>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pl.d_slab.as<Cp<double>>(),
                           (double *)d_out, n);
    else
        hipLaunchKernelGGL((ps_real_part<float>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pl.d_slab.as<Cp<float>>(),
                           (float *)d_out, n);
    IMPDAR_HIP_CHECK(hipGetLastError());
    return impdar_ctx_mark_produced(ctx);
}

// ===========================================================================
// 2-D v(x,z): Fourier finite-difference branch of phaseShift
// (mig_python.py:428-432, 448-487; fourierFiniteDiff :496-525; Sp_Matr :528-540)
//
// The reference keeps ONE FFX_last for the whole (tau, omega) nest (:478): the update of
// frequency iw uses the field of frequency iw-1 (and of the last frequency of the previous tau
// at iw = 0), so the nest is a single serial chain of snum*nt steps, each with a length-tnum
// inverse and forward FFT over the traces.  That order is reproduced literally: per step
//   [post of the previous step + retardation phase of this row]  ->  rocFFT inverse (row, in place)
//   ->  [thin-lens phase + finite-difference update, one workgroup]  ->  rocFFT forward (into the row).
// float64 throughout, operation order of the reference (the file is compiled with
// -ffp-contract=off).  Arrays are (row = frequency, column = trace), traces contiguous.
// ===========================================================================
typedef Cp<double> Cd;

__device__ inline Cd cmul(Cd a, Cd b) { return Cd{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

// (snum,tnum) real -> tapered (:253-258), zero padded to nt rows, complex
__global__ __launch_bounds__(256) void ffd_load(const double *__restrict__ in, Cd *__restrict__ X, int snum, int tnum,
                                                int nt, double htaper, double vtaper)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)nt * tnum) return;
    const int k = (int)(i / tnum), j = (int)(i % tnum);
    double v = 0.0;
    if (k < snum) v = in[i] * (impdar_taper_w(j, tnum, htaper) * impdar_taper_w(k, snum, vtaper));
    X[i] = Cd{v, 0.0};
}

struct FfdStep {
    Cd *FK;              // (nt, tnum)
    Cd *TK;              // (snum, tnum)
    const double *kx;    // (tnum)
    const double *vmig;  // (snum, tnum)
    const double *vbg;   // (snum) min over traces
    const double *thr;   // (snum) (tau / tt[-1] / 1e6)^2, :484
    int tnum;
    // "post" half: row post_iw of step (post_itau): zero evanescent columns, accumulate into TK
    int post_itau, post_iw;
    double post_w;
    // "pre" half: retardation phase of row pre_iw at depth step pre_itau
    int pre_itau, pre_iw;
    double pre_w;
    double dt;
};

__device__ inline double ffd_coss(double vbg, double kx, double w)
{
    const double a = 0.5 * vbg * kx / w;          // :456
    return 1.0 - a * a;
}

__global__ __launch_bounds__(256) void ffd_post_pre(FfdStep P)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= P.tnum) return;
    if (P.post_iw >= 0) {
        Cd *row = P.FK + (size_t)P.post_iw * P.tnum;
        const double coss = ffd_coss(P.vbg[P.post_itau], P.kx[k], P.post_w);
        Cd v = row[k];
        if (coss <= P.thr[P.post_itau]) {          // :484-485 (sticky: written back)
            v = Cd{0.0, 0.0};
            row[k] = v;
        }
        Cd *t = P.TK + (size_t)P.post_itau * P.tnum + k;   // :487
        t->x += v.x;
        t->y += v.y;
    }
    if (P.pre_iw >= 0) {
        Cd *row = P.FK + (size_t)P.pre_iw * P.tnum;
        const double coss = ffd_coss(P.vbg[P.pre_itau], P.kx[k], P.pre_w);
        // phase = (-w dt sqrt(coss)).real with a complex sqrt: 0 for coss < 0 (:458-460)
        const double root = coss >= 0.0 ? sqrt(coss) : 0.0;
        const double phase = -P.pre_w * P.dt * root;
        double sn, cs;
        sincos(phase, &sn, &cs);
        row[k] = cmul(row[k], Cd{cs, -sn});        // conj(cos + i sin), :461-464
    }
}

struct FfdMid {
    const Cd *row;       // ifft of FK[iw] (tnum)
    const Cd *last;      // FFX_last (tnum)
    Cd *next;            // FFX of this step (tnum): becomes FFX_last and the forward FFT's input
    const double *vmig;  // row itau of (snum, tnum)
    double vbg, w, dt, dx;
    int tnum, first;     // first: itau == 0 (no finite-difference term, :475)
};

// one workgroup: the last stencil row is a sum over all traces
__global__ __launch_bounds__(256) void ffd_mid(FfdMid P)
{
    extern __shared__ Cd sm[];          // thin-lensed field (tnum) + 2 x 256 partial sums
    Cd *fx = sm;
    Cd *part = sm + P.tnum;
    const int tid = threadIdx.x, n = P.tnum;
    Cd sf = {0.0, 0.0}, sl = {0.0, 0.0};
    for (int x = tid; x < n; x += 256) {
        const double v = P.vmig[x];
        const double ufg = 1. / v - 1. / P.vbg;                            // :453
        const double phase2 = 2. * ufg * P.w * P.dt + 1. * P.vbg * P.w * P.dt;   // :471
        double sn, cs;
        sincos(phase2, &sn, &cs);
        const Cd f = cmul(P.row[x], Cd{cs, sn});                           // :472-473
        fx[x] = f;
        sf.x += f.x;
        sf.y += f.y;
        if (!P.first) {
            sl.x += P.last[x].x;
            sl.y += P.last[x].y;
        }
    }
    part[tid] = sf;
    part[256 + tid] = sl;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            part[tid].x += part[tid + s].x;
            part[tid].y += part[tid + s].y;
            part[256 + tid].x += part[256 + tid + s].x;
            part[256 + tid].y += part[256 + tid + s].y;
        }
        __syncthreads();
    }
    const Cd sumf = part[0], suml = part[256];
    for (int x = tid; x < n; x += 256) {
        Cd out = fx[x];
        if (!P.first) {
            // stencil = Sp_Matr(N,-2,1,1) as built: zero main diagonal (:533 overwrites it), row 0 = e0,
            // last row all ones (:536-539)
            Cd a, b;
            if (x == 0) {
                a = fx[0];
                b = P.last[0];
            } else if (x == n - 1) {
                a = sumf;
                b = suml;
            } else {
                a = Cd{fx[x - 1].x + fx[x + 1].x, fx[x - 1].y + fx[x + 1].y};
                b = Cd{P.last[x - 1].x + P.last[x + 1].x, P.last[x - 1].y + P.last[x + 1].y};
            }
            const double vs = P.vmig[x] - P.vbg;                           // :452
            const double num = P.dt * 0.5 * (vs * vs);                     // dt*alpha*vs**2, alpha = 0.5
            const double den = 4. * P.w * (P.dx * P.dx);                   // imaginary part of 1j*4.*w*dx**2
            // real / (0 + i den) the way NumPy divides complex numbers (Smith: scale by 1/den), :517
            const double scl = 1.0 / den;
            const Cd c1 = {0.0, (0.0 - num) * scl};
            const double c2 = (-0.25 * (vs * vs)) / (4. * (P.w * P.w) * (P.dx * P.dx));   // :518
            const Cd t1 = cmul(c1, a);
            const Cd d = {a.x - b.x, a.y - b.y};
            const Cd l = P.last[x];
            out = Cd{(l.x + t1.x) + c2 * d.x, (l.y + t1.y) + c2 * d.y};   // :521
        }
        P.next[x] = out;
    }
}

// ---------------------------------------------------------------------------
// The whole (tau, omega) chain in ONE workgroup (power-of-two trace counts up to 512).  The chain is serial by the
// reference's construction (one FFX_last for the whole nest), each step touches one row of tnum complex numbers,
// and issued as four launches per step it costs ~26 us a step, all of it launch / enqueue overhead.  Here the
// row lives in LDS for the whole step: retardation phase on load, inverse FFT over the traces (radix-4 Stockham,
// float64, twiddles from a table built on the host), thin-lens phase and finite-difference update against the
// FFX_last kept in LDS, forward FFT, evanescent zeroing, row written back; the sums into TK[itau] stay in
// registers over the frequencies of a depth step.  Same element formulas, in the same operation order, as
// ffd_post_pre / ffd_mid above; the transforms are its own instead of rocFFT's (rounding-level differences).
// LDS: four rows (two transform buffers, FFX_last, the thin-lensed field) + the twiddles = 4.5 tnum x 16 B.
// ---------------------------------------------------------------------------
struct FfdChain {
    Cd *FK;              // (nt, tnum) spectrum, updated in place
    Cd *TK;              // (snum, tnum)
    const double *kx;    // (tnum)
    const double *vmig;  // (snum, tnum)
    const double *vbg;   // (snum)
    const double *thr;   // (snum)
    const double *w;     // (nt), zero frequency already replaced
    const Cd *tw;        // (tnum / 2) exp(-2 pi i k / tnum)
    int tnum, snum, nt;
    double dt, dx;
};

// Stockham autosort transform of n (a power of two) points between two LDS buffers: radix-4 passes, one radix-2
// pass at the end when log2 n is odd (every pass is one barrier; with a handful of waves the barriers are most of
// a pass).  W[m] = exp(-2 pi i m / n), m < n / 2.  Returns the buffer that holds the result.
template <bool INV>
__device__ __forceinline__ Cd *ffd_fft_lds(Cd *in, Cd *out, const Cd *W, int n, int tid, int nthr)
{
    const int half = n >> 1, quarter = n >> 2;
    auto twiddle = [&](int m) {             // exp(-+ 2 pi i m / n) for m < n
        Cd t = W[m < half ? m : m - half];
        if (m >= half) t = Cd{-t.x, -t.y};
        if (INV) t.y = -t.y;
        return t;
    };
    int Ns = 1;
    for (; Ns * 4 <= n; Ns <<= 2) {
        const int wstep = quarter / Ns;
        for (int j = tid; j < quarter; j += nthr) {
            const int k = j & (Ns - 1);
            const int m1 = k * wstep;
            const Cd v0 = in[j];
            const Cd v1 = cmul(in[j + quarter], twiddle(m1));
            const Cd v2 = cmul(in[j + 2 * quarter], twiddle(2 * m1));
            const Cd v3 = cmul(in[j + 3 * quarter], twiddle(3 * m1));
            const Cd t0 = {v0.x + v2.x, v0.y + v2.y}, t1 = {v0.x - v2.x, v0.y - v2.y};
            const Cd t2 = {v1.x + v3.x, v1.y + v3.y};
            const Cd d = {v1.x - v3.x, v1.y - v3.y};
            const Cd t3 = INV ? Cd{-d.y, d.x} : Cd{d.y, -d.x};      // (v1 - v3) times +i / -i
            const int j0 = ((j - k) << 2) + k;
            out[j0] = Cd{t0.x + t2.x, t0.y + t2.y};
            out[j0 + Ns] = Cd{t1.x + t3.x, t1.y + t3.y};
            out[j0 + 2 * Ns] = Cd{t0.x - t2.x, t0.y - t2.y};
            out[j0 + 3 * Ns] = Cd{t1.x - t3.x, t1.y - t3.y};
        }
        __syncthreads();
        Cd *sw = in;
        in = out;
        out = sw;
    }
    if (Ns < n) {                           // Ns == n / 2
        for (int j = tid; j < half; j += nthr) {
            const int k = j & (Ns - 1);
            const Cd a = in[j], b = cmul(in[j + half], twiddle(k));      // angle -2 pi k / (2 Ns) = -2 pi k / n
            const int j0 = ((j - k) << 1) + k;
            out[j0] = Cd{a.x + b.x, a.y + b.y};
            out[j0 + Ns] = Cd{a.x - b.x, a.y - b.y};
        }
        __syncthreads();
        Cd *sw = in;
        in = out;
        out = sw;
    }
    return in;
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void ffd_chain_kernel(FfdChain P)
{
    extern __shared__ Cd sm[];
    const int n = P.tnum, tid = threadIdx.x;
    Cd *A = sm, *B = sm + n, *L = sm + 2 * n, *F = sm + 3 * n, *W = sm + 4 * n;
    Cd *part = W + (n >> 1);                 // 2 x (BLOCK / 64) partial sums
    constexpr int NW = BLOCK / 64;
    constexpr int EPT = 2;                   // BLOCK >= tnum / 2
    const double inv_n = 1.0 / n;
    for (int x = tid; x < n; x += BLOCK) L[x] = Cd{0.0, 0.0};
    for (int k = tid; k < (n >> 1); k += BLOCK) W[k] = P.tw[k];
    __syncthreads();
    for (int itau = 0; itau < P.snum; ++itau) {
        const double vbg = P.vbg[itau], thr = P.thr[itau];
        const double *vm = P.vmig + (size_t)itau * n;
        // what a depth step's frequencies share, per owned trace / wavenumber (the same expressions, evaluated once)
        Cd tk[EPT];
        double kxe[EPT], ufg[EPT], vs2[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int x = tid + e * BLOCK;
            tk[e] = Cd{0.0, 0.0};
            kxe[e] = x < n ? P.kx[x] : 0.0;
            const double v = x < n ? vm[x] : vbg;
            ufg[e] = 1. / v - 1. / vbg;                                            // :453
            const double vs = v - vbg;                                             // :452
            vs2[e] = vs * vs;
        }
        for (int iw = 0; iw < P.nt; ++iw) {
            const double w = P.w[iw];
            Cd *row = P.FK + (size_t)iw * n;
            // retardation phase (:456-464); coss is needed again after the forward transform (:484)
            double coss[EPT];
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int k = tid + e * BLOCK;
                coss[e] = ffd_coss(vbg, kxe[e], w);
                if (k < n) {
                    const double root = coss[e] >= 0.0 ? sqrt(coss[e]) : 0.0;
                    const double phase = -w * P.dt * root;
                    double sn, cs;
                    sincos(phase, &sn, &cs);
                    A[k] = cmul(row[k], Cd{cs, -sn});
                }
            }
            __syncthreads();
            Cd *R = ffd_fft_lds<true>(A, B, W, n, tid, BLOCK);          // :467, unscaled
            Cd *S = R == A ? B : A;
            // thin-lens term (:470-473) and the sums the last stencil row needs
            Cd sf = {0.0, 0.0}, sl = {0.0, 0.0};
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int x = tid + e * BLOCK;
                if (x < n) {
                    const double phase2 = 2. * ufg[e] * w * P.dt + 1. * vbg * w * P.dt;   // :471
                    double sn, cs;
                    sincos(phase2, &sn, &cs);
                    const Cd r = {R[x].x * inv_n, R[x].y * inv_n};
                    const Cd f = cmul(r, Cd{cs, sn});
                    F[x] = f;
                    sf.x += f.x;
                    sf.y += f.y;
                    if (itau > 0) {
                        sl.x += L[x].x;
                        sl.y += L[x].y;
                    }
                }
            }
#pragma unroll
            for (int msk = 32; msk > 0; msk >>= 1) {
                sf.x += __shfl_xor(sf.x, msk, 64);
                sf.y += __shfl_xor(sf.y, msk, 64);
                sl.x += __shfl_xor(sl.x, msk, 64);
                sl.y += __shfl_xor(sl.y, msk, 64);
            }
            if ((tid & 63) == 0) {
                part[tid >> 6] = sf;
                part[NW + (tid >> 6)] = sl;
            }
            __syncthreads();
            Cd sumf = {0.0, 0.0}, suml = {0.0, 0.0};
#pragma unroll
            for (int q = 0; q < NW; ++q) {
                sumf.x += part[q].x;
                sumf.y += part[q].y;
                suml.x += part[NW + q].x;
                suml.y += part[NW + q].y;
            }
            // finite-difference update (:496-525) -> S
            const double den = 4. * w * (P.dx * P.dx);                   // imaginary part of 1j*4.*w*dx**2
            const double scl = 1.0 / den;                                // real / (0 + i den), NumPy's way (:517)
            const double den2 = 4. * (w * w) * (P.dx * P.dx);
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int x = tid + e * BLOCK;
                if (x < n) {
                    Cd out = F[x];
                    if (itau > 0) {
                        Cd a, b;
                        if (x == 0) {
                            a = F[0];
                            b = L[0];
                        } else if (x == n - 1) {
                            a = sumf;
                            b = suml;
                        } else {
                            a = Cd{F[x - 1].x + F[x + 1].x, F[x - 1].y + F[x + 1].y};
                            b = Cd{L[x - 1].x + L[x + 1].x, L[x - 1].y + L[x + 1].y};
                        }
                        const double num = P.dt * 0.5 * vs2[e];                    // dt*alpha*vs**2, alpha = 0.5
                        const Cd c1 = {0.0, (0.0 - num) * scl};
                        const double c2 = (-0.25 * vs2[e]) / den2;                 // :518
                        const Cd t1 = cmul(c1, a);
                        const Cd d = {a.x - b.x, a.y - b.y};
                        const Cd l = L[x];
                        out = Cd{(l.x + t1.x) + c2 * d.x, (l.y + t1.y) + c2 * d.y};   // :521
                    }
                    S[x] = out;
                }
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int x = tid + e * BLOCK;
                if (x < n) L[x] = S[x];                                            // FFX_last = FFX, :478
            }
            Cd *Q = ffd_fft_lds<false>(S, R, W, n, tid, BLOCK);                   // :481
            // zero outside the domain, accumulate, keep the row (:484-487)
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int k = tid + e * BLOCK;
                if (k < n) {
                    Cd v = Q[k];
                    if (coss[e] <= thr) v = Cd{0.0, 0.0};
                    row[k] = v;
                    tk[e].x += v.x;
                    tk[e].y += v.y;
                }
            }
            // the next load writes A[k] for the same k this thread just read from Q: no barrier needed before it,
            // and its own barrier comes before anyone reads a neighbour's element
        }
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int k = tid + e * BLOCK;
            if (k < n) P.TK[(size_t)itau * n + k] = tk[e];
        }
    }
}

// TK (snum,tnum) complex -> out (snum,tnum) real: ifft over the traces done by rocFFT, /snum here (:491)
__global__ __launch_bounds__(256) void ffd_real_out(const Cd *__restrict__ Z, double *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = Z[i].x;
}

__global__ __launch_bounds__(256) void ffd_scale(Cd *__restrict__ Z, size_t n, double div)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        Z[i].x /= div;
        Z[i].y /= div;
    }
}

extern "C" int impdar_phaseshift_ffd(impdar_ctx *ctx, const double *data, int snum, int tnum, int nt, const double *kx,
                                     const double *ws, double dt, const double *tt_us, const double *vmig2d,
                                     double dx_mean, double htaper, double vtaper, double *out)
{
    IMPDAR_ARG_CHECK(ctx && data && out && kx && ws && tt_us && vmig2d, "null argument");
    IMPDAR_ARG_CHECK(snum >= 1 && tnum >= 2 && nt >= snum, "bad sizes snum %d tnum %d nt %d", snum, tnum, nt);
    IMPDAR_ARG_CHECK((size_t)tnum * 16 + 2 * 256 * 16 <= 150 * 1024, "the finite-difference step keeps a trace row in LDS: "
                     "tnum %d is above its 9000-trace limit", tnum);
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t nreal = (size_t)snum * tnum;
    DevBuf din, dout, X, TK, L, N, d_kx, d_vm, d_vbg, d_thr;
    IMPDAR_HIP_CHECK(din.ensure(nreal * 8));
    IMPDAR_HIP_CHECK(dout.ensure(nreal * 8));
    IMPDAR_HIP_CHECK(X.ensure((size_t)nt * tnum * 16));
    IMPDAR_HIP_CHECK(TK.ensure(nreal * 16));
    IMPDAR_HIP_CHECK(L.ensure((size_t)tnum * 16));
    IMPDAR_HIP_CHECK(N.ensure((size_t)tnum * 16));
    IMPDAR_HIP_CHECK(d_kx.ensure((size_t)tnum * 8));
    IMPDAR_HIP_CHECK(d_vm.ensure(nreal * 8));
    IMPDAR_HIP_CHECK(d_vbg.ensure((size_t)snum * 8));
    IMPDAR_HIP_CHECK(d_thr.ensure((size_t)snum * 8));
    std::vector<double> w(ws, ws + nt), thr(snum), vbg(snum);
    for (int i = 0; i < nt; ++i)
        if (w[i] == 0.0) w[i] = 1.0e-10 / dt;                            // :445-447
    for (int i = 0; i < snum; ++i) {
        const double tau = tt_us[i] / 1.0e6;
        const double r = tau / tt_us[snum - 1] / 1e6;                    // :484
        thr[i] = r * r;
        double m = vmig2d[(size_t)i * tnum];
        for (int j = 1; j < tnum; ++j) m = std::min(m, vmig2d[(size_t)i * tnum + j]);   // :450
        vbg[i] = m;
    }
    IMPDAR_HIP_CHECK(hipMemcpyAsync(din.p, data, nreal * 8, hipMemcpyHostToDevice, st));
    IMPDAR_HIP_CHECK(hipMemcpyAsync(d_kx.p, kx, (size_t)tnum * 8, hipMemcpyHostToDevice, st));
    IMPDAR_HIP_CHECK(hipMemcpyAsync(d_vm.p, vmig2d, nreal * 8, hipMemcpyHostToDevice, st));
    IMPDAR_HIP_CHECK(hipMemcpyAsync(d_vbg.p, vbg.data(), (size_t)snum * 8, hipMemcpyHostToDevice, st));
    IMPDAR_HIP_CHECK(hipMemcpyAsync(d_thr.p, thr.data(), (size_t)snum * 8, hipMemcpyHostToDevice, st));
    IMPDAR_HIP_CHECK(hipMemsetAsync(TK.p, 0, nreal * 16, st));
    IMPDAR_HIP_CHECK(hipMemsetAsync(L.p, 0, (size_t)tnum * 16, st));

    const rocfft_array_type ci = rocfft_array_type_complex_interleaved;
    FftPlan f_x, f_t, row_inv, row_fwd, tk_inv;
    int rc;
    // fft2(data, (nt, tnum)) (:270): over the traces (contiguous), then over time (stride tnum)
    if ((rc = f_x.create(rocfft_transform_type_complex_forward, true, true, tnum, nt, ci, ci, 1, tnum, 1, tnum, 1.0, st))) return rc;
    if ((rc = f_t.create(rocfft_transform_type_complex_forward, true, true, nt, tnum, ci, ci, tnum, 1, tnum, 1, 1.0, st))) return rc;
    if ((rc = row_inv.create(rocfft_transform_type_complex_inverse, true, true, tnum, 1, ci, ci, 1, tnum, 1, tnum, 1.0 / tnum, st))) return rc;
    if ((rc = row_fwd.create(rocfft_transform_type_complex_forward, true, false, tnum, 1, ci, ci, 1, tnum, 1, tnum, 1.0, st))) return rc;
    if ((rc = tk_inv.create(rocfft_transform_type_complex_inverse, true, true, tnum, snum, ci, ci, 1, tnum, 1, tnum, 1.0 / tnum, st))) return rc;

    const unsigned gall = (unsigned)(((size_t)nt * tnum + 255) / 256);
    hipLaunchKernelGGL(ffd_load, dim3(gall), dim3(256), 0, st, din.as<double>(), X.as<Cd>(), snum, tnum, nt, htaper, vtaper);
    if ((rc = f_x.exec(X.p, nullptr))) return rc;
    if ((rc = f_t.exec(X.p, nullptr))) return rc;

    // power-of-two trace counts up to 512: the whole chain in one persistent workgroup (IMPDAR_FFD_CHAIN=0 keeps the
    // launch-per-step form below, which also serves every other trace count).  Measured us per step, one workgroup /
    // launch per step (the latter varies by box): 64 traces 4.5 / 11-17, 128 traces 5.4 / 12-15, 512 traces 8.2 / 12.4-17.5,
    // 1024 traces 19 / 18.5-20, 2048 traces 37 / 28 -- one CU does all of a step's float64 sincos and butterflies and
    // the barriers of a wider workgroup cost more (one trace per thread: 28 us at 1024; four per thread: 11.8 us at 512),
    // so the wide rows stay with the spread-out form.
    const bool pow2 = tnum >= 2 && tnum <= 512 && (tnum & (tnum - 1)) == 0;
    const char *ce = getenv("IMPDAR_FFD_CHAIN");
    if (pow2 && !(ce && ce[0] == '0')) {
        DevBuf d_w, d_tw;
        IMPDAR_HIP_CHECK(d_w.ensure((size_t)nt * 8));
        IMPDAR_HIP_CHECK(d_tw.ensure((size_t)(tnum / 2) * 16));
        std::vector<Cd> twid(tnum / 2);
        for (int k = 0; k < tnum / 2; ++k) {
            const long double ang = -2.0L * 3.141592653589793238462643383279502884L * (long double)k / (long double)tnum;
            twid[k] = Cd{(double)cosl(ang), (double)sinl(ang)};
        }
        IMPDAR_HIP_CHECK(hipMemcpyAsync(d_w.p, w.data(), (size_t)nt * 8, hipMemcpyHostToDevice, st));
        IMPDAR_HIP_CHECK(hipMemcpyAsync(d_tw.p, twid.data(), twid.size() * 16, hipMemcpyHostToDevice, st));
        FfdChain C;
        C.FK = X.as<Cd>();
        C.TK = TK.as<Cd>();
        C.kx = d_kx.as<double>();
        C.vmig = d_vm.as<double>();
        C.vbg = d_vbg.as<double>();
        C.thr = d_thr.as<double>();
        C.w = d_w.as<double>();
        C.tw = d_tw.as<Cd>();
        C.tnum = tnum, C.snum = snum, C.nt = nt;
        C.dt = dt, C.dx = dx_mean;
        const int block = std::max(64, tnum / 2);
        const size_t lds = ((size_t)tnum * 4 + tnum / 2 + 2 * (block / 64)) * 16;
        auto launch = [&](auto kern) {
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(1), dim3(block), lds, st, C);
        };
        switch (block) {
        case 64: launch(ffd_chain_kernel<64>); break;
        case 128: launch(ffd_chain_kernel<128>); break;
        case 256: launch(ffd_chain_kernel<256>); break;
        case 512: launch(ffd_chain_kernel<512>); break;
        default: launch(ffd_chain_kernel<1024>); break;
        }
        IMPDAR_HIP_CHECK(hipGetLastError());
        const unsigned gtk2 = (unsigned)((nreal + 255) / 256);
        hipLaunchKernelGGL(ffd_scale, dim3(gtk2), dim3(256), 0, st, TK.as<Cd>(), nreal, (double)snum);     // :491
        if ((rc = tk_inv.exec(TK.p, nullptr))) return rc;                                                   // :282
        hipLaunchKernelGGL(ffd_real_out, dim3(gtk2), dim3(256), 0, st, TK.as<Cd>(), dout.as<double>(), nreal);
        IMPDAR_HIP_CHECK(hipGetLastError());
        IMPDAR_HIP_CHECK(hipMemcpyAsync(out, dout.p, nreal * 8, hipMemcpyDeviceToHost, st));
        IMPDAR_HIP_CHECK(hipStreamSynchronize(st));      // the staging vectors above live until here
        ctx->m_entry = "impdar_phaseshift_ffd";
        ctx->m_kernel = "ffd_chain_kernel";
        ctx->m_kernel_ms = -1.f;
        ctx->timed = ctx->ktimed = false;
        ctx->m_extra[0] = 0;
        return IMPDAR_OK;
    }

    FfdStep S;
    S.FK = X.as<Cd>();
    S.TK = TK.as<Cd>();
    S.kx = d_kx.as<double>();
    S.vmig = d_vm.as<double>();
    S.vbg = d_vbg.as<double>();
    S.thr = d_thr.as<double>();
    S.tnum = tnum;
    S.dt = dt;
    S.post_iw = -1;
    S.post_itau = 0;
    S.post_w = 1.0;
    const unsigned grow = (unsigned)((tnum + 255) / 256);
    const size_t mid_lds = ((size_t)tnum + 512) * 16;
    IMPDAR_HIP_CHECK(hipFuncSetAttribute((const void *)ffd_mid, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mid_lds));
    Cd *last = L.as<Cd>(), *next = N.as<Cd>();
    for (int itau = 0; itau < snum; ++itau)
        for (int iw = 0; iw < nt; ++iw) {
            S.pre_itau = itau;
            S.pre_iw = iw;
            S.pre_w = w[iw];
            hipLaunchKernelGGL(ffd_post_pre, dim3(grow), dim3(256), 0, st, S);
            Cd *row = X.as<Cd>() + (size_t)iw * tnum;
            if ((rc = row_inv.exec(row, nullptr))) return rc;                       // :467
            FfdMid M;
            M.row = row;
            M.last = last;
            M.next = next;
            M.vmig = d_vm.as<double>() + (size_t)itau * tnum;
            M.vbg = vbg[itau];
            M.w = w[iw];
            M.dt = dt;
            M.dx = dx_mean;
            M.tnum = tnum;
            M.first = itau == 0;
            hipLaunchKernelGGL(ffd_mid, dim3(1), dim3(256), mid_lds, st, M);
            if ((rc = row_fwd.exec(next, row))) return rc;                          // :481
            std::swap(last, next);                                                   // FFX_last = FFX, :478
            S.post_itau = itau;
            S.post_iw = iw;
            S.post_w = w[iw];
        }
    S.pre_iw = -1;
    hipLaunchKernelGGL(ffd_post_pre, dim3(grow), dim3(256), 0, st, S);
    IMPDAR_HIP_CHECK(hipGetLastError());
    const unsigned gtk = (unsigned)((nreal + 255) / 256);
    hipLaunchKernelGGL(ffd_scale, dim3(gtk), dim3(256), 0, st, TK.as<Cd>(), nreal, (double)snum);     // :491
    if ((rc = tk_inv.exec(TK.p, nullptr))) return rc;                                                  // :282
    hipLaunchKernelGGL(ffd_real_out, dim3(gtk), dim3(256), 0, st, TK.as<Cd>(), dout.as<double>(), nreal);
    IMPDAR_HIP_CHECK(hipGetLastError());
    IMPDAR_HIP_CHECK(hipMemcpyAsync(out, dout.p, nreal * 8, hipMemcpyDeviceToHost, st));
    IMPDAR_HIP_CHECK(hipStreamSynchronize(st));
    ctx->m_entry = "impdar_phaseshift_ffd";
    ctx->m_kernel = "ffd_post_pre + rocFFT + ffd_mid per (step, frequency)";
    ctx->m_kernel_ms = -1.f;
    ctx->timed = ctx->ktimed = false;
    ctx->m_extra[0] = 0;
    return IMPDAR_OK;
}

extern "C" int impdar_taper(impdar_ctx *ctx, void *data_inout, int dtype, int snum, int tnum, double htaper,
                            double vtaper)
{
    IMPDAR_ARG_CHECK(ctx && data_inout, "null argument");
    IMPDAR_ARG_CHECK(dtype == IMPDAR_F32 || dtype == IMPDAR_F64, "dtype must be 0 (f32) or 1 (f64)");
    IMPDAR_ARG_CHECK(snum >= 1 && tnum >= 1, "bad sizes snum %d tnum %d", snum, tnum);
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t n = (size_t)snum * tnum, bytes = n * impdar_dtype_size(dtype);
    DevBuf d;
    IMPDAR_HIP_CHECK(d.ensure(bytes));
    IMPDAR_HIP_CHECK(hipMemcpyAsync(d.p, data_inout, bytes, hipMemcpyHostToDevice, ctx->stream));
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (dtype == IMPDAR_F32)
        hipLaunchKernelGGL((ps_taper_inplace<float>), dim3(grid), dim3(256), 0, ctx->stream, d.as<float>(), snum, tnum,
                           htaper, vtaper);
    else
        hipLaunchKernelGGL((ps_taper_inplace<double>), dim3(grid), dim3(256), 0, ctx->stream, d.as<double>(), snum, tnum,
                           htaper, vtaper);
    IMPDAR_HIP_CHECK(hipGetLastError());
    IMPDAR_HIP_CHECK(hipMemcpyAsync(data_inout, d.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    IMPDAR_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->m_entry = "impdar_taper";
    ctx->m_kernel = "ps_taper_inplace";
    ctx->m_kernel_ms = -1.f;
    ctx->timed = ctx->ktimed = false;
    ctx->m_extra[0] = 0;
    return IMPDAR_OK;
}
