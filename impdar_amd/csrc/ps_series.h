// Phase-shift frequency sum for a velocity that CHANGES inside a piece of the depth axis, as a few non-uniform FFTs that share
// their nodes (float32 and float64 data; included by phaseshift.hip after ps_nufft.h).
//
// Reference (mig_python.py:438-487): per depth step tau and frequency w of wavenumber kx
//     coss = 1 - (0.5 v_tau kx / w)^2;  FK[w] *= exp(i w dt sqrt(coss));  FK[w] = 0 for good once coss <= thr_tau;  TK[tau] += FK[w]
// i.e. TK[tau] = sum_w alive FK_w exp(i Phi_tau(w)), Phi_tau = sum_{t <= tau} dt sqrt(w^2 - c_t^2), c_t = v_t kx / 2.
// ps_nufft.h takes runs of CONSTANT velocity, where Phi is linear in the step: one type-1 non-uniform DFT per piece.  Here, on a
// piece [a, a + L) with c_t^2 = cbar^2 + eps_t (cbar^2 the piece's mean) and psi_w = sqrt(w^2 - cbar^2):
//     dt sqrt(w^2 - c_t^2) = dt psi - dt sum_m b_m eps_t^m psi^(1-2m),        b = 1/2, 1/8, 1/16, 5/128, ...   (sqrt(1 - x))
//     Phi(a + n) = Phi(a - 1) + (n + 1) dt psi + R(n, w),   R = -dt sum_m b_m E_m(n) psi^(1-2m),   E_m(n) = sum_{t = a .. a+n} eps_t^m
//     exp(i R) = sum_p y_p(n) z^p,   z = psi_min / psi <= 1          (the power series of exp of a polynomial: a recurrence per n)
// so that
//     TK[a + n] = sum_{p < J} y_p(n) * [ sum_w (C_w z_w^p) e^{i (n + 1) dt psi_w} ]
// -- J non-uniform DFTs with the SAME nodes dt psi_w (one set of window values serves all J; the spreading is a gather in a fixed
// order as in ps_nufft.h, J accumulators per grid point), J inverse FFTs in LDS, and per output step the J coefficients y_p(n)
// from the sums E_m(n) of the piece's velocity deviations (host tables, normalised).  The series in 1 / psi diverges at the
// evanescent boundary (psi -> 0): frequencies with psi < psi_min -- a narrow band above the boundary, chosen per (piece, kx) so
// that a majorant of the series' tail at z = 1 stays below a tolerance (1e-5 float32 data / 1e-11 float64: the relative error of
// the WORST frequency; the sums come out at 1e-7 ... 1e-6 / 1e-13) -- are summed DIRECTLY, step by step in float64 phases, with
// the reference's own expression for coss where it decides life and death (|coss| < 1e-12), and so is everything else that is
// alive and not regular (what fails the reference's test at a piece's first step is dead on arrival).  The Nyquist row of the
// Hermitian walk (w < 0: its phase runs backwards) is added per output step through the same series with conjugate coefficients.
// A frequency dies for good when coss <= thr at some step (:484-485); regular frequencies cannot (eps <= 0.1 psi^2 on them).
//
// Prototype and error table: profiles/tools/r06_series_proto.py, profiles/r06_series_proto.txt; measurements, stage by stage:
// profiles/r06_series.txt.  The library takes this path where the estimate from the planner's model beats the per-step kernels
// (phaseshift.hip): profiles whose velocity settles -- a firn column -- or falls; on a rising gradient every piece has a band of
// frequencies about to turn evanescent, and the direct sums there cost what the transforms save -- until a workgroup sums a PAIR of
// wavenumbers (float32, below), which halves everything but the gather.
//
// Work split: one workgroup of 1024 threads (float64: 512) per wavenumber; a thread owns 4 (8) frequencies (float64 phase in
// registers, NaN = dead) and walks the pieces in depth order.  Per piece: classification -> direct list (ranks by ballot, fixed
// order) -> direct sums (a wave per (group of 64 listed frequencies, chunk of 16 steps): lanes = frequencies, two passes -- the
// chunk's phase sum, then the phases, a sincos per (frequency, step) and a wave sum per step) -> coefficients / grid places of
// the regular frequencies -> gather -> J FFTs -> output.  Every sum has a fixed order: results are reproducible bit for bit.
#pragma once

#include "ps_series_plan.h"

template <typename T> struct SrCfg;
// NHALF: the regular frequencies go through LDS in this many rounds (float64: 36 bytes per frequency -- two rounds of 2048)
template <> struct SrCfg<float> { static constexpr int W = 8, GRID_BYTES = SR_GRID_BYTES, NHALF = 1, NTH = 1024; };
template <> struct SrCfg<double> { static constexpr int W = 14, GRID_BYTES = SR_GRID_BYTES, NHALF = 2, NTH = 512; };       // (256 registers a lane: J = 16 float64 accumulators)
static_assert(sr_pad(12345) == own_pad(12345), "the plan header pads LDS rows as own_fft.h does");

struct SrParams {
    PsParams P;
    const SrPiece *pieces;
    int npieces;
    const void *ev;                 // (T) the per-step tables of all pieces
    const double *rw;               // [nf] 1 / w, by slot
    const void *corr;               // 1 / psihat tables (T), as ps_nufft.h
    int corr_off[13];
    const void *tw[14];
    double kxh_max;
    int grid_bytes;                 // LDS of the grids (the launch's largest J * G)
};

constexpr int SR_TWLDS = 512;       // grid lengths up to this take their twiddles from LDS (a butterfly waits for nothing but LDS)

template <typename T> struct SrFq { T dx, dy, fr; float uh; };    // coefficient, u - floor(u), floor(u)
template <typename T> struct SrFq2 { T dx, dy, ex, ey, fr; float uh; };      // a pair of wavenumbers: + the partner's coefficient, mirrored (PAIR)
template <typename T, bool PAIR> using SrFqT = std::conditional_t<PAIR, SrFq2<T>, SrFq<T>>;
// PAIR: the regular frequencies go through LDS in twice as many rounds (a pair's record is 8 / 16 bytes longer)
template <typename T> __host__ __device__ constexpr int sr_nhalf(bool pair) { return SrCfg<T>::NHALF * (pair ? 2 : 1); }
struct SrDirect {
    double ph, w, rw;               // running phase (NaN once dead), frequency, 1 / w
    double fx, fy;                  // spectrum (weighted); PAIR: the sum of the two rows' ...
    double gx, gy;                  // PAIR: ... and their difference (f1 e^{ip} + conj(f2) e^{-ip} from the two, at one sum's cost)
};

// the Nyquist row of the Hermitian walk (w < 0: its phase runs backwards) while it is regular: added per output step through the
// same series with the conjugate coefficients -- F e^{i (ph0 - (n + 1) inc)} conj(sum_p y_p(n) z^p)
struct SrNyq {
    double ph0, inc;
    double fx, fy, z;
    double f2x, f2y;                // PAIR: the partner row's
    int valid, pad_;
};

template <typename T> __host__ __device__ constexpr size_t sr_lds_bytes(int grid_bytes, bool pair = false)
{
    return (size_t)(own_pad(SR_NFMAX / sr_nhalf<T>(pair)) + 1) * ((pair ? sizeof(SrFq2<T>) : sizeof(SrFq<T>)) + sizeof(T)) + (size_t)grid_bytes + 64 * sizeof(OCp<T>) +
           (size_t)SR_DMAX * sizeof(SrDirect) + 512 + SR_MSER * sizeof(double) + (size_t)SR_TWLDS * sizeof(OCp<T>);
}

__device__ __forceinline__ float sr_window(float x, const PnWinF &k) { return pn_window(x, k); }     // (ps_nufft.h: no select, e^{-beta} = 1e-8 beyond the support)
struct SrWinD {};
__device__ __forceinline__ double sr_window(double x, const SrWinD &) { return pn_window(x); }     // (ps_nufft.h)

// sqrt(x) and 1 / sqrt(x), x > 0 of any size: the hardware's float64 reciprocal root as the seed, two Newton steps (~1e-16 relative)
__device__ __forceinline__ double sr_sqrt(double x, double *rinv)
{
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    *rinv = y;
    return x * y;
}

// the sum of a value over the 64 lanes, in every lane: DPP inside a row of 16, gfx950's permlane swaps across rows (wrs_halve's)
template <typename T> __device__ __forceinline__ T sr_wave_sum(T v)
{
    v += lane_xor<T, 1>(v);
    v += lane_xor<T, 2>(v);
    v += lane_xor<T, 4>(v);
    v += lane_xor<T, 8>(v);
    if constexpr (sizeof(T) == 4) {
        const unsigned a = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane16_swap(a, a, false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        const unsigned b = __float_as_uint(v);
        const auto q = __builtin_amdgcn_permlane32_swap(b, b, false, false);
        v = __uint_as_float(q[0]) + __uint_as_float(q[1]);
    } else {
        {
            const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
            const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
            const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
            v = __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
        }
        {
            const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
            const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
            const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
            v = __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
        }
    }
    return v;
}

// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ double sr_wave_scan(double v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}

// the J FFTs of length G that sit side by side in LDS (grid j at s + j * gstride): own_fft_passes over all of them at once
template <typename T>
__device__ __forceinline__ void sr_fft_passes(OCp<T> *s, int J, int gstride, int M, int logm, int tid, const OCp<T> *tw, int tws)
{
    int ll = logm;
    const int lq4 = logm - 2;                                 // log2 of the butterflies per grid and radix-4 pass
    while (ll >= 2) {
        const int lq = ll - 2, q = 1 << lq, L = 1 << ll, tstep = (M >> ll) * tws;
        for (int bb = tid; bb < (J << lq4); bb += SrCfg<T>::NTH) {
            OCp<T> *g = s + (bb >> lq4) * gstride;
            const int b = bb & ((1 << lq4) - 1);
            const int gi = b >> lq, j = b & (q - 1), base = gi * L + j;
            const OCp<T> a0 = g[own_pad(base)], a1 = g[own_pad(base + q)], a2 = g[own_pad(base + 2 * q)], a3 = g[own_pad(base + 3 * q)];
            const OCp<T> t0 = own_add(a0, a2), t1 = own_sub(a0, a2), t2 = own_add(a1, a3), d = own_sub(a1, a3);
            const OCp<T> t3 = OCp<T>{-d.y, d.x};              // inverse: +i d
            OCp<T> w1 = tw[j * tstep];
            w1.y = -w1.y;
            const OCp<T> w2 = own_mul(w1, w1), w3 = own_mul(w2, w1);
            g[own_pad(base)] = own_add(t0, t2);
            g[own_pad(base + q)] = own_mul(own_add(t1, t3), w1);
            g[own_pad(base + 2 * q)] = own_mul(own_sub(t0, t2), w2);
            g[own_pad(base + 3 * q)] = own_mul(own_sub(t1, t3), w3);
        }
        __syncthreads();
        ll -= 2;
    }
    if (ll == 1) {
        for (int bb = tid; bb < (J << (logm - 1)); bb += SrCfg<T>::NTH) {
            OCp<T> *g = s + (bb >> (logm - 1)) * gstride;
            const int b = bb & ((1 << (logm - 1)) - 1);
            const OCp<T> a0 = g[own_pad(2 * b)], a1 = g[own_pad(2 * b + 1)];
            g[own_pad(2 * b)] = own_add(a0, a1);
            g[own_pad(2 * b + 1)] = own_sub(a0, a1);
        }
        __syncthreads();
    }
}

// the gather of one piece: grid point m (S threads each, a sub-range of the frequencies in reach apiece) <- J sums over the
// regular frequencies with |u_w - m| < W/2
// MIRROR (a pair of wavenumbers, second pass): the partner's coefficients, whose nodes are the mirror images -u_w -- the points
// mm = -W/2 .. G/2 + W/2 that the nodes reach, summed with psi(u_w - mm) as in the first pass and ADDED to grid point -mm; grid p
// takes (-z)^p: the series' coefficients y_p are real for even p and imaginary for odd p (the phase residual is odd in 1 / psi),
// so conj(sum_p y_p ghat_p) of the partner's row needs its odd grids negated and nothing else
template <typename T, int J, typename FQ, bool MIRROR = false>
__device__ __forceinline__ void sr_gather(const FQ *__restrict__ fq, const T *__restrict__ zz, OCp<T> *grids, int gstride, int G,
                                          int tid, int ifirst, int ilast, float a2, float c2, float inv_dw, bool add)
{
    constexpr int W = SrCfg<T>::W;
    // S threads per grid point (G < 1024), q fastest: the partial sums of a grid point sit in neighbouring lanes
    constexpr int NTH = SrCfg<T>::NTH;
    const std::conditional_t<sizeof(T) == 4, PnWinF, SrWinD> wk_;
    const int logs = G >= NTH ? 0 : __builtin_ctz(NTH / G), S = 1 << logs;
    const int npts = MIRROR ? G / 2 + W + 1 : G;
    for (int mb = 0; mb < npts; mb += (NTH >> logs)) {
        const int m = mb + (tid >> logs), q = tid & (S - 1);
        // centred: the regular frequencies sit in [0, G/2] (the Nyquist row is direct) and reach W/2 to either side (G >= 32 > 2 W)
        const float mm = MIRROR ? (float)(m - W / 2) : (float)(m > G / 2 + W / 2 ? m - G : m);
        T ax[J], ay[J];
#pragma unroll
        for (int p = 0; p < J; ++p) ax[p] = ay[p] = 0;
        const float uhi = mm + 0.5f * W, ulo = fmaxf(mm - 0.5f * W, 0.f);
        if (uhi > 0.f && m < npts) {
            int ilo = (int)(__builtin_amdgcn_sqrtf(fmaf(a2 * ulo, ulo, c2)) * inv_dw) - 3;
            int ihi = (int)(__builtin_amdgcn_sqrtf(fmaf(a2 * uhi, uhi, c2)) * inv_dw) + 2;
            ilo = max(ilo, ifirst);
            ihi = min(ihi, ilast);
            const int len = max(ihi - ilo + 1, 0), per = (len + S - 1) >> logs;
            const int i0 = ilo + q * per, i1 = min(i0 + per, ihi + 1);
            for (int i = i0; i < i1; ++i) {
                const FQ f = fq[own_pad(i - ifirst)];
                const T x = (T)(f.uh - mm) + f.fr;
                const T wgt = sr_window(x, wk_);
                T tx, ty;
                if constexpr (MIRROR) {
                    tx = f.ex * wgt;
                    ty = f.ey * wgt;
                } else {
                    tx = f.dx * wgt;
                    ty = f.dy * wgt;
                }
                ax[0] += tx;
                ay[0] += ty;
                if (J > 1) {
                    const T z = MIRROR ? -zz[own_pad(i - ifirst)] : zz[own_pad(i - ifirst)];
#pragma unroll
                    for (int p = 1; p < J; ++p) {
                        tx *= z;
                        ty *= z;
                        ax[p] += tx;
                        ay[p] += ty;
                    }
                }
            }
        }
        // the S partial sums of the grid point, in a fixed order (a butterfly over the low lane bits)
        for (int o = 1; o < S; o <<= 1) {
#pragma unroll
            for (int p = 0; p < J; ++p) {
                ax[p] += __shfl_xor(ax[p], o, 64);
                ay[p] += __shfl_xor(ay[p], o, 64);
            }
        }
        if (q == 0 && m < npts) {
            const int at = MIRROR ? own_pad((W / 2 - m) & (G - 1)) : own_pad(m);
#pragma unroll
            for (int p = 0; p < J; ++p) {
                OCp<T> *g = grids + p * gstride + at;
                *g = (add || MIRROR) ? OCp<T>{g->x + ax[p], g->y + ay[p]} : OCp<T>{ax[p], ay[p]};       // (this thread's own grid point in every round)
            }
        }
    }
}

// PAIR (round 6, as ps_nufft_kernel<T, true>): the wavenumbers k and tnum - k in one workgroup, summed as their Hermitian combination
// G = (TK[k] + conj TK[tnum - k]) / 2 -- all that the real part of the inverse transform over the wavenumbers keeps.  Classification,
// direct lists, the pieces' series and coefficients' phases depend on kx^2 only and are made once; the direct band costs one sum
// (SrDirect); the J grids take the partner's mirrored nodes in a second gather pass (sr_gather<..., MIRROR>); FFTs and the output pass
// with its y_p recurrences once.  G goes to row k, conj G to row tnum - k.
template <typename T, bool PAIR = false>
__global__ __launch_bounds__(SrCfg<T>::NTH, 1) void ps_series_kernel(SrParams Q)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sr_lds[];
    const PsParams &P = Q.P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nf = P.nf;
    // small |kx| (few evanescent frequencies: the long workgroups) first
    const int bq = (int)blockIdx.x, kb = PAIR ? bq : ((bq & 1) ? P.nk - 1 - (bq >> 1) : (bq >> 1)), k = P.k0 + kb;
    const int k2 = PAIR ? (k == 0 ? 0 : P.nk - k) : k;                                  // (PAIR: the whole axis, k0 = 0)
    constexpr int NTH = SrCfg<T>::NTH, NWV = NTH / 64, PER = SR_NFMAX / NTH, NHALF = sr_nhalf<T>(PAIR), HN = SR_NFMAX / NHALF, PERH = PER / NHALF;
    constexpr int NOUT = 2048 / NTH;                                                    // output steps per thread: pieces of up to 2048 steps
    using FQ = SrFqT<T, PAIR>;
    FQ *fq = reinterpret_cast<FQ *>(sr_lds);                                            // [own_pad(HN) + 1]
    T *zz = reinterpret_cast<T *>(fq + own_pad(HN) + 1);                                // [own_pad(HN) + 1]
    OCp<T> *grids = reinterpret_cast<OCp<T> *>(zz + own_pad(HN) + 1);             // J grids; scratch of the direct sums before
    OCp<T> *gend = reinterpret_cast<OCp<T> *>(reinterpret_cast<unsigned char *>(grids) + Q.grid_bytes);
    SrDirect *dl = reinterpret_cast<SrDirect *>(gend + 64);                             // [SR_DMAX]
    int *cnt = reinterpret_cast<int *>(dl + SR_DMAX);                                   // [PER * NWV = 64] compaction counts
    SrNyq *nyq = reinterpret_cast<SrNyq *>(cnt + 64);
    OCp<T> *twl = reinterpret_cast<OCp<T> *>(cnt + 128 + 2 * SR_MSER);                  // [SR_TWLDS] e^{-2 pi i k / SR_TWLDS}: the twiddles of the grids up to that length
    T *cmv = reinterpret_cast<T *>(gend);                                               // [SR_MJMAX] c_m of the piece (in the 64-element gap)
    const Cp<T> *Frow = reinterpret_cast<const Cp<T> *>(P.F) + (size_t)k * P.fstride;
    Cp<T> *TKrow = reinterpret_cast<Cp<T> *>(P.TK) + (size_t)kb * P.snum;
    Cp<T> *TKrow2 = reinterpret_cast<Cp<T> *>(P.TK) + (size_t)k2 * P.snum;
    const double kxk = P.kx[k], kxh = 0.5 * fabs(kxk);
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    const T inv_snum = (T)1 / (T)P.snum;
    const T *evt = reinterpret_cast<const T *>(Q.ev);

    // ---- this thread's frequencies: index i = tid + 1024 j in ascending |w| (Hermitian walk: slot i + 1; the Nyquist row --
    // slot 0, w < 0 -- is index nf - 1)
    // (w, 1 / w and the spectrum are re-read from L2 where a piece needs them: 24 registers that a piece's gather has better use for)
    double ph[PER];
    auto slot_of = [&](int i) { return i == nf - 1 ? 0 : i + 1; };
#pragma unroll
    for (int j = 0; j < PER; ++j) ph[j] = tid + NTH * j < nf ? 0.0 : nan;
    const double dw_d = fabs(P.w[1]);
    const float inv_dw = (float)(1.0 / dw_d);
    const int jk = min(max((int)ceil(kxh / Q.kxh_max * SR_NKX) - 1, 0), SR_NKX - 1);

    for (int m = tid; m < SR_TWLDS; m += NTH) twl[m] = reinterpret_cast<const OCp<T> *>(Q.tw[9])[m];
    static_assert(SR_TWLDS == 512, "Q.tw[9] is the table of 512 points");

    for (int r = 0; r < Q.npieces; ++r) {
        const SrPiece &pc = Q.pieces[r];
        auto ufi = [](int x) { return __builtin_amdgcn_readfirstlane(x); };
        auto ufd = [](double x) { return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x))); };
        const int pstart = ufi(pc.start), loglp = ufi(pc.loglp), mser = ufi(pc.mser), mj = ufi(pc.mj), ev_off = ufi(pc.ev_off);
        const int L = ufi(pc.len), J = ufi(pc.J), Lp = 1 << loglp, G = 2 * Lp, logg = loglp + 1, gstride = own_pad(G) + 1;
        const double lam = ufd((double)pc.lam[jk]), pcs = ufd(pc.s);
        const double psi_min = kxh * lam, pm2 = psi_min * psi_min, cb2 = kxh * kxh * ufd(pc.vb2);
        // ---- classification: regular (the transforms), direct (alive, not regular), dead.  An alive frequency that is not regular
        // and fails the reference's test at the piece's FIRST step dies there without a contribution (:484-487): most of what
        // is not regular, and all of the evanescent half of the plane at the first piece
        // (w = (i + 1) dw here and in the coefficients -- the host holds the axis to that within 1e-15 --; the direct sums and every
        // life-and-death test read the axis itself)
        bool dir[PER];
        unsigned regm = 0;
        const double v_first = P.vz[pstart], thr_first = P.thr[pstart];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int i = tid + NTH * j;
            const double wj = (double)(i + 1) * dw_d, w2 = wj * wj, p2 = w2 - cb2;
            const bool alive = ph[j] == ph[j];
            const bool reg = alive && p2 > pm2 && p2 > 4e-8 * w2;
            dir[j] = alive && !reg;
            if (dir[j]) {
                const int slot = slot_of(i);
                double cs = pm_coss(v_first, kxk, Q.rw[slot]);
                if (fabs(cs) < 1e-12) {
                    const double a = 0.5 * v_first * kxk / P.w[slot];
                    cs = 1.0 - a * a;
                }
                if (cs <= thr_first) {
                    dir[j] = false;
                    ph[j] = nan;
                }
            }
            regm |= reg ? 1u << j : 0u;
        }
        // ---- direct list: ranks in index order (j major, thread minor)
        int rank[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const unsigned long long b = __ballot(dir[j]);
            rank[j] = __popcll(b & ((1ull << lane) - 1ull));
            if (lane == 0) cnt[j * NWV + wave] = __popcll(b);
        }
        __syncthreads();
        int ndir = 0;
        {
            int before[PER];
#pragma unroll
            for (int j = 0; j < PER; ++j) before[j] = 0;
            for (int q = 0; q < PER * NWV; ++q) {
                const int c = cnt[q];
#pragma unroll
                for (int j = 0; j < PER; ++j)
                    if (q < j * NWV + wave) before[j] += c;
                ndir += c;
            }
#pragma unroll
            for (int j = 0; j < PER; ++j) rank[j] += before[j];
        }
        // ---- direct sums.  Passes of up to SR_DMAX listed frequencies in groups of 64: a wave takes (group, chunk of SR_DCH steps)
        // -- lanes = frequencies, the steps of a round shared out over the waves.  Pass 1: the chunk's phase increments (kept in
        // registers), their sum and the first dead step; pass 2, after the chunks' sums are exchanged through LDS: the phases, a
        // sincos per (frequency, step), and the sum over the lanes by a reduce-scatter of the 16 steps' (re, im).
        T dsx[NOUT], dsy[NOUT];                             // this thread's output steps n = tid + NTH jj
#pragma unroll
        for (int jj = 0; jj < NOUT; ++jj) dsx[jj] = dsy[jj] = 0;
        if (ndir > 0) {                                     // (uniform)
            constexpr int CSB = NWV * 64 * 12;              // chunk sums (double) + first dead steps (int) of every wave and lane
            double *csum = reinterpret_cast<double *>(grids);
            int *cdead = reinterpret_cast<int *>(csum + NWV * 64);
            OCp<T> *part = reinterpret_cast<OCp<T> *>(reinterpret_cast<unsigned char *>(grids) + CSB);     // [group][L]
            const int gfit = max(1, (Q.grid_bytes - CSB) / (L * (int)sizeof(OCp<T>)));      // (>= 1 by the host's sizing of grid_bytes)
            const int wv_u = __builtin_amdgcn_readfirstlane(wave);
            for (int base = 0; base < ndir; base += 64 * min(min(gfit, SR_DMAX / 64), NWV)) {
                const int gp = min(min(min(gfit, SR_DMAX / 64), NWV), (ndir - base + 63) >> 6), nd = min(ndir - base, 64 * gp);
                const int nch = NWV / gp, R = nch * SR_DCH;
#pragma unroll
                for (int j = 0; j < PER; ++j)
                    if (dir[j] && rank[j] >= base && rank[j] < base + nd) {
                        const int slot = slot_of(tid + NTH * j);
                        const Cp<T> f = PAIR ? ps_load_slot_k<T>(P, k, slot) : ps_load_slot<T>(Frow, P, slot);
                        SrDirect d;
                        d.ph = ph[j];
                        d.w = P.w[slot];
                        d.rw = Q.rw[slot];
                        d.fx = d.gx = (double)f.x;
                        d.fy = d.gy = (double)f.y;
                        if (PAIR) {
                            const Cp<T> f2 = ps_load_slot_k<T>(P, k2, slot);
                            d.fx = (double)(f.x + f2.x);
                            d.fy = (double)(f.y + f2.y);
                            d.gx = (double)(f.x - f2.x);
                            d.gy = (double)(f.y - f2.y);
                        }
                        dl[rank[j] - base] = d;
                    }
                for (int m = tid; m < gp * L; m += NTH) part[m] = OCp<T>{(T)0, (T)0};
                __syncthreads();
                const int g = wv_u / nch, c = wv_u - g * nch, e = 64 * g + lane;
                const bool mine = g < gp && e < nd;
                for (int n0 = 0; n0 < L; n0 += R) {
                    const int ns = n0 + c * SR_DCH;                       // this wave's first step of the round
                    double ph0 = nan, w = 1.0, rw = 1.0, sum = 0.0;
                    int dead = 0x7fffffff;
                    if (mine) {
                        ph0 = dl[e].ph;
                        w = dl[e].w;
                        rw = dl[e].rw;
                    }
                    const bool live = ph0 == ph0;                         // (dead in an earlier round: NaN)
                    // the step's phase increment w dt sqrt(coss) (:456-460) and the reference's test (:484); the same function in
                    // both passes: the increments are not kept (16 float64 registers a lane and the unrolled bodies spilled 170)
                    // (the chunk's velocities and thresholds: lane q loads step q's, the loop broadcasts them -- a scalar load per step
                    // made every step of both passes wait ~0.5 us for memory)
                    double v_l = 1.0, thr_l = 0.0;
                    if (lane < SR_DCH && ns + lane < L) {
                        v_l = P.vz[pstart + ns + lane];
                        thr_l = P.thr[pstart + ns + lane];
                    }
                    auto bcast = [&](double x, int q) {
                        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), q), __builtin_amdgcn_readlane(__double2loint(x), q));
                    };
                    auto step = [&](int q, bool *dies) -> double {
                        const double v = bcast(v_l, q);
                        double cs = pm_coss(v, kxk, rw);
                        if (fabs(cs) < 1e-12) {                           // the reference's own rounding where it decides
                            const double a = 0.5 * v * kxk / w;
                            cs = 1.0 - a * a;
                        }
                        *dies = cs <= bcast(thr_l, q);
                        if (cs >= 1e-8) return w * P.dt * pm_sqrt01(cs);
                        return cs > 0.0 ? w * P.dt * sqrt(cs) : 0.0;
                    };
                    if (g < gp) {
#pragma unroll 2
                        for (int q = 0; q < SR_DCH; ++q) {
                            const int n = ns + q;
                            if (n < L) {                                  // (uniform)
                                bool dies;
                                sum += step(q, &dies);
                                if (live && dies) dead = min(dead, n);
                            }
                        }
                        csum[wv_u * 64 + lane] = sum;
                        cdead[wv_u * 64 + lane] = dead;
                    }
                    __syncthreads();
                    if (g < gp) {
                        double start = ph0;
                        int dead_at = 0x7fffffff;
                        double total = 0.0;
                        for (int cc = 0; cc < nch; ++cc) {
                            const double sc = csum[(g * nch + cc) * 64 + lane];
                            if (cc < c) start += sc;
                            total += sc;
                            dead_at = min(dead_at, cdead[(g * nch + cc) * 64 + lane]);
                        }
                        const T fx = mine ? (T)dl[e].fx : (T)0, fy = mine ? (T)dl[e].fy : (T)0;
                        const T hx = mine ? (T)dl[e].gx : (T)0, hy = mine ? (T)dl[e].gy : (T)0;       // (PAIR: the difference of the rows; else the same)
                        T keepx = 0, keepy = 0;                            // lane q keeps step q's sum
#pragma unroll 2
                        for (int q = 0; q < SR_DCH; ++q) {
                            const int n = ns + q;
                            if (n >= L) break;                            // (uniform)
                            bool dies;
                            start += step(q, &dies);
                            T vx = 0, vy = 0;
                            if (mine && live && n < dead_at) {
                                T sn, cs_;
                                pn_sincos(start, &sn, &cs_);
                                vx = fma(fx, cs_, -(fy * sn));               // :464, :487
                                vy = fma(hx, sn, hy * cs_);
                            }
                            vx = sr_wave_sum<T>(vx);
                            vy = sr_wave_sum<T>(vy);
                            if (lane == q) {
                                keepx = vx;
                                keepy = vy;
                            }
                        }
                        if (lane < SR_DCH && ns + lane < L) part[(size_t)g * L + ns + lane] = OCp<T>{keepx, keepy};
                        // the phase after the round (the last chunk's wave holds the same total as every other; one writer)
                        if (mine && live && c == nch - 1) dl[e].ph = dead_at != 0x7fffffff ? nan : pm_wrap(ph0 + total);
                    }
                    __syncthreads();
                }
#pragma unroll
                for (int jj = 0; jj < NOUT; ++jj) {
                    const int n = tid + NTH * jj;
                    if (n < L)
                        for (int q = 0; q < gp; ++q) {
                            const OCp<T> v = part[(size_t)q * L + n];
                            dsx[jj] += v.x;
                            dsy[jj] += v.y;
                        }
                }
#pragma unroll
                for (int j = 0; j < PER; ++j)
                    if (dir[j] && rank[j] >= base && rank[j] < base + nd) ph[j] = dl[rank[j] - base].ph;
                __syncthreads();
            }
        }
        // ---- regular frequencies: coefficient at the middle of the piece, grid place, z = psi_min / psi; the phase at the end
        const double ug = (double)G * 0.15915494309189535;                    // G / 2 pi
        for (int m = tid; m < J * gstride; m += NTH) grids[m] = OCp<T>{(T)0, (T)0};
        if (tid < SR_MJMAX) {
            // c_m = -dt psi_min b_m rho^m, rho = kxh^2 s / psi_min^2 = s / lam^2: r_m(n) = c_m * (normalised E_m(n)) is the
            // coefficient of z^(2m-1) in R
            const double rho = lam > 0.0 ? pcs / (lam * lam) : 0.0;
            double bm = 0.5, c = -P.dt * psi_min * bm * rho;
            for (int m = 1; m <= tid; ++m) {
                const double bn = bm * (double)(2 * m - 1) / (double)(2 * m + 2);        // b_{m+1} = b_m (2m - 1) / (2m + 2)
                c *= rho * bn / bm;
                bm = bn;
            }
            cmv[tid] = (T)c;
        }
        const double ks = kxh * kxh * pcs;
#pragma unroll
        for (int half = 0; half < NHALF; ++half) {
            if (half > 0) __syncthreads();
#pragma unroll
            for (int jh = 0; jh < PERH; ++jh) {
                const int j = half * PERH + jh, il = tid + NTH * jh;
                FQ f{};
                T z = 0;
                const double wj = (double)(il + half * HN + 1) * dw_d, p2 = wj * wj - cb2;
                const bool reg = (regm >> j) & 1u;
                if (il + half * HN == nf - 1) {
                    // the Nyquist row: not on the grid (see SrNyq)
                    SrNyq q;
                    q.valid = 0;
                    q.ph0 = q.inc = q.fx = q.fy = q.z = q.f2x = q.f2y = 0.0;
                    if (reg) {
                        double rinv;
                        const double psi = sr_sqrt(p2, &rinv), inc = P.dt * psi;
                        const Cp<T> fs = PAIR ? ps_load_slot_k<T>(P, k, 0) : ps_load_slot<T>(Frow, P, 0);
                        q.valid = 1;
                        q.ph0 = ph[j];
                        q.inc = inc;
                        q.fx = (double)fs.x;
                        q.fy = (double)fs.y;
                        if (PAIR) {
                            const Cp<T> f2 = ps_load_slot_k<T>(P, k2, 0);
                            q.f2x = (double)f2.x;
                            q.f2y = (double)f2.y;
                        }
                        q.z = psi_min * rinv;
                        double ser = 0.0;
                        if (mser > 0) {
                            const double t = ks * rinv * rinv;
                            for (int m = mser - 1; m >= 0; --m) ser = fma(ser, t, pc.be[m]);
                            ser *= t;
                        }
                        ph[j] = pm_wrap(ph[j] - inc * ((double)L - ser));
                    }
                    *nyq = q;
                } else if (reg) {
                    double rinv;
                    const double psi = sr_sqrt(p2, &rinv), inc = P.dt * psi;
                    T sn, c;
                    pn_sincos(ph[j] + (double)(1 + Lp / 2) * inc, &sn, &c);
                    const Cp<T> fs = PAIR ? ps_load_slot_k<T>(P, k, slot_of(il + half * HN)) : ps_load_slot<T>(Frow, P, slot_of(il + half * HN));
                    f.dx = fma(fs.x, c, -(fs.y * sn));
                    f.dy = fma(fs.x, sn, fs.y * c);
                    if constexpr (PAIR) {
                        const Cp<T> f2 = ps_load_slot_k<T>(P, k2, slot_of(il + half * HN));      // the partner's, mirrored: conj(f2 e^{i theta})
                        f.ex = fma(f2.x, c, -(f2.y * sn));
                        f.ey = -fma(f2.x, sn, f2.y * c);
                    }
                    // the phase at the end of the piece: L dt psi - dt psi sum_m be_m t^m, t = kxh^2 s / psi^2 <= 0.1
                    double ser = 0.0;
                    if (mser > 0) {
                        const double t = ks * rinv * rinv;
                        for (int m = mser - 1; m >= 0; --m) ser = fma(ser, t, pc.be[m]);
                        ser *= t;
                    }
                    ph[j] = pm_wrap(ph[j] + inc * ((double)L - ser));
                    const double u = inc * ug, fl = floor(u);                     // in [0, G/2]
                    f.uh = (float)fl;
                    f.fr = (T)(u - fl);
                    z = (T)(psi_min * rinv);
                }
                fq[own_pad(il)] = f;
                zz[own_pad(il)] = z;
            }
            __syncthreads();
            {
                // gather: indices from the dispersion relation, u = (G dt / 2 pi) sqrt(w^2 - cbar^2), w = (i + 1) dw
                const float c2 = (float)cb2;
                const float a = (float)(6.283185307179586 / ((double)G * P.dt)), a2 = a * a;
                const int ifirst = half * HN, ilast = min(nf - 2, ifirst + HN - 1);
                const bool add = half > 0;
#define SR_GATHER(JJ)                                                                                                             \
    do {                                                                                                                          \
        sr_gather<T, JJ, FQ>(fq, zz, grids, gstride, G, tid, ifirst, ilast, a2, c2, inv_dw, add);                                 \
        if constexpr (PAIR) {                                                                                                     \
            __syncthreads();               /* (the mirrored points overlap the first pass's at both ends of the half grid) */    \
            sr_gather<T, JJ, FQ, true>(fq, zz, grids, gstride, G, tid, ifirst, ilast, a2, c2, inv_dw, true);                      \
        }                                                                                                                         \
    } while (0)
                switch (J) {
                case 1: SR_GATHER(1); break;
                case 2: SR_GATHER(2); break;
                case 4: SR_GATHER(4); break;
                case 6: SR_GATHER(6); break;
                case 8: SR_GATHER(8); break;
                case 12: SR_GATHER(12); break;
                default: SR_GATHER(16); break;
                }
#undef SR_GATHER
            }
        }
        __syncthreads();
        if (G <= SR_TWLDS) sr_fft_passes<T>(grids, J, gstride, G, logg, tid, twl, SR_TWLDS / G);
        else sr_fft_passes<T>(grids, J, gstride, G, logg, tid, reinterpret_cast<const OCp<T> *>(Q.tw[logg]), 1);
        {
            const T *corr = reinterpret_cast<const T *>(Q.corr) + Q.corr_off[loglp];
#pragma unroll 2
            for (int jj = 0; jj < NOUT; ++jj) {                                  // (not unrolled: 40 registers of series coefficients per step)
                const int n = tid + NTH * jj;
                if (n >= L) break;
                T dx = dsx[0], dy = dsy[0];
#pragma unroll
                for (int q = 1; q < NOUT; ++q)
                    if (jj == q) {
                        dx = dsx[q];
                        dy = dsy[q];
                    }
                const int np = n - Lp / 2;
                const int at = own_pad(own_rev(np & (G - 1), G, logg));
                const T cf = corr[np < 0 ? -np : np];
                // y_p(n): exp(i sum_m r_m z^(2m-1)) = sum_p y_p z^p,  p y_p = sum_{odd k <= p} k (i r_(k+1)/2) y_(p-k)
                T yx[SR_JMAX], yy[SR_JMAX], rm[SR_MJMAX];
                yx[0] = 1;
                yy[0] = 0;
                const OCp<T> g0 = grids[at];
                T sx = g0.x, sy = g0.y;
                const T zn = (T)nyq->z;
                T zp = 1, px = 1, py = 0;                                          // sum_p y_p zn^p
                if (J > 1) {
#pragma unroll
                    for (int m = 0; m < SR_MJMAX; ++m) rm[m] = m < mj ? (T)(2 * m + 1) * cmv[m] * evt[ev_off + (size_t)n * mj + m] : (T)0;   // k r_m
#pragma unroll
                    for (int p = 1; p < SR_JMAX; ++p) {
                        if (p < J) {                                           // (uniform)
                            T ax = 0, ay = 0;
#pragma unroll
                            for (int m = 0; 2 * m + 1 <= p; ++m) {             // k (i r) y = k r (-y.y, y.x)
                                ax = fma(-rm[m], yy[p - 2 * m - 1], ax);
                                ay = fma(rm[m], yx[p - 2 * m - 1], ay);
                            }
                            const T ip = (T)1 / (T)p;
                            yx[p] = ax * ip;
                            yy[p] = ay * ip;
                            const OCp<T> g = grids[p * gstride + at];
                            sx = fma(yx[p], g.x, fma(-yy[p], g.y, sx));
                            sy = fma(yx[p], g.y, fma(yy[p], g.x, sy));
                            zp *= zn;
                            px = fma(yx[p], zp, px);
                            py = fma(yy[p], zp, py);
                        }
                    }
                }
                if (nyq->valid) {
                    T sn, c;
                    pn_sincos(nyq->ph0 - (double)(n + 1) * nyq->inc, &sn, &c);
                    const T fx = (T)nyq->fx, fy = (T)nyq->fy;
                    const T ex = fma(fx, c, -(fy * sn)), ey = fma(fx, sn, fy * c);
                    dx += fma(ex, px, ey * py);                                    // e * conj(poly)
                    dy += fma(ey, px, -(ex * py));
                    if (PAIR) {
                        // + conj(e2 conj(poly)) = conj(e2) poly, e2 the partner's
                        const T f2x = (T)nyq->f2x, f2y = (T)nyq->f2y;
                        const T e2x = fma(f2x, c, -(f2y * sn)), e2y = fma(f2x, sn, f2y * c);
                        dx += fma(e2x, px, e2y * py);
                        dy += fma(e2x, py, -(e2y * px));
                    }
                }
                T ox = (sx * cf + dx) * inv_snum, oy = (sy * cf + dy) * inv_snum;                                    // :492
                if (PAIR) {
                    ox *= (T)0.5;
                    oy *= (T)0.5;
                    if (k2 != k) TKrow2[pstart + n] = Cp<T>{ox, -oy};
                }
                TKrow[pstart + n] = Cp<T>{ox, oy};
            }
        }
        __syncthreads();
    }
}

