// Phase-shift frequency sum as a NON-UNIFORM FAST FOURIER TRANSFORM (float32 and float64 data, a few long runs of constant velocity;
// included by phaseshift.hip after own_fft.h and ps_mfma.h).
//
// Inside a run of constant velocity every frequency turns by a fixed angle phi_w per depth step, so what the reference
// accumulates step by step (mig_python.py:418-420, :464, :487) is a sum of exponentials sampled on the integers,
//     TK[tau0 + n] = sum_w  C_w e^{i phi_w (n + 1)},     C_w = FK_w e^{i Phi_w(tau0)},   n = 0 .. L - 1,
// a type-1 non-uniform discrete Fourier transform ("non-uniform frequencies phi_w in (0, 2 pi) -> uniform samples n").  The
// kernels of ps.hip / ps_mfma.h / ps_runs.h evaluate it directly, O(frequencies x steps) rotations per wavenumber -- on the
// vector units, or as a matrix product on the matrix cores.  Here it costs O(frequencies x W + G log G):
//   1. the coefficient of every frequency, turned to the middle of the piece (the transform's band is centred),
//      D_w = C_w e^{i phi_w (1 + Lp/2)}, and its place on a fine grid of G = 2 Lp points, u_w = phi_w G / 2 pi;
//   2. SPREADING: g[m] = sum_w D_w psi(u_w - m) with a window psi of W = 8 grid points ("exponential of semicircle",
//      psi(x) = exp(beta (sqrt(1 - (2x/W)^2) - 1)), beta = 2.30 W: Barnett, Magland, af Klinteberg 2019).  As a GATHER: a
//      thread owns grid points and walks the frequencies that reach them -- u_w grows with the slot index, and the range
//      comes from the dispersion relation itself, w(u) = sqrt((2 pi u / G dt)^2 + (v kx / 2)^2) -- so every sum has a fixed
//      order (no atomics: results are reproducible bit for bit);
//   3. one inverse FFT of length G in LDS (own_fft.h's passes);
//   4. TK[tau0 + n] = ghat[n - Lp/2] / psihat(n - Lp/2): the window's transform divided out (a table per Lp, host, float64
//      Simpson quadrature, once per plan).
// Error of the scheme with W = 8, twofold oversampling, float32 arithmetic: 3.5e-7 of the result (rel-L2; measured against a
// float64 direct sum on config-5 geometries before any of this was written) -- the matrix-core paths, with their float16
// hi/lo operands, are at 1.1e-6.
//
// Work split: one workgroup of 1024 threads per wavenumber -- per PAIR of wavenumbers (k, tnum - k) when the whole antisymmetric axis
// is summed and the sums go on into the inverse transform (template parameter PAIR, below: every migrate call on one GPU); a thread
// owns 4 frequencies with their phase in a float64 register and walks the pieces in depth order -- runs cut to <= 4096 steps (G <= 8192
// grid points: the coefficients and the spreading cost per PIECE, frequencies x 8 window values, the FFT per grid point: long
// pieces are the cheap ones; LDS by the call's longest piece, <= 126 KB, 159 KB for a pair).  The few single steps a layer boundary
// is smeared over are summed directly (a sincos per frequency and step, block reduction in a fixed order): a seventh of a piece
// each.  Which tables come here and which go to ps_runs_kernel: an estimate of both (ps_run, phaseshift.hip).  Frequencies on the
// evanescent boundary of some run take no part and are listed for ps_edge_kernel, as in ps_mfma.h / ps_runs.h; for a constant
// velocity the reference's own test decides them (:411-412).
// Measured at 8192^2: round 5 (profiles/r05_ps_nufft.txt) config 5 3.7 ms (ps_mfma_kernel 9.1), constant velocity 1.6 ms (5.3);
// round 6 with pairs (profiles/r06_transforms.txt) 1.64 and 0.90 ms; float64 data 6.7 and 4.0 ms.
#pragma once

constexpr int PN_NFMAX = 4096;              // frequencies per wavenumber this kernel takes (one workgroup holds them all)
constexpr int PN_SHORT = 8;                 // runs of up to this many steps are summed directly
// float32 data: a window of 8 grid points (3.5e-7 of the result in float32 arithmetic), 1024 threads;
// float64 data: 14 points (5e-13 in float64 arithmetic; the stated bar against the reference is 1e-10), 512 threads.
// A v(z) TABLE on float64 data: inside a "run" the interpolated velocity carries ~4e-13 of rounding noise (2 * gradient(z(t));
// the host cuts runs at 1e-11), which over 8192 steps moves a phase by 1e-8 rad -- the 1e-10 bar resolves that.  It enters as the
// first-order term of ps_series.h's series: with c^2 = cbar^2 + eps_t inside the piece (cbar^2 the piece's mean),
//     Phi(n) = (n + 1) phi_w - (dt / 2 psi_w) E(n),  E(n) = sum_{t <= n} eps_t,   e^{i Phi} = e^{i (n + 1) phi} (1 + i kappa_w E(n)),
// kappa_w = -dt / 2 psi_w (second order: 1e-15): a SECOND transform with the coefficients kappa_w D_w on the same nodes, scaled
// by i E(n) at the output.  The correction is 1e-8 of the sum, so its grid, its FFT and its products are FLOAT32 (1e-7 of 1e-8):
// one more accumulation per window value and 17 KB of LDS instead of a second float64 grid.
template <typename T> struct PnCfg;
// LMAX: steps per piece at most (G = 2 LMAX grid points): spreading and coefficients cost per PIECE -- as long as LDS allows
template <> struct PnCfg<float> { static constexpr int W = 8, NTH = 1024, OCC = 1, LMAX = 4096; };
template <> struct PnCfg<double> { static constexpr int W = 14, NTH = 512, OCC = 1, LMAX = 1024; };       // (1024 threads at 128 registers: 54 spilled, 9.34 -> 9.82 ms at 8192^2)

struct PnPiece {
    double v;               // velocity (kind 0)
    int start, len;         // first depth step, steps
    int kind;               // 0: transform, 1: direct sums (len <= PN_SHORT steps, each at its own velocity: the single steps of a
                            // smeared layer boundary, and short runs, taken in ONE pass over the frequencies)
    int loglp;              // kind 0: log2 of the padded length Lp >= len (G = 2 Lp)
    double vs[PN_SHORT];    // kind 1: the steps' velocities
};
struct PnParams {
    PsParams P;
    const PnPiece *pieces;
    int npieces;
    const double *rw;               // [nf] 1 / w, by slot
    const void *corr;               // the tables 1 / psihat (T), concatenated
    int corr_off[13];               // ... of Lp = 2^l at corr + corr_off[l], Lp/2 + 1 entries (|n - Lp/2| = 0 .. Lp/2)
    const void *tw[14];             // e^{-2 pi i k / G} (complex T), G = 2^l
    int *edge_cnt, *edge_list;      // v(z): boundary frequencies for ps_edge_kernel (null: constant velocity, none)
    int vz;
    int gmax;                       // grid points of the longest piece (the LDS layout)
    const double *e1;               // float64 v(z) tables: [snum] sum over the piece's steps up to this one of v_t^2 / vb2 - 1 (null: none)
};

// LDS: the grid of the call's longest piece (gmax points), then coefficients / grid places of all frequencies, the block reduction
// float64 pairs take the frequencies a quarter at a time (two float64 coefficient arrays of 4096 do not fit beside the grids), and
// with a quarter's arrays LDS has room for pieces of 2048 steps: half as many pieces, and a piece's cost -- coefficients, window
// values -- does not depend on its length
template <typename T> __host__ __device__ constexpr int pn_halves(bool pair) { return pair && sizeof(T) == 8 ? 4 : 1; }
template <typename T> __host__ __device__ constexpr int pn_lmax(bool pair) { return pair && sizeof(T) == 8 ? 2 * PnCfg<T>::LMAX : PnCfg<T>::LMAX; }
template <typename T> __host__ __device__ constexpr size_t pn_lds_bytes(int gmax, bool first_order = false, bool pair = false)
{
    // (coefficient, fraction and floor of the grid place: three arrays -- one 16-byte record per frequency and a ds_read_b128 per
    // window value measured 36 % SLOWER at config 5: 2.23 against 1.64 ms at a constant velocity, same box)
    return (size_t)(own_pad(gmax) + 1) * 2 * sizeof(T) + (size_t)(PN_NFMAX / pn_halves<T>(pair)) * ((pair ? 4 : 2) * sizeof(T) + sizeof(T) + 2) +
           16 * 2 * PN_SHORT * sizeof(T) + (first_order ? (size_t)(own_pad(gmax) + 1) * 2 * sizeof(float) : 0);
}

// the float32 grid of the first-order term: own_fft_passes<float, true> with the twiddles of the float64 table
__device__ __forceinline__ void pn_fft_f32(OCp<float> *s, int M, int logm, int tid, int nth, const OCp<double> *__restrict__ tw)
{
    int ll = logm;
    while (ll >= 2) {
        const int lq = ll - 2, q = 1 << lq, L = 1 << ll, tstep = M >> ll;
        for (int b = tid; b < (M >> 2); b += nth) {
            const int g = b >> lq, j = b & (q - 1), base = g * L + j;
            const OCp<float> a0 = s[own_pad(base)], a1 = s[own_pad(base + q)], a2 = s[own_pad(base + 2 * q)], a3 = s[own_pad(base + 3 * q)];
            const OCp<float> t0 = own_add(a0, a2), t1 = own_sub(a0, a2), t2 = own_add(a1, a3), d = own_sub(a1, a3);
            const OCp<float> t3 = OCp<float>{-d.y, d.x};
            const OCp<double> wd = tw[j * tstep];
            const OCp<float> w1{(float)wd.x, -(float)wd.y};
            const OCp<float> w2 = own_mul(w1, w1), w3 = own_mul(w2, w1);
            s[own_pad(base)] = own_add(t0, t2);
            s[own_pad(base + q)] = own_mul(own_add(t1, t3), w1);
            s[own_pad(base + 2 * q)] = own_mul(own_sub(t0, t2), w2);
            s[own_pad(base + 3 * q)] = own_mul(own_sub(t1, t3), w3);
        }
        __syncthreads();
        ll -= 2;
    }
    if (ll == 1) {
        for (int b = tid; b < (M >> 1); b += nth) {
            const OCp<float> a0 = s[own_pad(2 * b)], a1 = s[own_pad(2 * b + 1)];
            s[own_pad(2 * b)] = own_add(a0, a1);
            s[own_pad(2 * b + 1)] = own_sub(a0, a1);
        }
        __syncthreads();
    }
}

// psi(x) = exp(beta (sqrt(1 - (2x/W)^2) - 1)) for |x| <= W/2 and e^{-beta} (1e-8 / 1e-14 of its maximum) beyond: no select -- on
// the select and its compare cost the gather 5 % (1.72 -> 1.64 ms at config 5's size, constant velocity); the gather's index
// range keeps what lies beyond the support to two or three frequencies
struct PnWinF {                                                        // the window's two constants, in vector registers
    float c4, bl;                                                      // (a scalar-register or literal operand halves a v_fma_f32's rate)
    __device__ __forceinline__ PnWinF()
    {
        constexpr int W = PnCfg<float>::W;
        c4 = 4.0f / (W * W);
        bl = 2.30f * W * 1.4426950408889634f;                           // beta log2(e)
        asm volatile("" : "+v"(c4), "+v"(bl));
    }
};
__device__ __forceinline__ float pn_window(float x, const PnWinF &k)
{
    const float z = fmaxf(fmaf(-x * x, k.c4, 1.0f), 0.f);
    return __builtin_amdgcn_exp2f(fmaf(__builtin_amdgcn_sqrtf(z), k.bl, -k.bl));
}
__device__ __forceinline__ float pn_window(float x) { return pn_window(x, PnWinF()); }
__device__ __forceinline__ float pn_winT(float x, const PnWinF &k) { return pn_window(x, k); }
__device__ __forceinline__ double pn_window(double x);
__device__ __forceinline__ double pn_winT(double x, const PnWinF &) { return pn_window(x); }
// float64: the library's sqrt and exp (all ranges, ~65 float64 instructions together) replaced by what this argument range needs
// -- sqrt on (0, 1] from the float32 reciprocal root + two Newton steps (pm_sqrt01), exp on [-beta, 0] by y = k ln 2 + r,
// |r| <= 0.35, the series of e^r to r^12 (1.7e-16) and one ldexp: ~38 instructions; the window is what a float64 gather spends
// its time on (config 5 on float64 data 14.0 -> 13.6 ms in the frequency-sum kernels, a firn column 66 -> 63.4)
__device__ __forceinline__ double pn_window(double x)
{
    constexpr int W = PnCfg<double>::W;
    const double z = fmax(fma(-x * x, 4.0 / (W * W), 1.0), 1e-8);       // (below 1e-8: e^{-beta} (1 + 3e-3), 1e-14 of the maximum either way)
    const double y = (2.30 * W) * (pm_sqrt01(z) - 1.0);                  // in [-beta, 0]
    const double k = rint(y * 1.4426950408889634);
    double r = fma(-k, 6.93147180369123816490e-01, y);                  // ln 2 in two pieces (Cody-Waite)
    r = fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.0 / 479001600.0;
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}
__device__ __forceinline__ void pn_sincos(double x, float *s, float *c) { pm_sincos(x, s, c); }
__device__ __forceinline__ void pn_sincos(double x, double *s, double *c) { pss_sincos_small(pm_wrap(x), s, c); }

// PAIR (round 6): the wavenumbers k and tnum - k in ONE transform.  Only the real part of the inverse transform over the wavenumbers
// is kept (mig_python.py:282), so all that is needed of the rows TK[k], TK[tnum - k] is the Hermitian combination
//     G[k] = (TK[k] + conj TK[tnum - k]) / 2 = 1/2 sum_w [ F_w(k) e^{+i Phi_w} + conj F_w(tnum - k) e^{-i Phi_w} ]
// -- the phases depend on kx^2 only -- which is ONE non-uniform transform with the nodes +-phi_w: the regular frequencies' nodes
// fill the grid's half [0, G/2] and left the other half empty; the mirrored ones fill it.  A thread owns the grid points m and -m:
// the window value psi(u_w - m) serves a_w -> g[m] and b_w -> g[-m] (psi is even), the coefficients' sincos both rows', one FFT,
// one output pass -- G to row k, conj G to row tnum - k, so that everything downstream sees a Hermitian TK and reads the same
// image out of it.  The Nyquist row (w < 0: its node on the negative half) is, mirrored, the node that continues the positive
// frequencies' sorted order: index nf - 1 with the two coefficients' roles swapped, and no special case is left in the gather.
template <typename T, bool PAIR = false>
__global__ __launch_bounds__(PnCfg<T>::NTH, PnCfg<T>::OCC) void ps_nufft_kernel(PnParams Q)
{
    constexpr int PN_W = PnCfg<T>::W, PN_NTH = PnCfg<T>::NTH, PN_PER = PN_NFMAX / PN_NTH;
    constexpr int PN_NH = pn_halves<T>(PAIR), PN_NFH = PN_NFMAX / PN_NH;     // the frequencies whose coefficients LDS holds at a time
    extern __shared__ __attribute__((aligned(16))) unsigned char pn_lds[];
    const PsParams &P = Q.P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nf = P.nf;
    // small |kx| (few evanescent frequencies: the long workgroups) first
    // (PAIR: the whole axis, k = 0 .. tnum / 2 with its partner tnum - k; rows 0 and tnum / 2 are their own)
    const int bq = (int)blockIdx.x, kb = PAIR ? bq : ((bq & 1) ? P.nk - 1 - (bq >> 1) : (bq >> 1)), k = P.k0 + kb;
    const int k2 = PAIR ? (k == 0 ? 0 : P.nk - k) : k;
    OCp<T> *grid = reinterpret_cast<OCp<T> *>(pn_lds);                          // [own_pad(G)]
    OCp<T> *D = grid + own_pad(Q.gmax) + 1;                                         // [PN_NFH] coefficients, by index
    OCp<T> *D2 = D + (PAIR ? PN_NFH : 0);                                               // PAIR: [PN_NFH] the partner row's, mirrored
    T *fr = reinterpret_cast<T *>(D + (PAIR ? 2 : 1) * PN_NFH);                         // [PN_NFH] u - floor(u)
    unsigned short *m0 = reinterpret_cast<unsigned short *>(fr + PN_NFH);               // [PN_NFH] floor(u)
    T *red = reinterpret_cast<T *>(m0 + PN_NFH);                                        // [waves][2 PN_SHORT] block reduction
    OCp<float> *grid2 = reinterpret_cast<OCp<float> *>(red + 16 * 2 * PN_SHORT);        // [own_pad(G) + 1] the first-order term's grid (Q.e1)
    const bool fo = sizeof(T) == 8 && Q.e1 != nullptr;
    const Cp<T> *Frow = reinterpret_cast<const Cp<T> *>(P.F) + (size_t)k * P.fstride;
    T *TKrow = reinterpret_cast<T *>(reinterpret_cast<Cp<T> *>(P.TK) + (size_t)kb * P.snum);
    T *TKrow2 = reinterpret_cast<T *>(reinterpret_cast<Cp<T> *>(P.TK) + (size_t)k2 * P.snum);      // (PAIR: k0 = 0)
    const double kxk = P.kx[k];
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    const T inv_snum = (T)1 / (T)P.snum;

    // ---- this thread's frequencies: index i = tid + 256 j in ascending |w| (Hermitian walk: slot i + 1, the Nyquist row --
    // slot 0 -- last: P.w is by slot).  phase NaN = out of every run from here on (evanescent, boundary band, past nf)
    double ph[PN_PER];
    bool edge[PN_PER];
#pragma unroll
    for (int j = 0; j < PN_PER; ++j) {
        const int i = tid + PN_NTH * j;
        ph[j] = i < nf ? 0.0 : nan;
        edge[j] = false;
    }
    auto slot_of = [&](int i) { return i == nf - 1 ? 0 : i + 1; };
    // phase per depth step of frequency `slot` at velocity v; 0 and *alive = false where it is evanescent.  Constant velocity:
    // the reference's own test and expression (:411-415), as ps_setup_kernel; v(z): :456-460 off the boundary band
    auto step_phase = [&](int slot, double v, bool *alive) -> double {
        const double w = P.w[slot];
        if (!Q.vz) {
            // the reference's own test and expression (:411-415): a frequency ON the boundary (round velocities and spacings
            // produce them) is kept with a phase of ~1e-8 w dt -- deciding it from 1 - (v kx / 2w)^2 in another rounding dropped
            // it: 1.3e-2 of the image in two of 600 fuzz cases (profiles/r05_fuzz.txt)
            const double vk = v * kxk / 2.0, vkx2 = vk * vk;
            *alive = vkx2 < w * w;
            return *alive ? w * P.dt * sqrt(1.0 - vkx2 / (w * w)) : 0.0;
        }
        const double cs = pm_coss(v, kxk, Q.rw[slot]);
        *alive = cs > 0.0;
        return *alive ? w * P.dt * pm_sqrt01(cs) : 0.0;
    };
    if (Q.vz) {
        double vprev = 0.0;
        for (int r = 0; r < Q.npieces; ++r) {
            const int nv = Q.pieces[r].kind == 1 ? Q.pieces[r].len : 1;
            for (int q = 0; q < nv; ++q) {
                const double v = Q.pieces[r].kind == 1 ? Q.pieces[r].vs[q] : Q.pieces[r].v;
                if (v == vprev) continue;                                     // uniform
                vprev = v;
#pragma unroll
                for (int j = 0; j < PN_PER; ++j) {
                    const int i = tid + PN_NTH * j;
                    if (i < nf) edge[j] = edge[j] || fabs(pm_coss(v, kxk, Q.rw[slot_of(i)])) < 1e-8;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < PN_PER; ++j) {
            const int i = tid + PN_NTH * j;
            if (edge[j] && i < nf) {
                const int at = atomicAdd(Q.edge_cnt + k, 1);
                if (at < PM_EMAX) Q.edge_list[(size_t)k * PM_EMAX + at] = slot_of(i);
                if (PAIR && k2 != k) {                                        // the partner's list: the same frequencies
                    const int at2 = atomicAdd(Q.edge_cnt + k2, 1);
                    if (at2 < PM_EMAX) Q.edge_list[(size_t)k2 * PM_EMAX + at2] = slot_of(i);
                }
                ph[j] = nan;
            }
        }
    }
    // the frequency axis is uniform: w_i = (i + 1) dw (the Nyquist row, last, too) -- what the gather inverts
    const double dw_d = fabs(P.w[1]);
    const float inv_dw = (float)(1.0 / dw_d);

    for (int r = 0; r < Q.npieces; ++r) {
        const PnPiece pc = Q.pieces[r];
        const double v = pc.v;
        const int L = pc.len;
        if (pc.kind == 1) {
            // ---- single steps, summed directly: FK e^{i Phi} with Phi advanced by every step's own phase (:464, :487); a
            // frequency that turns evanescent at one of them is out from there on (:484-485)
            T acc[2 * PN_SHORT];
#pragma unroll
            for (int s_ = 0; s_ < 2 * PN_SHORT; ++s_) acc[s_] = 0;
#pragma unroll
            for (int j = 0; j < PN_PER; ++j) {
                const int i = tid + PN_NTH * j;
                if (i >= nf || !(ph[j] == ph[j])) continue;
                const int slot = slot_of(i);
                Cp<T> f = PAIR ? ps_load_slot_k<T>(P, k, slot) : ps_load_slot<T>(Frow, P, slot), fq = f;
                if (PAIR) {
                    // f1 e^{ip} + conj(f2) e^{-ip} = [(f1 + f2).x c - (f1 + f2).y s] + i [(f1 - f2).x s + (f1 - f2).y c]: the one sum's cost
                    const Cp<T> f2 = ps_load_slot_k<T>(P, k2, slot);
                    fq = Cp<T>{f.x - f2.x, f.y - f2.y};
                    f = Cp<T>{f.x + f2.x, f.y + f2.y};
                }
                double p = ph[j];
#pragma unroll
                for (int s_ = 0; s_ < PN_SHORT; ++s_)
                    if (s_ < L) {                                             // uniform
                        bool alive;
                        const double inc = step_phase(slot, pc.vs[s_], &alive);
                        if (!alive) p = nan;
                        p += inc;
                        if (p == p) {
                            T sn, c;
                            pn_sincos(p, &sn, &c);
                            acc[2 * s_] += fma(f.x, c, -(f.y * sn));
                            acc[2 * s_ + 1] += fma(fq.x, sn, fq.y * c);
                        }
                    }
                ph[j] = pm_wrap(p);                                           // NaN stays NaN
            }
#pragma unroll
            for (int s_ = 0; s_ < 2 * PN_SHORT; ++s_) {
                T x = acc[s_];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
                if (lane == 0) red[wave * 2 * PN_SHORT + s_] = x;
            }
            __syncthreads();
            if (tid < 2 * L) {
                T sum = 0;
#pragma unroll
                for (int q = 0; q < PN_NTH / 64; ++q) sum += red[q * 2 * PN_SHORT + tid];
                if (PAIR) {
                    sum *= (T)0.5;
                    if (k2 != k) TKrow2[2 * (size_t)(pc.start + (tid >> 1)) + (tid & 1)] = (tid & 1) ? -(sum * inv_snum) : sum * inv_snum;
                }
                TKrow[2 * (size_t)(pc.start + (tid >> 1)) + (tid & 1)] = sum * inv_snum;      // :492
            }
            __syncthreads();
            continue;
        }
        // ---- a piece of a long run: coefficients and grid places, spreading, inverse FFT, window divided out
        const int Lp = 1 << pc.loglp, G = 2 * Lp, logg = pc.loglp + 1;
        const double ug = (double)G * 0.15915494309189535;                    // G / 2 pi
        if (!PAIR)                                                            // (PAIR: the two gathers store every grid point)
            for (int m = tid; m < own_pad(G) + 1; m += PN_NTH) grid[m] = OCp<T>{(T)0, (T)0};
        // first-order term: kappa_w = -dt / 2 psi_w = kc / u_w (u = psi dt G / 2 pi: kc = -dt^2 G (v kx / 2)^2 ... / 4 pi), per unit of E / cbar^2
        const double c2d = 0.25 * v * v * kxk * kxk;
        const float kc = (float)(-P.dt * P.dt * (double)G * c2d / 12.566370614359172);
        // gather: grid point m takes the frequencies with |u_w - m| < W/2.  Their indices from the dispersion relation:
        // u = (G dt / 2 pi) sqrt(w^2 - c^2), c = v kx / 2, w = (i + 1) dw  ->  i(u) = sqrt((2 pi u / G dt)^2 + c^2) / dw - 1
        const float cq = (float)(0.5 * v * kxk), c2 = cq * cq;
        const float a = (float)(6.283185307179586 / ((double)G * P.dt)), a2 = a * a;
        const PnWinF wk_;
#pragma unroll
        for (int h = 0; h < PN_NH; ++h) {
            // (PN_NH = 4, float64 pairs: the frequencies a quarter at a time -- LDS holds the coefficients of 1024 pairs beside the grids)
            if (h) __syncthreads();
#pragma unroll
            for (int jj = 0; jj < PN_PER / PN_NH; ++jj) {
                const int j = h * (PN_PER / PN_NH) + jj;
                const int i = tid + PN_NTH * j, il = i - h * PN_NFH;
                if (i >= nf) continue;
                const int slot = slot_of(i);
                bool alive;
                const double inc = step_phase(slot, v, &alive);                   // (negative for the Nyquist row: w = -pi / dt)
                if (!alive) ph[j] = nan;
                OCp<T> d{(T)0, (T)0}, d2{(T)0, (T)0};
                if (ph[j] == ph[j]) {
                    const Cp<T> f = PAIR ? ps_load_slot_k<T>(P, k, slot) : ps_load_slot<T>(Frow, P, slot);
                    T sn, c;
                    pn_sincos(ph[j] + (double)(1 + Lp / 2) * inc, &sn, &c);
                    d = OCp<T>{fma(f.x, c, -(f.y * sn)), fma(f.x, sn, f.y * c)};
                    if (PAIR) {
                        const Cp<T> f2 = ps_load_slot_k<T>(P, k2, slot);        // the partner's, mirrored: conj(f2 e^{i theta})
                        d2 = OCp<T>{fma(f2.x, c, -(f2.y * sn)), -fma(f2.x, sn, f2.y * c)};
                    }
                    double adv = (double)L * inc;
                    if (fo) {
                        // kappa_w (signed with the frequency: inc = +- dt psi) times E at the piece's last step
                        const double kap = -P.dt * P.dt * c2d / (2.0 * inc);
                        adv += kap * Q.e1[pc.start + L - 1];
                        if (!PAIR && i == nf - 1) {                               // the Nyquist row's second coefficient (its u is wrapped)
                            float *dn2 = reinterpret_cast<float *>(red);
                            dn2[0] = (float)((double)d.x * kap);
                            dn2[1] = (float)((double)d.y * kap);
                        }
                    }
                    ph[j] = pm_wrap(ph[j] + adv);
                } else if (!PAIR && fo && i == nf - 1) {
                    float *dn2 = reinterpret_cast<float *>(red);
                    dn2[0] = dn2[1] = 0.f;
                }
                // place on the grid: phi mod 2 pi in units of the grid spacing (dead frequencies: 0 -- they carry D = 0)
                double u = inc * ug;
                if (PAIR) {
                    // every node by its |phi|: the Nyquist row (inc < 0) trades places with its mirror image
                    const bool neg = u < 0.0;
                    u = fabs(u);
                    D[il] = neg ? d2 : d;
                    D2[il] = neg ? d : d2;
                } else {
                    u -= (double)G * floor(u / (double)G);
                    D[il] = d;
                }
                const double fl = floor(u);
                m0[il] = (unsigned short)min((int)fl, G - 1);
                fr[il] = (T)(u - fl);
            }
            __syncthreads();
            if constexpr (PAIR) {
                // gather over the points mm = -W/2 .. G/2 + W/2 that the nodes in [0, G/2] reach: A = sum D psi(u - mm) is g[mm],
                // B = sum D2 psi(u - mm) is g[-mm].  The two index sets cover the grid and overlap in two strips (|mm| <= W/2 and
                // |mm - G/2| <= W/2): there B waits in registers for a barrier and is added; elsewhere it is stored at once (a second
                // half of the frequencies adds) -- every sum in a fixed order, as before.  First-order term: kappa = kc / u at the
                // node +u, -kc / u at its mirror image
                const int i0 = h * PN_NFH, i1 = min(nf, i0 + PN_NFH) - 1, NE = G / 2 + PN_W + 1;
                T s0x = 0, s0y = 0, s1x = 0, s1y = 0;
                float t0x = 0.f, t0y = 0.f, t1x = 0.f, t1y = 0.f;
                int at0 = -1, at1 = -1;
                for (int idx = tid; idx < NE; idx += PN_NTH) {
                    const int mi = idx - PN_W / 2;
                    const float mm = (float)mi;
                    T gx = 0, gy = 0, bx = 0, by = 0;
                    float hx = 0.f, hy = 0.f, b2x = 0.f, b2y = 0.f;
                    const float uhi = mm + 0.5f * PN_W, ulo = fmaxf(mm - 0.5f * PN_W, 0.f);
                    if (uhi > 0.f) {
                        // (w(u) / dw = i + 1: the frequencies with ulo <= u_i <= uhi are i = (int)(w(ulo)/dw) - 1 ... (int)(w(uhi)/dw) - 1 at the
                        // widest; one more on either side for the float32 estimate's rounding (1e-7 of up to 4096).  The one-row kernel below
                        // keeps round 5's two and three: every index beyond the support costs a window value and adds e^{-beta})
                        int ilo = (int)(__builtin_amdgcn_sqrtf(fmaf(a2 * ulo, ulo, c2)) * inv_dw) - 2;
                        int ihi = (int)(__builtin_amdgcn_sqrtf(fmaf(a2 * uhi, uhi, c2)) * inv_dw);
                        ilo = max(ilo, i0);
                        ihi = min(ihi, i1);
                        for (int i = ilo - i0; i <= ihi - i0; ++i) {
                            const T x = (T)((int)m0[i] - mi) + fr[i];
                            const T wgt = pn_winT(x, wk_);
                            const OCp<T> d = D[i], e = D2[i];
                            gx = fma(d.x, wgt, gx);
                            gy = fma(d.y, wgt, gy);
                            bx = fma(e.x, wgt, bx);
                            by = fma(e.y, wgt, by);
                            if (fo) {
                                const float wk = (float)wgt * kc * __builtin_amdgcn_rcpf(fmaxf((float)m0[i] + (float)fr[i], 1e-3f));
                                hx = fmaf((float)d.x, wk, hx);
                                hy = fmaf((float)d.y, wk, hy);
                                b2x = fmaf(-(float)e.x, wk, b2x);
                                b2y = fmaf(-(float)e.y, wk, b2y);
                            }
                        }
                    }
                    const int at = own_pad(mi & (G - 1)), atb = own_pad((-mi) & (G - 1));
                    const bool low = mi <= PN_W / 2, high = mi >= G / 2 - PN_W / 2;
                    if (h == 0) {
                        grid[at] = OCp<T>{gx, gy};
                        if (fo) grid2[at] = OCp<float>{hx, hy};
                        if (!low && !high) {
                            grid[atb] = OCp<T>{bx, by};
                            if (fo) grid2[atb] = OCp<float>{b2x, b2y};
                        }
                    } else {
                        const OCp<T> g = grid[at];
                        grid[at] = OCp<T>{g.x + gx, g.y + gy};
                        if (fo) {
                            const OCp<float> g2 = grid2[at];
                            grid2[at] = OCp<float>{g2.x + hx, g2.y + hy};
                        }
                        if (!low && !high) {
                            const OCp<T> gb = grid[atb];
                            grid[atb] = OCp<T>{gb.x + bx, gb.y + by};
                            if (fo) {
                                const OCp<float> g2 = grid2[atb];
                                grid2[atb] = OCp<float>{g2.x + b2x, g2.y + b2y};
                            }
                        }
                    }
                    if (low) {
                        s0x = bx, s0y = by, t0x = b2x, t0y = b2y, at0 = atb;
                    } else if (high) {
                        s1x = bx, s1y = by, t1x = b2x, t1y = b2y, at1 = atb;
                    }
                }
                __syncthreads();
                if (at0 >= 0) {
                    const OCp<T> g = grid[at0];
                    grid[at0] = OCp<T>{g.x + s0x, g.y + s0y};
                    if (fo) {
                        const OCp<float> g2 = grid2[at0];
                        grid2[at0] = OCp<float>{g2.x + t0x, g2.y + t0y};
                    }
                }
                if (at1 >= 0) {
                    const OCp<T> g = grid[at1];
                    grid[at1] = OCp<T>{g.x + s1x, g.y + s1y};
                    if (fo) {
                        const OCp<float> g2 = grid2[at1];
                        grid2[at1] = OCp<float>{g2.x + t1x, g2.y + t1y};
                    }
                }
            } else {
                const int ilast = nf - 2;                                         // regular frequencies: indices 0 .. nf - 2
                // the Nyquist row (index nf - 1): anywhere on the grid, looked at by every grid point
                const T uN = (T)m0[nf - 1] + fr[nf - 1];
                const OCp<T> dN = D[nf - 1];
                for (int m = tid; m < G; m += PN_NTH) {
                    const float mm = (float)(m > G / 2 + PN_W / 2 ? m - G : m);  // centred: the regular frequencies sit in [0, G/2], reach W/2 to either side (G >= 32 > 2 W)
                    T gx = 0, gy = 0;
                    float hx = 0.f, hy = 0.f;                                    // the first-order term's grid point (float32)
                    const float uhi = mm + 0.5f * PN_W, ulo = fmaxf(mm - 0.5f * PN_W, 0.f);
                    if (uhi > 0.f) {
                        int ilo = (int)(__builtin_amdgcn_sqrtf(fmaf(a2 * ulo, ulo, c2)) * inv_dw) - 3;
                        int ihi = (int)(__builtin_amdgcn_sqrtf(fmaf(a2 * uhi, uhi, c2)) * inv_dw) + 2;
                        ilo = max(ilo, 0);
                        ihi = min(ihi, ilast);
                        for (int i = ilo; i <= ihi; ++i) {
                            const T x = (T)((int)m0[i] - (int)mm) + fr[i];
                            const T wgt = pn_winT(x, wk_);
                            const OCp<T> d = D[i];
                            gx = fma(d.x, wgt, gx);
                            gy = fma(d.y, wgt, gy);
                            if (fo) {
                                const float wk = (float)wgt * kc * __builtin_amdgcn_rcpf(fmaxf((float)m0[i] + (float)fr[i], 1e-3f));
                                hx = fmaf((float)d.x, wk, hx);
                                hy = fmaf((float)d.y, wk, hy);
                            }
                        }
                    }
                    {
                        T x = uN - (T)m;
                        x -= (T)G * rint(x / (T)G);
                        const T wgt = pn_window(x);
                        gx = fma(dN.x, wgt, gx);
                        gy = fma(dN.y, wgt, gy);
                        if (fo) {
                            const float *dn2 = reinterpret_cast<const float *>(red);
                            hx = fmaf(dn2[0], (float)wgt, hx);
                            hy = fmaf(dn2[1], (float)wgt, hy);
                        }
                    }
                    grid[own_pad(m)] = OCp<T>{gx, gy};
                    if (fo) grid2[own_pad(m)] = OCp<float>{hx, hy};
                }
            }
        }
        __syncthreads();
        own_fft_passes<T, true>(grid, G, logg, tid, PN_NTH, reinterpret_cast<const OCp<T> *>(Q.tw[logg]), 1);
        if constexpr (sizeof(T) == 8) {
            if (fo) pn_fft_f32(grid2, G, logg, tid, PN_NTH, reinterpret_cast<const OCp<double> *>(Q.tw[logg]));
        }
        {
            const T *corr = reinterpret_cast<const T *>(Q.corr) + Q.corr_off[pc.loglp];
            for (int n = tid; n < L; n += PN_NTH) {
                const int np = n - Lp / 2;                                    // the band is centred: n' in [-Lp/2, Lp/2)
                const OCp<T> z = grid[own_pad(own_rev(np & (G - 1), G, logg))];
                const T cf = corr[np < 0 ? -np : np] * inv_snum;
                T ox = z.x * cf, oy = z.y * cf;
                if (fo) {
                    // + i E(n) * (second transform): (x + i y) i e = (-y e, x e)
                    const OCp<float> z2 = grid2[own_pad(own_rev(np & (G - 1), G, logg))];
                    const T e = (T)Q.e1[pc.start + n] * cf;
                    ox -= (T)z2.y * e;
                    oy += (T)z2.x * e;
                }
                if (PAIR) {
                    ox *= (T)0.5;
                    oy *= (T)0.5;
                    if (k2 != k) reinterpret_cast<Cp<T> *>(TKrow2)[pc.start + n] = Cp<T>{ox, -oy};
                }
                reinterpret_cast<Cp<T> *>(TKrow)[pc.start + n] = Cp<T>{ox, oy};
            }
        }
        __syncthreads();
    }
}
