// The pieces of a velocity profile for ps_series_kernel (ps_series.h): plain C++, no device code -- compiled into the library by
// phaseshift.hip and, by itself, into the CPU suite's checker of the scheme (tests/test_series_scheme.py).
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

constexpr int SR_NFMAX = 4096;      // frequencies per wavenumber (one workgroup holds them all)
constexpr int SR_MSER = 16;         // terms of the end-of-piece phase series at most
constexpr int SR_JMAX = 16;
constexpr int SR_MJMAX = SR_JMAX / 2;
constexpr int SR_NKX = 16;          // entries of a piece's cut table over the wavenumber axis
constexpr int SR_DMAX = 256;        // direct list entries per pass
constexpr int SR_DCH = 16;          // steps per wave and round of the direct sums (their phase increments stay in registers)

constexpr int SR_GRID_BYTES = 50 * 1024;   // LDS of a piece's J grids
constexpr int sr_pad(int i) { return i + (i >> 5); }

struct SrPiece {
    int start, len, loglp, J;       // J transforms (1: the velocity is constant inside the piece)
    int mser, mj;                   // terms of the phase series at the end of the piece; sums E_m per step (m <= mj = J / 2)
    int ev_off;                     // per-step table: ev[ev_off + n * mj + (m - 1)] = sum_{t <= n} ((v_t^2 - vb2) / s)^m
    int pad_;
    double vb2, s;                  // the piece's reference v^2 (mean) and max |v^2 - vb2|
    double be[SR_MSER];             // b_m * (normalised E_m at the last step): the phase at the end of the piece
    float lam[SR_NKX];              // psi_min = (|kx| / 2) * lam[j], j = the first entry with (j + 1) kxh_max / SR_NKX >= |kx| / 2
};

// ---------------------------------------------------------------------------------------------------------------------------
// Host side: the pieces of a velocity profile.
//
// A piece is (start, length L, J); what it costs per wavenumber is modelled in lane-nanoseconds -- a fixed part (barriers, the
// FFTs, the output steps), a part per regular frequency (coefficient + W window values with J accumulations each) and a part per
// (direct frequency, step) -- and the walk takes, from where it stands, the (L, J) with the lowest cost per step, judged at two
// wavenumbers (0.35 and 0.7 of the largest).  The cut of a (piece, wavenumber): the smallest lam (psi_min = kxh lam) for which
//   (a) max |eps| / psi_min^2 = s / lam^2 <= SR_RHO (the phase series at the end of the piece converges, no regular frequency
//       can turn evanescent), and
//   (b) the majorant of the dropped terms, sum_{p >= J} of the coefficients of exp(sum_m rhat_m z^(2m-1)) at z = 1 with
//       rhat_m = dt kxh lam b_m (s / lam^2)^m max_n |E~_m(n)|, is <= tol.
// ---------------------------------------------------------------------------------------------------------------------------
static inline double sr_rho(bool) { return 0.1; }     // (0.25 for float32 data: a narrower band, 16 terms of the phase series -- measured slower)
constexpr int SR_MMAJ = 12;          // sums E_m the majorant looks at

struct SrModel {                     // lane-ns
    double fix, cls, set, win, perj, dir, out, fft, tol;
    int W;
    double dir_measured;             // a direct pair as measured (the walk is steered by `dir`: pieces chosen with the measured figure ran slower)
};
static inline SrModel sr_model(bool dbl)
{
    return dbl ? SrModel{3.0e6, 30., 220., 80., 7.5, 300., 120., 6., 1e-11, 14, 300.} : SrModel{3.0e6, 20., 110., 20., 4., 150., 60., 3., 1e-5, 8, 400.};
}

// Measured at 8192 x 8192 on one MI355X (profiles/r06_series.txt): the kernel takes ~2.0e-7 (float32) / ~3.5e-7 (float64) ms per unit
// of the model and 8192 wavenumbers; the per-step kernels 5.3e-6 (float32) / 10.8e-6 (float64) ms per alive pair.  The library takes this path
// where the estimate says it wins by a margin (phaseshift.hip).
constexpr double SR_MS_PER_PAIR_F32 = 5.3e-6, SR_MS_PER_PAIR_F64 = 10.8e-6;
// The kernel's time per 8192 wavenumbers from the model's two parts -- the gathers (window values and accumulations: per wavenumber even
// when a workgroup sums a PAIR of them) and the rest (classification, direct band, coefficients, FFTs, output: once per workgroup) --
// least squares over the four 8192^2 profiles of profiles/r06_series.txt section 7 (rising / falling gradient, wavy, firn column):
//   float32, one row per workgroup   8.4e-7 gather + 1.43e-7 rest   -> 108.9 / 52.3 / 85.3 / 14.1 ms against 109 / 51 / 86 / 13.6 measured
//   float32, a pair per workgroup    8.4e-7 gather + 0.76e-7 rest   ->  76.5 / 40.4 / 66.3 /  9.7          76.4 / 40.1 / 66.5 / 10.4
//   float64, one row                 6.45e-7 gather + 2.24e-7 rest  -> 371 / 253 / 442 / 59                366 / 272 / 434 / 64
// (a single factor on the whole model, the first form, was up to 17 % off)
constexpr double SR_MS_GATHER_F32 = 8.4e-7, SR_MS_REST_F32 = 1.43e-7, SR_MS_REST_PAIR_F32 = 0.76e-7, SR_MS_GATHER_F64 = 6.45e-7, SR_MS_REST_F64 = 2.24e-7;
constexpr double SR_MARGIN = 0.9;        // the path is taken where its estimate is this share of the other kernel's or less

static inline double sr_b(int m)     // sqrt(1 - x) = 1 - sum_m b_m x^m
{
    double b = 0.5;
    for (int q = 1; q < m; ++q) b *= (double)(2 * q - 1) / (double)(2 * q + 2);
    return b;
}

struct SrStats {                     // of a candidate piece [a, a + L)
    double vb2 = 0, s = 0;
    double amax[SR_MMAJ] = {};       // max_n |E~_m(n)|, m = 1 ..
    double aend[SR_MSER] = {};       // E~_m(L - 1)
};

static void sr_stats(const double *v, int a, int L, SrStats *st, std::vector<double> *table, int mj)
{
    // (mean of v^2: a sum in extended precision would not change what the kernel is told -- vb2 is whatever this says)
    double sum = 0.0;
    for (int t = 0; t < L; ++t) sum += v[a + t] * v[a + t];
    st->vb2 = sum / L;
    bool flat = true;
    for (int t = 1; t < L; ++t) flat = flat && v[a + t] == v[a];
    if (flat) st->vb2 = v[a] * v[a];
    double s = 0.0;
    for (int t = 0; t < L; ++t) s = std::max(s, std::fabs(v[a + t] * v[a + t] - st->vb2));
    st->s = s;
    for (int m = 0; m < SR_MMAJ; ++m) st->amax[m] = 0.0;
    for (int m = 0; m < SR_MSER; ++m) st->aend[m] = 0.0;
    if (table) table->assign((size_t)L * mj, 0.0);
    if (s == 0.0) return;
    double run[SR_MSER] = {};
    const double is = 1.0 / s;
    for (int t = 0; t < L; ++t) {
        const double e = (v[a + t] * v[a + t] - st->vb2) * is;
        double pw = 1.0;
        for (int m = 0; m < SR_MSER; ++m) {
            pw *= e;
            run[m] += pw;
            if (m < SR_MMAJ) st->amax[m] = std::max(st->amax[m], std::fabs(run[m]));
            if (table && m < mj) (*table)[(size_t)t * mj + m] = run[m];
        }
    }
    for (int m = 0; m < SR_MSER; ++m) st->aend[m] = run[m];
}

// sum_{p >= J} of the coefficients of exp(sum_m rhat[m-1] z^(2m-1)) at z = 1
static double sr_majorant_tail(const double *rhat, int J)
{
    constexpr int P = SR_JMAX + 2 * SR_MMAJ + 8;
    double q[P] = {}, y[P];
    for (int m = 1; m <= SR_MMAJ; ++m)
        if (2 * m - 1 < P) q[2 * m - 1] = rhat[m - 1];
    y[0] = 1.0;
    double tail = 0.0;
    for (int p = 1; p < P; ++p) {
        double acc = 0.0;
        for (int k = 1; k <= p; k += 2) acc += (double)k * q[k] * y[p - k];
        y[p] = acc / p;
        if (p >= J) tail += y[p];
    }
    return tail;
}

// x with sum_{p >= J} x^p / p! = tol: where the first-order term alone would put the cut (the search below starts there)
static double sr_xj(int J, double tol)
{
    double lo = 0.0, hi = 64.0;
    for (int it = 0; it < 60; ++it) {
        const double x = 0.5 * (lo + hi);
        double term = 1.0, tail = 0.0;
        for (int p = 1; p < J + 60; ++p) {
            term *= x / p;
            if (p >= J) tail += term;
        }
        if (tail > tol) hi = x;
        else lo = x;
    }
    return lo;
}

// refine: bisection steps after the bracket (the plan's own tables: 6; candidate pieces of the walk: 0)
static double sr_cut(const SrStats &st, double dt, double kxh, int J, double tol, double xj, int refine, double SR_RHO)
{
    if (st.s == 0.0 || kxh == 0.0) return st.s == 0.0 ? 0.0 : std::sqrt(st.s / SR_RHO);
    if (J <= 1) return 1e300;                       // (a piece whose velocity changes has no J = 1 form)
    double bm[SR_MMAJ];
    for (int m = 1; m <= SR_MMAJ; ++m) bm[m - 1] = sr_b(m);
    auto tail = [&](double lam) {
        double rhat[SR_MMAJ];
        const double rho = st.s / (lam * lam);
        double pw = 1.0;
        for (int m = 0; m < SR_MMAJ; ++m) {
            pw *= rho;
            rhat[m] = dt * kxh * lam * bm[m] * pw * st.amax[m];
        }
        return sr_majorant_tail(rhat, J);
    };
    const double floor_ = std::sqrt(st.s / SR_RHO);
    // rhat_1 = dt kxh (s / lam) amax_1 / 2 <= xj
    double hi = std::max(floor_, 0.5 * dt * kxh * st.s * st.amax[0] / xj), lo = floor_;
    if (hi == floor_ && tail(hi) <= tol) return hi;
    for (int it = 0; it < 200 && tail(hi) > tol; ++it) {
        lo = hi;
        hi *= 1.12;
    }
    for (int it = 0; it < refine && lo < hi; ++it) {
        const double mid = std::sqrt(lo * hi);
        if (mid <= lo || mid >= hi) break;
        if (tail(mid) > tol) lo = mid;
        else hi = mid;
    }
    return hi;
}

struct SrHostPlan {
    std::vector<SrPiece> pieces;
    std::vector<double> ev;          // per-step tables (float64; converted for float32 data at upload)
    double kxh_max = 0;
    int grid_bytes = 0;
    double model_cost = 0;           // the walk's modelled cost of the whole profile (lane-ns per wavenumber, mean of the two judged)
    double model_gather = 0;         // ... of which the gathers' window values and accumulations: what a PAIR of wavenumbers does not share
    double alive_pairs = 0;          // (frequency, step) pairs that are alive, mean of the same two wavenumbers: what a per-step kernel walks
    // what it was made from
    std::vector<double> v;
    double dt = 0, dw = 0;
    int nf = 0;
    bool dbl = false;
};

// L the grids of J transforms fit: J * (sr_pad(2 Lp) + 1) complex numbers
static inline bool sr_fits(int Lp, int J, size_t csize, int grid_bytes) { return (size_t)J * (sr_pad(2 * Lp) + 1) * csize <= (size_t)grid_bytes; }

static bool sr_make_plan(SrHostPlan &hp, const double *v, int snum, double dt, double dw, int nf, double kxh_max, bool dbl)
{
    const SrModel M = sr_model(dbl);
    const double SR_RHO = sr_rho(dbl);
    const size_t csize = dbl ? 16 : 8;
    const int grid_budget = SR_GRID_BYTES;
    hp.pieces.clear();
    hp.ev.clear();
    hp.kxh_max = kxh_max;
    hp.grid_bytes = 0;
    hp.model_cost = 0;
    hp.model_gather = 0;
    hp.alive_pairs = 0;
    hp.v.assign(v, v + snum);
    hp.dt = dt;
    hp.dw = dw;
    hp.nf = nf;
    hp.dbl = dbl;
    static const int JS[] = {2, 4, 6, 8, 12, 16};
    double xj[6];
    for (int ji = 0; ji < 6; ++ji) xj[ji] = sr_xj(JS[ji], M.tol);
    const double kreps[2] = {0.35 * kxh_max, 0.7 * kxh_max};
    double vmax_prev = 0.0;                            // running maximum of |v|: frequencies below kxh vmax are dead for good
    int a = 0;
    while (a < snum) {
        int bestL = 0, bestJ = 0;
        double best = 1e300, best_est = 0.0, best_gather = 0.0;
        // candidate lengths: powers of two, the rest of the record, and the end of the run of constant velocity we stand in
        int cand[16], nc = 0;
        for (int L = 32; L <= 2048 && nc < 12; L *= 2) cand[nc++] = std::min(L, snum - a);
        {
            int e = a + 1;
            while (e < snum && v[e] == v[a]) ++e;
            if (e - a >= 32) cand[nc++] = std::min(e - a, 2048);
        }
        for (int ci = 0; ci < nc; ++ci) {
            const int L = cand[ci];
            bool dup = false;
            for (int cj = 0; cj < ci; ++cj) dup = dup || cand[cj] == L;
            if (dup) continue;
            int l = 4;
            while ((1 << l) < L) ++l;
            SrStats st;
            sr_stats(v, a, L, &st, nullptr, 0);
            const int G = 2 << l;
            for (int ji = -1; ji < 6; ++ji) {
                const int J = ji < 0 ? 1 : JS[ji];
                if ((J == 1) != (st.s == 0.0)) continue;
                if (!sr_fits(1 << l, J, csize, grid_budget)) continue;
                double cost = 0.0, cost_est = 0.0, cost_gather = 0.0;
                for (double kxh : kreps) {
                    const double lam = sr_cut(st, dt, kxh, J, M.tol, ji < 0 ? 1.0 : xj[ji], 0, SR_RHO);
                    const double cb2 = kxh * kxh * st.vb2, pm2 = kxh * kxh * lam * lam;
                    // (alive: above every velocity so far AND the piece's first -- what fails there is dead on arrival, at no cost)
                    const double w_alive = kxh * std::max(vmax_prev, std::fabs(v[a])), w_cut = std::sqrt(cb2 + pm2);
                    const double i_alive = std::min(std::max(w_alive / dw, 0.0), (double)nf), i_cut = std::min(std::max(w_cut / dw, i_alive), (double)nf);
                    const double n_dir = i_cut - i_alive + 1.0, n_reg = (double)nf - i_cut;
                    const double rest = M.fix + nf * M.cls + n_reg * (M.set + M.W * (M.win + M.perj * J)) + L * (M.out + 0.5 * J * J) +
                                        (double)J * G * (l + 1) * M.fft;
                    cost += rest + n_dir * L * M.dir;
                    cost_est += rest + n_dir * L * M.dir_measured;
                    cost_gather += n_reg * M.W * (M.win + M.perj * J);
                }
                const double per_step = cost / L;
                if (per_step < best) {
                    best = per_step;
                    best_est = cost_est;
                    best_gather = cost_gather;
                    bestL = L;
                    bestJ = J;
                }
            }
        }
        if (!bestL) return false;                      // (nothing fits: not ours)
        hp.model_cost += 0.5 * best_est;
        hp.model_gather += 0.5 * best_gather;
        SrPiece pc{};
        pc.start = a;
        pc.len = bestL;
        pc.J = bestJ;
        int l = 4;
        while ((1 << l) < bestL) ++l;
        pc.loglp = l;
        pc.mj = bestJ / 2;
        pc.ev_off = (int)hp.ev.size();
        SrStats st;
        std::vector<double> table;
        sr_stats(v, a, bestL, &st, &table, pc.mj);
        hp.ev.insert(hp.ev.end(), table.begin(), table.end());
        pc.vb2 = st.vb2;
        pc.s = st.s;
        // terms of the phase series: b_m E~_m(L-1) rho^m dt psi falls below 1e-14 (float64 data) / 1e-9 (float32) radians at rho = SR_RHO
        pc.mser = 0;
        if (st.s > 0.0) {
            const double want = dbl ? 1e-14 : 1e-9;                            // dt psi <= pi, |E~_m| <= L
            double rp = 1.0;
            for (int m = 1; m <= SR_MSER; ++m) {
                rp *= SR_RHO;
                pc.be[m - 1] = sr_b(m) * st.aend[m - 1];
                const double bound = m <= SR_MMAJ ? st.amax[m - 1] : (double)bestL;
                if (3.2 * sr_b(m) * bound * rp > want || m <= 2) pc.mser = m;
            }
        }
        for (int j = 0; j < SR_NKX; ++j) {
            const double lam = sr_cut(st, dt, (double)(j + 1) * kxh_max / SR_NKX, bestJ, M.tol, sr_xj(bestJ, M.tol), 4, SR_RHO);
            pc.lam[j] = (float)(lam * (1.0 + 1e-6));
        }
        hp.grid_bytes = std::max(hp.grid_bytes, (int)((size_t)bestJ * (sr_pad(2 << l) + 1) * csize));
        hp.pieces.push_back(pc);
        for (int t = a; t < a + bestL; ++t) {
            vmax_prev = std::max(vmax_prev, std::fabs(v[t]));
            for (double kxh : kreps) hp.alive_pairs += 0.5 * std::max((double)nf - kxh * vmax_prev / dw, 0.0);
        }
        a += bestL;
    }
    return true;
}

#ifdef SR_PLAN_PROBE
// the plan as flat arrays (tests/test_series_scheme.py compiles this header by itself and holds the scheme -- pieces, cuts, series --
// to a float64 direct sum in NumPy)
extern "C" int impdar_sr_plan_probe(const double *v, int snum, double dt, double dw, int nf, double kxh_max, int dbl, int max_pieces,
                                    int *ints /* [max_pieces][6]: start, len, loglp, J, mser, mj */,
                                    double *dbls /* [max_pieces][2 + SR_MSER + SR_NKX]: vb2, s, be, lam */, double *ev, int ev_cap, int *ev_offs)
{
    SrHostPlan hp;
    if (!sr_make_plan(hp, v, snum, dt, dw, nf, kxh_max, dbl != 0)) return -1;
    if ((int)hp.pieces.size() > max_pieces || (int)hp.ev.size() > ev_cap) return -2;
    for (size_t i = 0; i < hp.pieces.size(); ++i) {
        const SrPiece &p = hp.pieces[i];
        int *I = ints + 6 * i;
        I[0] = p.start; I[1] = p.len; I[2] = p.loglp; I[3] = p.J; I[4] = p.mser; I[5] = p.mj;
        double *D = dbls + (2 + SR_MSER + SR_NKX) * i;
        D[0] = p.vb2; D[1] = p.s;
        for (int m = 0; m < SR_MSER; ++m) D[2 + m] = p.be[m];
        for (int j = 0; j < SR_NKX; ++j) D[2 + SR_MSER + j] = p.lam[j];
        ev_offs[i] = p.ev_off;
    }
    for (size_t i = 0; i < hp.ev.size(); ++i) ev[i] = hp.ev[i];
    dbls[1] = dbls[1];
    if (max_pieces > 0 && ev_cap > 0) ev[hp.ev.size() < (size_t)ev_cap ? hp.ev.size() : (size_t)ev_cap - 1] = hp.model_cost;   // (one past the tables)
    return (int)hp.pieces.size();
}
#endif
