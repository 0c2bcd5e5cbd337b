// Kirchhoff diffraction-summation migration for gfx950 (MI355X).
//
// Behaviour restated from the reference (paths relative to the ImpDAR tree):
//   src/impdar/lib/migrationlib/mig_python.py:35-60   migrationKirchhoffLoop
//   src/impdar/lib/migrationlib/mig_python.py:63-123  migrationKirchhoff
//   src/impdar/lib/migrationlib/mig_cython.h:11       mig_kirch_loop (native hook)
//
// Kernels:
//   kirch_prep_kernel    time gradient (numpy.gradient semantics, :93) fused with the
//                        (snum,tnum) -> trace-major (tnum,snum) transpose, or into groups of 8
//                        traces that are sample-major inside (the quad kernel's image).
//   kirch_exact_kernel   per-pair fp64 index math in the reference's operation
//                        order; any dist / travel_time; float or double data.
//   kirch_tablex_kernel / kirch_exact_tab_kernel
//                        the same fp64 picks and weights tabulated per (sample, |offset|) on uniform
//                        grids: one gather + one fp64 FMA per pair (the float64 parity path).
//   kirch_tableq_kernel  fp64 pick per (sample, trace offset) in the reference's operation order
//                        (uniform grids), stored by step block as the LDS offset of the picked row.
//   kirch_quad_kernel    the MI355X hot path: fp32 data, uniform grids.  A workgroup owns 256
//                        output samples x 24 output traces; input traces stream by LDS-DMA through
//                        a 40-trace LDS ring read with ds_read_b128; picks come from the table.
//   kirch_table_kernel / kirch_tab_kernel
//                        same idea with a trace-major ring, ds_read_b32 and an fp32 weight table,
//                        for geometries whose moveout does not fit the quad ring.
#include "common.h"
#include "kirch_plan.h"
#include <limits>
#include <cmath>
#include <algorithm>
#include <chrono>
#include <thread>


// ===========================================================================
// prep: gradient + transpose
// ===========================================================================
struct PrepParams {
    const void *data;       // (snum, ld) row-major, column block [0,nloc)
    int ld;
    int snum, nloc, jlo;    // image rows written: jlo .. jlo+nloc-1
    void *GT;               // (tnum_pad, snum)
    void *DT;               // same or null
    int grad_uniform;       // see impdar_hip.h
    int precomputed;        // data already holds the gradient (mig_kirch_loop)
    double grad_h;
    const double *ga, *gb, *gc;
    int clean;              // 1: map non-finite values to 0 (fp32 ring kernels); 2: NaN only (fp64 ring kernel)
    int i8;                 // image layout: 0 trace-major [row][k]; 1 groups of 8 rows (4 for float64: 32 bytes per
                            // sample either way), sample-major inside a group: element (row, k) at
                            // ((row / G) * snum + k) * G + (row % G)
};

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void kirch_prep_kernel(PrepParams P)
{
    // tiles in the image's element type: 2 x 18 KB for float32.  Row stride 65 for the trace-major
    // store (lanes walk a tile column), 72 for the grouped store (a half wave reads 4 rows x 8 columns)
    __shared__ TO tg[64 * 72];
    __shared__ TO td[64 * 72];
    const int ld = P.i8 ? 72 : 65;
    const TI *f = reinterpret_cast<const TI *>(P.data);
    const int k0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n = P.snum;
    for (int r = ty; r < 64; r += 4) {
        const int k = k0 + r, j = j0 + tx;
        double g = 0.0, d = 0.0;
        if (k < n && j < P.nloc) {
            const size_t c = (size_t)j;
            d = (double)f[(size_t)k * P.ld + c];
            if (P.precomputed) {
                g = d;
            } else if (k == 0) {
                const double h = P.grad_uniform ? P.grad_h : P.ga[0];
                g = ((double)f[(size_t)1 * P.ld + c] - d) / h;
            } else if (k == n - 1) {
                const double h = P.grad_uniform ? P.grad_h : P.ga[n - 1];
                g = (d - (double)f[(size_t)(k - 1) * P.ld + c]) / h;
            } else {
                const double fm = (double)f[(size_t)(k - 1) * P.ld + c];
                const double fp = (double)f[(size_t)(k + 1) * P.ld + c];
                if (P.grad_uniform) {
                    g = (fp - fm) / (2.0 * P.grad_h);
                } else {
                    g = (P.ga[k] * fm + P.gb[k] * d) + P.gc[k] * fp;
                }
            }
            if (sizeof(TI) == 4 && !P.precomputed) g = (double)(float)g;  // numpy keeps f32 gradients in f32
            if (P.clean == 1) {
                if (!(fabs(g) <= 1.79e308)) g = 0.0;
                if (!(fabs(d) <= 1.79e308)) d = 0.0;
            }
        }
        tg[r * ld + tx] = (TO)g;
        td[r * ld + tx] = (TO)d;
    }
    __syncthreads();
    TO *GT = reinterpret_cast<TO *>(P.GT);
    TO *DT = reinterpret_cast<TO *>(P.DT);
    if (P.i8) {
        // a wave stores 8 samples x 8 rows = 256 contiguous bytes per instruction
        const int kk = tx >> 3, jj = tx & 7;
        for (int g8 = 2 * ty; g8 < 2 * ty + 2; ++g8)
            for (int kb = 0; kb < 8; ++kb) {
                const int r = kb * 8 + kk, c = g8 * 8 + jj;
                const int j = j0 + c, k = k0 + r;
                if (j < P.nloc && k < n) {
                    const int row = P.jlo + j;
                    const size_t o = ((size_t)(row >> 3) * n + k) * 8 + (row & 7);
                    GT[o] = tg[r * ld + c];
                    if (DT) DT[o] = td[r * ld + c];
                }
            }
        return;
    }
    for (int r = ty; r < 64; r += 4) {
        const int j = j0 + r, k = k0 + tx;
        if (j < P.nloc && k < n) {
            const size_t o = (size_t)(P.jlo + j) * n + k;
            GT[o] = tg[tx * ld + r];
            if (DT) DT[o] = td[tx * ld + r];
        }
    }
}

// LDS-free variant for the grouped layout, float32 (the quad kernel's image).
// The diffraction sum of the previous radargram keeps all of a CU's LDS, so a producer kernel that
// needs LDS cannot start before that kernel drains (and then takes LDS from the next one); this one
// runs on the CUs' spare wave slots underneath it: 8.58 -> 8.30 ms per step at config 3, and the
// difference between a working and a counter-productive overlap for the short kernels of an 8-rank run.
// One thread per (sample k, trace j): reads are coalesced along j, writes are 32-byte runs per group.
template <typename T, int GRP>
__global__ __launch_bounds__(256) void kirch_prep_direct_kernel(PrepParams P)
{
    const T *f = reinterpret_cast<const T *>(P.data);
    const int j = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
    const int n = P.snum;
    if (j >= P.nloc) return;
    const size_t c = (size_t)j;
    double d = (double)f[(size_t)k * P.ld + c], g;
    if (P.precomputed) {
        g = d;
    } else if (k == 0) {
        const double h = P.grad_uniform ? P.grad_h : P.ga[0];
        g = ((double)f[(size_t)1 * P.ld + c] - d) / h;
    } else if (k == n - 1) {
        const double h = P.grad_uniform ? P.grad_h : P.ga[n - 1];
        g = (d - (double)f[(size_t)(k - 1) * P.ld + c]) / h;
    } else {
        const double fm = (double)f[(size_t)(k - 1) * P.ld + c];
        const double fp = (double)f[(size_t)(k + 1) * P.ld + c];
        if (P.grad_uniform)
            g = (fp - fm) / (2.0 * P.grad_h);
        else
            g = (P.ga[k] * fm + P.gb[k] * d) + P.gc[k] * fp;
    }
    if (sizeof(T) == 4 && !P.precomputed) g = (double)(float)g;                  // numpy keeps f32 gradients in f32
    if (P.clean == 1) {                                        // fp32 ring kernels: every non-finite value
        if (!(fabs(g) <= 1.79e308)) g = 0.0;
        if (!(fabs(d) <= 1.79e308)) d = 0.0;
    } else if (P.clean == 2) {                                 // fp64 ring kernel: NaN terms are what nansum skips
        if (g != g) g = 0.0;                                   // (mig_python.py:53); infinities stay and propagate
        if (d != d) d = 0.0;
    }
    const int row = P.jlo + j;
    const size_t o = ((size_t)(row / GRP) * n + k) * GRP + (row % GRP);
    reinterpret_cast<T *>(P.GT)[o] = (T)g;
    if (P.DT) reinterpret_cast<T *>(P.DT)[o] = (T)d;
}

// ===========================================================================
// exact kernel: reference operation order, fp64 index math, any geometry
// ===========================================================================
struct ExactParams {
    const void *GT, *DT;
    void *out;
    int ldo;
    int snum, tnum, xlo, xhi;
    const double *dist, *zs, *zs2, *tt;
    double vel, tmax, r2lim, inv_dt, tt0;
    int dist_sorted;     // dist is non-decreasing: the traces inside the aperture are one index range
};

template <typename T, bool NEAR>
__global__ __launch_bounds__(256) void kirch_exact_kernel(ExactParams P)
{
    const int ti = blockIdx.x * 256 + threadIdx.x;
    const int xi = P.xlo + blockIdx.y;
    if (ti >= P.snum || xi >= P.xhi) return;
    const T *GT = reinterpret_cast<const T *>(P.GT);
    const T *DT = reinterpret_cast<const T *>(P.DT);
    const double dxi = P.dist[xi];
    const double z = P.zs[ti], z2 = P.zs2[ti];
    const int n = P.snum;
    double far = 0.0, near = 0.0;
    int jlo = 0, jhi = P.tnum;
    if (P.dist_sorted) {
        // traces with (dist - dist[xi])^2 + z^2 <= r2lim form one index range: bisect its ends (a guard
        // trace on both sides; the per-pair test below still decides)
        const double rad2 = P.r2lim - z2;
        if (rad2 < 0.0) {
            jhi = 0;
        } else {
            const double rad = sqrt(rad2) * (1.0 + 1e-12);
            int a = 0, b = xi;                       // first j with dist[j] >= dxi - rad
            while (a < b) {
                const int m = (a + b) >> 1;
                if (P.dist[m] < dxi - rad) a = m + 1; else b = m;
            }
            jlo = max(a - 1, 0);
            a = xi;
            b = P.tnum;                              // first j with dist[j] > dxi + rad
            while (a < b) {
                const int m = (a + b) >> 1;
                if (P.dist[m] <= dxi + rad) a = m + 1; else b = m;
            }
            jhi = min(a + 1, P.tnum);
        }
    }
    for (int j = jlo; j < jhi; ++j) {
        const double dx = P.dist[j] - dxi;
        const double q = dx * dx + z2;                 // :44 (compiled with -ffp-contract=off)
        if (q > P.r2lim) continue;                     // far outside the aperture
        const double rs = sqrt(q);
        const double cost = z / rs;                    // :47
        const double t = 2.0 * rs / P.vel;             // :49
        if (t > P.tmax) continue;                      // :52 -> term is 0 (or NaN, skipped)
        // nearest sample, ties to the lower index (:49 argmin)
        int k0 = (int)floor((t - P.tt0) * P.inv_dt);
        k0 = min(max(k0, 0), n - 1);
        while (k0 < n - 1 && P.tt[k0 + 1] <= t) ++k0;
        while (k0 > 0 && P.tt[k0] > t) --k0;
        const int k1 = min(k0 + 1, n - 1);
        const int k = (fabs(P.tt[k1] - t) < fabs(P.tt[k0] - t)) ? k1 : k0;
        const size_t o = (size_t)j * n + k;
        const double term = (double)GT[o] * cost / P.vel;   // :53
        if (term == term) far += term;                       // nansum
        if (NEAR) {
            const double term2 = (double)DT[o] * cost / (rs * rs);   // :58
            if (term2 == term2) near += term2;
        }
    }
    const double c = 1.0 / (2.0 * 3.141592653589793);
    T *out = reinterpret_cast<T *>(P.out);
    out[(size_t)ti * P.ldo + (xi - P.xlo)] = (T)(c * (far + near));   // :60
}

// ---------------------------------------------------------------------------
// exact path on uniform grids: the pick and the obliquity factor of a pair depend on (sample, |trace
// offset|) only, so they are tabulated once per geometry -- in fp64, in the reference's operation order
// (mig_python.py:44-58) -- and the diffraction sum becomes one gather + one fp64 FMA per pair.  Same
// picks as kirch_exact_kernel; the weight cos/vel is rounded once more than the reference's
// (g*cos)/vel (1 ulp, far inside the 1e-12 bar).
// ---------------------------------------------------------------------------
struct TableXParams {
    int *XK;                 // [ntab][snum] picked sample (0 where the pair is dropped)
    double *XW, *XW2;        // cos/vel and cos/rs^2 (0 where dropped: t > t_max or the 0/0 apex)
    const double *zs, *zs2, *tt;
    double dx, vel, tmax, inv_dt, tt0;
    int snum, ntab, near;
};

__global__ __launch_bounds__(256) void kirch_tablex_kernel(TableXParams P)
{
    const int ti = blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y;
    if (ti >= P.snum) return;
    const size_t o = (size_t)n * P.snum + ti;
    int kk = 0;
    double w = 0.0, w2 = 0.0;
    const double dx = (double)n * P.dx;
    const double q = dx * dx + P.zs2[ti];
    const double rs = sqrt(q);
    const double cost = P.zs[ti] / rs;
    const double t = 2.0 * rs / P.vel;
    if (!(t > P.tmax) && cost == cost) {
        const int ns = P.snum;
        int k0 = (int)floor((t - P.tt0) * P.inv_dt);
        k0 = min(max(k0, 0), ns - 1);
        while (k0 < ns - 1 && P.tt[k0 + 1] <= t) ++k0;
        while (k0 > 0 && P.tt[k0] > t) --k0;
        const int k1 = min(k0 + 1, ns - 1);
        kk = (fabs(P.tt[k1] - t) < fabs(P.tt[k0] - t)) ? k1 : k0;
        w = cost / P.vel;
        w2 = cost / (rs * rs);
    }
    P.XK[o] = kk;
    P.XW[o] = w;
    if (P.near) P.XW2[o] = w2;
}

// Every table above assumes that the pick of a pair depends on (sample, |trace offset|) only.  That fails exactly
// where the reference's own answer is decided by rounding noise: a travel time that falls (to float64 rounding) on
// the midpoint between two samples, or on t_max itself.  The reference evaluates t from dist[j] - dist[xi], whose
// last bits differ from pair to pair with the same offset, so such ties break either way inside ONE offset
// (geometries with a rational moveout 2dx/(v dt), e.g. 2.5 samples per trace, are full of them: 4a^2 + 25n^2 is
// an odd square for whole families of (a, n)).  This scan flags a geometry that has any such entry inside an
// aperture; the plan then keeps the per-pair kernel, which repeats the reference's arithmetic pair by pair.
// Margin: the relative rounding noise of t, ~1e-15, plus the noise of dist[j] - dist[xi] against n * dx, times
// |t|/dt, in samples.  `xnoise` is that position noise in units of dx, MEASURED on the profile by the host: twice the
// largest deviation of dist[] from the fitted grid (the plan accepts up to 1e-9 dx as "uniform") plus the rounding of
// the largest |dist| (a profile that starts 50 km along a line has ulp(dist)/dx ~ 7e-12, far above what a profile
// starting at 0 has), never less than 4.5e-16 * tnum.
__global__ __launch_bounds__(256) void kirch_tiescan_kernel(TableXParams P, double xnoise, int *count, int2 *list, int cap)
{
    const int ti = blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y;
    if (ti >= P.snum) return;
    const double dx = (double)n * P.dx;
    const double q = dx * dx + P.zs2[ti];
    const double rs = sqrt(q);
    const double cost = P.zs[ti] / rs;
    const double t = 2.0 * rs / P.vel;
    if (!(cost == cost)) return;                       // 0/0 apex: dropped whatever the noise
    const double u = (t - P.tt0) * P.inv_dt, um = (P.tmax - P.tt0) * P.inv_dt;
    // ten times the noise estimate: a wider net only makes the list (a handful of entries) longer
    const double eps = 10.0 * (fabs(t * P.inv_dt) + 1.0) * (1.0e-15 + xnoise / (double)max(n, 1));
    if (u > um + eps) return;                          // clearly outside the aperture
    bool amb = fabs(u - um) <= eps && n > 0;           // the t > t_max test itself (n = 0: t = tt exactly)
    if (u >= 0.0 && u <= um + eps && n > 0) {
        const double fr = u - floor(u);
        amb = amb || fabs(fr - 0.5) <= eps;
    }
    if (amb) {
        const int at = atomicAdd(count, 1);
        if (at < cap) list[at] = make_int2(ti, n);
    }
}

// ---------------------------------------------------------------------------
// ... and the table-driven kernels keep running on such geometries: the flagged (sample, offset) entries (one in
// 1.1e7 for dx 1 m, dt 10 ns, v 1.68e8 m/s -- the velocity RadarData.migrate defaults to -- a few thousand for a
// moveout of exactly 2.5) are re-done pair by pair in the reference's arithmetic after the diffraction sum, and
// where a pair's own answer (pick, or the t > t_max drop) differs from the table's, the output gets the
// difference.  One thread per (sample with flagged offsets, output trace), its offsets and the two sides in a
// fixed order: deterministic, no atomics.  Per-pair parity on every geometry at the cost of a few microseconds.
// ---------------------------------------------------------------------------
struct TieFixParams {
    const void *GT, *DT;     // images (layout: grp = 0 trace-major [row][k]; else groups of `grp` rows, sample-major inside)
    void *out;
    int ldo, snum, tnum, xlo, xhi, grp, near;
    const double *dist, *zs, *zs2, *tt;
    double dx, vel, tmax, inv_dt, tt0;
    const int *g_ti, *g_off, *g_n;   // groups: sample, [first, last) into g_n (offsets, ascending)
    const int *hmax;                 // per 256-sample chunk: offsets beyond are not walked
    int nmax;                        // offsets >= nmax are dropped by the tables whatever t is
};

template <typename T>
__global__ __launch_bounds__(256) void kirch_tiefix_kernel(TieFixParams P)
{
    const int g = blockIdx.y;
    const int xi = P.xlo + blockIdx.x * 256 + threadIdx.x;
    if (xi >= P.xhi) return;
    const int ti = P.g_ti[g];
    const T *GT = reinterpret_cast<const T *>(P.GT);
    const T *DT = reinterpret_cast<const T *>(P.DT);
    const int ns = P.snum;
    auto at = [&](const T *img, int j, int k) -> double {
        const size_t o = P.grp ? ((size_t)(j / P.grp) * ns + k) * P.grp + (j % P.grp) : (size_t)j * ns + k;
        return (double)img[o];
    };
    auto pick = [&](double t) {
        int k0 = (int)floor((t - P.tt0) * P.inv_dt);
        k0 = min(max(k0, 0), ns - 1);
        while (k0 < ns - 1 && P.tt[k0 + 1] <= t) ++k0;
        while (k0 > 0 && P.tt[k0] > t) --k0;
        const int k1 = min(k0 + 1, ns - 1);
        return (fabs(P.tt[k1] - t) < fabs(P.tt[k0] - t)) ? k1 : k0;      // :49, ties to the lower index
    };
    const double z = P.zs[ti], z2 = P.zs2[ti];
    const double c2pi = 1.0 / (2.0 * 3.141592653589793);
    const int hm = P.hmax[ti >> 8];
    double far = 0.0, nearsum = 0.0;
    for (int e = P.g_off[g]; e < P.g_off[g + 1]; ++e) {
        const int n = P.g_n[e];
        // what the table holds for (ti, n): the pick at dx = n * spacing (kirch_tableq / tabled / tablex kernels)
        const double dxn = (double)n * P.dx;
        const double rsn = sqrt(dxn * dxn + z2);
        const double costn = z / rsn;
        const double tn = 2.0 * rsn / P.vel;
        const bool drop_tab = n >= P.nmax || n > hm || tn > P.tmax || !(costn == costn);
        const int k_tab = drop_tab ? 0 : pick(tn);
        for (int sgn = -1; sgn <= 1; sgn += 2) {
            const int j = xi + sgn * n;
            if (j < 0 || j >= P.tnum) continue;
            const double dxj = P.dist[j] - P.dist[xi];
            const double q = dxj * dxj + z2;                   // :44
            const double rs = sqrt(q);
            const double cost = z / rs;                        // :47
            const double t = 2.0 * rs / P.vel;                 // :49
            const bool drop = t > P.tmax || !(cost == cost);   // :52, nansum
            const int k = drop ? 0 : pick(t);
            if (drop == drop_tab && (drop || k == k_tab)) continue;
            // the weight the diffraction sum gave this offset (a function of (ti, n) in every table-driven kernel)
            const double wf = costn / P.vel, wn = costn / (rsn * rsn);
            const double gt = drop ? 0.0 : at(GT, j, k), gb = drop_tab ? 0.0 : at(GT, j, k_tab);
            far += (gt - gb) * wf;
            if (P.near) {
                const double dt_ = drop ? 0.0 : at(DT, j, k), db = drop_tab ? 0.0 : at(DT, j, k_tab);
                nearsum += (dt_ - db) * wn;
            }
        }
    }
    const double delta = c2pi * (far + nearsum);
    if (delta != 0.0) {
        T *o = reinterpret_cast<T *>(P.out) + (size_t)ti * P.ldo + (xi - P.xlo);
        *o = (T)((double)*o + delta);
    }
}

struct ExactTabParams {
    const void *GT, *DT;
    void *out;
    int ldo, snum, tnum, xlo, xhi;
    const int *XK;
    const double *XW, *XW2;
    const int *hmax;         // per 256-sample chunk: largest aperture half width of its samples
    int ntab;
};

template <typename T, bool NEAR, int XE>
__global__ __launch_bounds__(256) void kirch_exact_tab_kernel(ExactTabParams P)
{
    const int ti_raw = blockIdx.x * 256 + threadIdx.x;
    const int ti = min(ti_raw, P.snum - 1);
    const int x0 = P.xlo + blockIdx.y * XE;
    const T *GT = reinterpret_cast<const T *>(P.GT);
    const T *DT = reinterpret_cast<const T *>(P.DT);
    const int hm = min(P.hmax[blockIdx.x], P.ntab - 1);
    const int nlo = max(-hm, -(x0 + XE - 1)), nhi = min(hm, P.tnum - 1 - x0);
    double far[XE], near[XE];
#pragma unroll
    for (int e = 0; e < XE; ++e) far[e] = near[e] = 0.0;
    const size_t sn = (size_t)P.snum;
    for (int n = nlo; n <= nhi; ++n) {
        const size_t o = (size_t)abs(n) * sn + ti;
        const int k = P.XK[o];
        const double w = P.XW[o];
        const double w2 = NEAR ? P.XW2[o] : 0.0;
#pragma unroll
        for (int e = 0; e < XE; ++e) {
            const int j = x0 + e + n;
            if (j >= 0 && j < P.tnum) {
                const size_t g = (size_t)j * sn + k;
                const double term = (double)GT[g] * w;            // :53
                if (term == term) far[e] += term;                 // nansum
                if (NEAR) {
                    const double term2 = (double)DT[g] * w2;      // :58
                    if (term2 == term2) near[e] += term2;
                }
            }
        }
    }
    if (ti_raw < P.snum) {
        const double c = 1.0 / (2.0 * 3.141592653589793);
        T *out = reinterpret_cast<T *>(P.out);
#pragma unroll
        for (int e = 0; e < XE; ++e)
            if (x0 + e < P.xhi) out[(size_t)ti_raw * P.ldo + (x0 + e - P.xlo)] = (T)(c * (far[e] + near[e]));   // :60
    }
}

// ===========================================================================
// fast kernel
// ===========================================================================
struct FastParams {
    const float *GT, *DT;
    float *out;
    int ldo;
    int snum, tnum, xlo, xhi;
    const int *hmax;               // per sample-chunk aperture half width (+1 guard)     [nchunks]
    int nchunks, nxt, tiles_per_xcd, G;
    int parts_log2;                // persistent ring kernels: every tile's walk as 1 << parts_log2 queue items (pieces)
    void *partial;                 // ... their partial images [piece][snum][ldo] (element type = image type)
    size_t part_stride;            // ... elements per partial image
    int *queue;                    // quad kernel, persistent workgroups: 8 item counters (one per XCD, 64 bytes apart), zeroed
                                   // before the launch; null = one item per block
    const short *tilemap;          // ring kernels: [nchunks][tiles_per_xcd][8] output tile of (chunk, slot, XCD), -1 = none
                                   // (host-balanced over the XCDs by step count, build_tilemap); null: arithmetic rule
    // tab kernel only:
    const int *klo, *khi;          // per (chunk, |n|) first / last sample a trace at offset n is asked for
    int nb;                        // entries per chunk in klo/khi
    int zero_row;                  // index of an all-zero image row (out-of-profile traces)
    const unsigned short *TK;      // pick table (|n|, ti): offset of the picked sample inside a ring slot [ntab][snum]
    const float *TW, *TW2;         // far / near weights (0 where the reference drops the pair)
    // quad kernel: cos(theta) = sign(a) * rsq(1 + c1 n^2) with a = tt/dt, c1 = alpha / a^2; the far-field sum is
    // scaled by fin = sign(a) / (2 pi v) once at the end, the near-field weight is c2 * cos^3 in those units
    const float *c1, *c2, *fin;    // per sample [snum]
    const double *c1d, *c2d, *find;   // the same in float64 (dquad kernel): c1 = (dx/zs)^2, fin = 1/(2 pi v), c2: see kirch_dquad_kernel
    // quad kernel, step-block tables.  Every workgroup walks the trace offset n in blocks of 8 steps
    // n = 8 m + 1 .. 8 m + 8 (tiles start at multiples of 8, so the alignment is the same for all);
    // table row = m + mrow0:
    const void *TKB;               // [nrows][snum] x 8 picks of 16 bits: one 16-byte load per lane per block
    const int2 *WIN;               // [nchunks][nrows] staging window of the 8 traces a block adds:
                                   //   x = kmin | (kmin mod W) << 16, y = kmax
    int nrows, mrow0;
    int ntab;                      // tab kernel: table rows; the last row is all zero (|n| beyond every aperture)
};

// ---------------------------------------------------------------------------
// pick / weight table.  For a uniform trace spacing the sample picked by the
// pair (output sample ti, input trace xi+n) and its obliquity weight depend on
// (ti, |n|) only.  One thread per entry evaluates them in fp64 in the
// reference's operation order (mig_python.py:44-58), so the diffraction-sum
// kernel below inherits the reference's picks, its t>tmax test and its NaN
// skipping exactly; the table is rebuilt by every prep (it is cheap: ~1e7
// entries) so nothing about it is cached across migrations.
// ---------------------------------------------------------------------------
struct TableParams {
    unsigned short *TK;
    float *TW, *TW2;
    const double *zs, *zs2, *tt;
    double dx, vel, tmax, inv_dt, tt0;
    int snum, ntab, near;
    int wmod, kscale;          // slot byte offset of sample k = (k % wmod) * kscale
    unsigned short sentinel;   // TK value for dropped pairs (0xFFFF for the quad kernel, 0 for tab: its weight is 0)
    int write_w;
};

__global__ __launch_bounds__(256) void kirch_table_kernel(TableParams P)
{
    const int ti = blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y;
    if (ti >= P.snum) return;
    const size_t o = (size_t)n * P.snum + ti;
    unsigned short kb = P.sentinel;      // pair dropped by the reference
    float w = 0.f, w2 = 0.f;
    if (n < P.ntab - 1) {
        const double dx = (double)n * P.dx;
        const double q = dx * dx + P.zs2[ti];
        const double rs = sqrt(q);
        const double cost = P.zs[ti] / rs;
        const double t = 2.0 * rs / P.vel;
        if (!(t > P.tmax) && cost == cost) {
            const int ns = P.snum;
            int k0 = (int)floor((t - P.tt0) * P.inv_dt);
            k0 = min(max(k0, 0), ns - 1);
            while (k0 < ns - 1 && P.tt[k0 + 1] <= t) ++k0;
            while (k0 > 0 && P.tt[k0] > t) --k0;
            const int k1 = min(k0 + 1, ns - 1);
            const int k = (fabs(P.tt[k1] - t) < fabs(P.tt[k0] - t)) ? k1 : k0;
            kb = (unsigned short)((k % P.wmod) * P.kscale);
            const double c2pi = 1.0 / (2.0 * 3.141592653589793);
            w = (float)(c2pi * (cost / P.vel));
            if (P.near) w2 = (float)(c2pi * (cost / (rs * rs)));
        }
    }
    P.TK[o] = kb;
    if (P.write_w) {
        P.TW[o] = w;
        if (P.near) P.TW2[o] = w2;
    }
}

// Same picks for the quad kernel, laid out by step block: row r holds, per sample, the 8 picks of the
// offsets n = 8 (r - mrow0) + 1 + s, s = 0..7 (|n| decides the pick; blocks left of the apex are stored
// mirrored, so the kernel never computes |n|).  One thread per (row, sample), one 16-byte store.
struct TableQParams {
    uint4 *TKB;
    const double *zs, *zs2, *tt;
    double dx, vel, tmax, inv_dt, tt0;
    int snum, nrows, mrow0, nmax;   // offsets |n| >= nmax are outside every aperture
    int wmod, sh;                   // ring rows; entries are LDS byte offsets >> sh
    int ps;                         // bytes per 32-row piece of the LDS image (kq_piece_bytes)
};

// Row stride of the step-block pick table in 16-byte entries.  Not snum: with snum a power of two a chunk's
// slice of consecutive rows (256 entries out of every snum) lands on 1/16 of the L2's sets (and of whatever else
// indexes by address bits); the odd number of 256-byte lines of padding walks the slice over all of them.
#ifndef KQ_TKB_PAD
#define KQ_TKB_PAD 272       // entries (4352 bytes)
#endif
__host__ __device__ static inline size_t kq_tkb_stride(int snum) { return (size_t)snum + KQ_TKB_PAD; }

// byte offset of half 0 of ring row r in the quad kernel's LDS image (layout: see kirch_quad_kernel)
__host__ __device__ static inline unsigned kq_row_offset(int r, int ps)
{
    return (unsigned)(r >> 5) * (unsigned)ps + (unsigned)(r & 31) * 32u + (((unsigned)(r >> 3) & 1u) << 4);
}

__global__ __launch_bounds__(256) void kirch_tableq_kernel(TableQParams P)
{
    // One thread evaluates the 9 picks |n| = 8a .. 8a+8 of one sample and fills TWO rows with them: the
    // block right of the apex n = 8a+1 .. 8a+8 (row a + mrow0) and its mirror n = -8a-7 .. -8a
    // (row mrow0 - a - 1, picks in reverse order): 9 evaluations instead of 16.  A pick beyond the
    // sample's own aperture (t > t_max) is the all-zero row; so is everything once t exceeded t_max
    // (t grows with |n|), which skips the fp64 work for the outer part of the table.
    const int ti = blockIdx.x * 256 + threadIdx.x;
    const int a = blockIdx.y;
    if (ti >= P.snum) return;
    const unsigned zero_row = 1024u >> P.sh;         // KQ_ZERO
    unsigned pk[9];
    bool out = false;                                // t already beyond t_max at a smaller offset
    const double zs = P.zs[ti], zs2 = P.zs2[ti];
#pragma unroll
    for (int s = 0; s < 9; ++s) {
        const int n = 8 * a + s;
        unsigned kb = zero_row;
        if (n < P.nmax && !out) {
            const double dx = (double)n * P.dx;
            const double q = dx * dx + zs2;
            const double rs = sqrt(q);
            const double cost = zs / rs;
            const double t = 2.0 * rs / P.vel;
            if (t > P.tmax) {
                out = true;
            } else if (cost == cost) {
                const int ns = P.snum;
                int k0 = (int)floor((t - P.tt0) * P.inv_dt);
                k0 = min(max(k0, 0), ns - 1);
                while (k0 < ns - 1 && P.tt[k0 + 1] <= t) ++k0;
                while (k0 > 0 && P.tt[k0] > t) --k0;
                const int k1 = min(k0 + 1, ns - 1);
                const int k = (fabs(P.tt[k1] - t) < fabs(P.tt[k0] - t)) ? k1 : k0;
                kb = kq_row_offset(k % P.wmod, P.ps) >> P.sh;
            }
        }
        pk[s] = kb;
    }
    const int rp = a + P.mrow0, rn = P.mrow0 - a - 1;
    if (rp < P.nrows)
        P.TKB[(size_t)rp * kq_tkb_stride(P.snum) + ti] =
            make_uint4(pk[1] | (pk[2] << 16), pk[3] | (pk[4] << 16), pk[5] | (pk[6] << 16), pk[7] | (pk[8] << 16));
    if (rn >= 0)
        P.TKB[(size_t)rn * kq_tkb_stride(P.snum) + ti] =
            make_uint4(pk[7] | (pk[6] << 16), pk[5] | (pk[4] << 16), pk[3] | (pk[2] << 16), pk[1] | (pk[0] << 16));
}

// ---------------------------------------------------------------------------
// table-driven ring kernel, trace-major LDS layout: a 256-sample x XB-trace tile walks the
// trace offset n; the XB+8 input traces in use sit in an LDS ring of 512-sample circular
// slots, the per-(sample, offset) pick and weight come from the table (two coalesced loads
// per lane per step, prefetched a block ahead), one conflict-free ds_read_b32 + FMA per pair.
// ---------------------------------------------------------------------------
template <int XB, int S, bool NEAR, int OCC>
__global__ __launch_bounds__(KF_THREADS, OCC) void kirch_tab_kernel(FastParams P)
{
    constexpr int R = XB + 2 * S;
    constexpr int W = KF_W;
    static_assert(R % S == 0, "ring must be a whole number of step blocks");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *ldsG = lds;
    float *ldsD = lds + R * W;

    const int b = blockIdx.x;
    const int xcd = b & 7, r = b >> 3;
    const int chunk = r / P.tiles_per_xcd;
    const int qx = r - chunk * P.tiles_per_xcd;
    const int xt = ((qx / P.G) * 8 + xcd) * P.G + (qx % P.G);
    if (chunk >= P.nchunks || xt >= P.nxt) return;

    const int tid = threadIdx.x;
    const int s0 = chunk * KF_THREADS;
    const int x0 = P.xlo + xt * XB;
    const int snum = P.snum, tnum = P.tnum;
    const int ti_raw = s0 + tid;
    const int ti = min(ti_raw, snum - 1);

    const int hmax = P.hmax[chunk];
    const int *klo = P.klo + (size_t)chunk * P.nb;
    const int *khi = P.khi + (size_t)chunk * P.nb;
    const int nlo = max(-hmax, -(x0 + XB - 1));
    const int nhi = min(hmax, tnum - 1 - x0);
    const int nsteps = nhi - nlo + 1;
    const int nblocks = (nsteps + S - 1) / S;
    const int nsteps_pad = nblocks * S;
    const int jbase = x0 + nlo;
    const int ntab1 = P.ntab - 1;

    auto window = [&](int q, int &kmin, int &kmax) {
        const int pmin = max(0, q - (XB - 1)), pmax = min(q, nsteps_pad - 1);
        const int na = nlo + pmin, nb = nlo + pmax;
        const int lo = (na <= 0 && nb >= 0) ? 0 : min(abs(na), abs(nb));
        const int hi = max(abs(na), abs(nb));
        kmin = klo[min(lo, P.nb - 1)];
        kmax = min(khi[min(hi, P.nb - 1)], kmin + W - 1);
    };

    for (int e = tid; e < R * W * (NEAR ? 2 : 1); e += KF_THREADS) lds[e] = 0.f;
    __syncthreads();
    for (int q = 0; q < XB + S - 1; ++q) {
        int kmin, kmax;
        window(q, kmin, kmax);
        const int j = jbase + q;
        const int jr = (j >= 0 && j < tnum) ? j : P.zero_row;
        const int slot = q % R;
        for (int e = kmin + tid; e <= kmax; e += KF_THREADS) {
            ldsG[slot * W + (e & (W - 1))] = P.GT[(size_t)jr * snum + e];
            if (NEAR) ldsD[slot * W + (e & (W - 1))] = P.DT[(size_t)jr * snum + e];
        }
    }
    // table entries of block 0
    unsigned short tkc[S];
    float twc[S], tw2c[NEAR ? S : 1];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const size_t o = (size_t)min(abs(nlo + s), ntab1) * snum + ti;
        tkc[s] = P.TK[o];
        twc[s] = P.TW[o];
        if (NEAR) tw2c[s] = P.TW2[o];
    }
    __syncthreads();

    float acc[XB];
#pragma unroll
    for (int i = 0; i < XB; ++i) acc[i] = 0.f;

    for (int blk0 = 0; blk0 < nblocks; blk0 += R / S) {
#pragma unroll
        for (int bb = 0; bb < R / S; ++bb) {
            const int blk = blk0 + bb;
            if (blk >= nblocks) break;
            float pfG[S][2], pfD[S][2];
            int pk[S];
            unsigned short tkn[S];
            float twn[S], tw2n[NEAR ? S : 1];
            const bool more = (blk + 1 < nblocks);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                pfG[s][0] = pfG[s][1] = 0.f;
                pfD[s][0] = pfD[s][1] = 0.f;
                pk[s] = 0;
                tkn[s] = 0;
                twn[s] = 0.f;
                if (NEAR) tw2n[s] = 0.f;
                if (more) {
                    const int q = (blk + 1) * S + XB - 1 + s;
                    int kmin, kmax;
                    window(q, kmin, kmax);
                    pk[s] = kmin;
                    const int j = jbase + q;
                    const int jr = (j >= 0 && j < tnum) ? j : P.zero_row;
                    const float *src = P.GT + (size_t)jr * snum;
                    const int e0 = min(kmin + tid, snum - 1), e1 = min(kmin + tid + KF_THREADS, snum - 1);
                    pfG[s][0] = src[e0];
                    pfG[s][1] = src[e1];
                    if (NEAR) {
                        const float *srd = P.DT + (size_t)jr * snum;
                        pfD[s][0] = srd[e0];
                        pfD[s][1] = srd[e1];
                    }
                    const size_t o = (size_t)min(abs(nlo + (blk + 1) * S + s), ntab1) * snum + ti;
                    tkn[s] = P.TK[o];
                    twn[s] = P.TW[o];
                    if (NEAR) tw2n[s] = P.TW2[o];
                }
            }
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int pm = bb * S + s;
                const float w = twc[s];
                const float w2 = NEAR ? tw2c[s] : 0.f;
                const char *base = reinterpret_cast<const char *>(ldsG) + tkc[s];
                const char *based = reinterpret_cast<const char *>(ldsD) + tkc[s];
#pragma unroll
                for (int i = 0; i < XB; ++i) {
                    const int slot = (pm + i) % R;
                    acc[i] = fmaf(w, *reinterpret_cast<const float *>(base + slot * W * 4), acc[i]);
                    if (NEAR) acc[i] = fmaf(w2, *reinterpret_cast<const float *>(based + slot * W * 4), acc[i]);
                }
            }
            if (more) {
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const int slot = ((bb + 1) * S + XB - 1 + s) % R;
                    const int e0 = pk[s] + tid, e1 = e0 + KF_THREADS;
                    ldsG[slot * W + (e0 & (W - 1))] = pfG[s][0];
                    ldsG[slot * W + (e1 & (W - 1))] = pfG[s][1];
                    if (NEAR) {
                        ldsD[slot * W + (e0 & (W - 1))] = pfD[s][0];
                        ldsD[slot * W + (e1 & (W - 1))] = pfD[s][1];
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < S; ++s) {
                tkc[s] = tkn[s];
                twc[s] = twn[s];
                if (NEAR) tw2c[s] = tw2n[s];
            }
            __syncthreads();
        }
    }

    if (ti_raw < snum) {
        float *o = P.out + (size_t)ti_raw * P.ldo + (x0 - P.xlo);
#pragma unroll
        for (int i = 0; i < XB; ++i)
            if (x0 + i < P.xhi) o[i] = acc[i];
    }
}

// ---------------------------------------------------------------------------
// quad kernel: table-driven ring of 40 trace slots (5 groups of 8 traces) in LDS.
// At one step every output i of a lane reads the SAME sample k from XB consecutive ring slots:
// 6-7 aligned ds_read_b128 per step instead of 24 ds_read_b32.
//
// LDS image (filled by LDS-DMA, which writes 1 KiB contiguously per wave instruction):
//   piece  = 32 consecutive ring rows (row = sample mod W, W a multiple of 32)
//   byte(row, group g, half h) = (row >> 5) * KQ_PS + g * KQ_GS + (row & 31) * 32 + 16 * (h ^ ((row >> 3) & 1))
// i.e. per (piece, group) a contiguous [32 rows][8 traces] block of 1 KiB plus one spare 32-byte row; KQ_PS is a
// multiple of 256 B, which keeps rows of neighbouring pieces on distinct banks (15 % conflict cycles otherwise)
// (the spare row of piece 0 stays zero: it is what dropped pairs pick).  The two 16-byte halves of a
// row are swapped in rows with bit 3 set: the 16 lanes a ds_read_b128 services together hold 16
// consecutive samples (lane permutation below), whose rows r and r+8 would otherwise share banks.
// The DMA keeps the image linear and applies the swap on its per-lane SOURCE address; the pick table
// holds the byte offset of half 0 of the picked row, half 1 is that offset ^ 16.
// ---------------------------------------------------------------------------
#define KQ_GS 1056          // bytes between the trace groups inside a piece (32 rows x 32 B + spare row)
#define KQ_ZERO 1024        // byte offset of the all-zero row (spare row of piece 0, group 0; + g * KQ_GS)
// ring slots for an XB-trace output tile (XB + 15 traces are live, in whole 8-trace groups) and bytes per
// 32-row piece: the groups + pad to a multiple of 256 B (rows of neighbouring pieces then keep distinct
// banks inside one ds_read_b128 lane group)
__host__ __device__ constexpr int kq_ring_slots(int xb) { return ((xb + 15 + 7) / 8) * 8; }
__host__ __device__ constexpr int kq_piece_bytes(int xb) { return ((kq_ring_slots(xb) / 8 * KQ_GS + 255) / 256) * 256; }
// ... and with LK extra ring groups (deeper staging lookahead, see kirch_quad_kernel)
__host__ __device__ constexpr int kq_piece_bytes_lk(int xb, int lk) { return (((kq_ring_slots(xb) / 8 + lk) * KQ_GS + 255) / 256) * 256; }
static_assert(kq_piece_bytes_lk(40, 0) == 7424 && kq_piece_bytes_lk(40, 1) == 8448 && kq_piece_bytes_lk(24, 0) == 5376, "");
static_assert(kq_ring_slots(24) == 40 && kq_piece_bytes(24) == 5376 && kq_ring_slots(32) == 48 && kq_piece_bytes(32) == 6400 &&
              kq_ring_slots(40) == 56 && kq_piece_bytes(40) == 7424, "");
#ifndef KQ_DEFAULT_NH
#define KQ_DEFAULT_NH 1     // output tiles per workgroup of the quad kernel (IMPDAR_KIRCH_NH overrides)
#endif
#ifndef KD_DEFAULT_NH
#define KD_DEFAULT_NH 1     // the same for the float64 ring kernel (IMPDAR_KIRCH_NHD), with tiles of KD_DEFAULT_XB2 traces
#endif
#ifndef KD_DEFAULT_XB2
#define KD_DEFAULT_XB2 16
#endif
#ifndef KQ_DEFAULT_LK
#define KQ_DEFAULT_LK 0     // extra ring groups / blocks of staging lookahead with NH >= 2 (IMPDAR_KIRCH_LK overrides)
#endif
#ifndef KQ_PER
#define KQ_PER 4            // quads per interleave slice (1..7 all measure within 2 %; 4 keeps 97 VGPRs)
#endif
typedef float kq_f4 __attribute__((ext_vector_type(4)));
typedef unsigned kq_u4 __attribute__((ext_vector_type(4)));

// NH > 1: NH tiles of XB output traces share ONE ring in one workgroup of 256 NH threads (waves 4h .. 4h+3 own
// tile h = outputs x0 + h XB ...).  Tile h walks the trace offset XB steps BEHIND tile h - 1
// (n_h = n_0 - h XB), so that at every step all tiles read the same XB + 16 ring slots: the staged traces, the
// DMA that brings them and the barrier are shared, only the pick row (h XB / 8 rows behind, re-read from L2 a few
// blocks after the neighbour tile fetched it) and the obliquity factor differ.  The staging window grows by the
// moveout over (NH - 1) XB traces.  Per output this stages NH x fewer image bytes and puts 4 NH waves on the CU.
//
// LK = 1: one more 8-trace group in the ring, and the staging DMA runs one block further ahead (it then has two
// blocks, ~2.6 us, to land instead of one).  What the kernel loses to vector memory is not bytes (NH = 2 halves
// the fabric traffic and gains nothing) but the tail of the load latency: s_waitcnt vmcnt retires loads in issue
// order and every wave must see its DMA landed in front of each block's barrier, so one slow fetch parks four
// (or 4 NH) waves.  The extra group only fits with NH >= 2 (one workgroup per CU, LDS to spare).  The pick rows
// and window lookups are ordinary loads the compiler counts; their pipeline is deepened to PD blocks so that
// each is older than everything the block's wait leaves in flight (the compiler then adds no waits of its own,
// which would drain the younger DMA as well).
template <int XB, bool NEAR, int OCC, int SH, int NH, int LK>
__global__ __launch_bounds__(KF_THREADS * NH, OCC) void kirch_quad_kernel(FastParams P, int W)
{
    constexpr int S = 8;
    constexpr int RG = kq_ring_slots(XB) + 8 * LK;     // ring slots
    constexpr int KQ_PS = kq_piece_bytes_lk(XB, LK);
    constexpr int PD = 2 * LK + 1;            // blocks between the issue of a pick row / window lookup and its first use
    static_assert(LK == 0 || LK == 1, "staging lookahead of one or two blocks");
    static_assert(LK == 0 || !NEAR, "the deeper ring is not laid out for the second (near field) image");
    constexpr int G0 = XB / 8;                // a block's new traces belong to image / ring group blk + G0
    constexpr int KQ_PARTS = (XB / 4 + 1 + KQ_PER - 1) / KQ_PER;   // interleave slices per step
    constexpr int NB = RG / S;                // step blocks per ring revolution (unroll length)
    constexpr int NQ = RG / 4;                // slot quads
    static_assert(XB + 2 * S - 1 <= RG && RG % S == 0 && XB % 4 == 0, "ring too small");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int npieces = W >> 5;               // W is a multiple of 32
    const unsigned img_bytes = (unsigned)npieces * KQ_PS;      // one image (gradient; data image behind it)

    const int tid = threadIdx.x & (KF_THREADS - 1);                  // lane-and-wave index inside the tile
    const int half = NH > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : 0;   // which tile of the workgroup
    // P.queue != null: persistent workgroups.  As many workgroups as fit the chip are launched, and each keeps
    // pulling (chunk, tile) items: first from the list of the XCD it runs on (the balanced tile map, longest
    // first), then, when that is empty, from the other XCDs' lists.  The XCDs of one MI355X differ by ~3 % in
    // speed on this loop and blocks are dealt to them statically, so with one item per block the slowest XCD set
    // the kernel's end; the queue also closes the gaps between workgroups and shortens the tail.
    bool first_item = true;
    int *item_slot = reinterpret_cast<int *>(lds) + (size_t)(img_bytes / 4) * (NEAR ? 2 : 1);
    for (;;) {
    int chunk, xt;
    // part: which piece of the tile's aperture walk this item is.  Plans of 4+ ranks cut every walk into 2 or 4
    // pieces at fixed offsets (n = 1 and n = 1 -+ 56 k: whole ring revolutions, the same for every tile of a
    // chunk) -- one walk of a shallow chunk is as long as such a rank's whole step should be.  A piece writes its
    // sums to its own partial image; kirch_combine_kernel adds the pieces in a fixed order.
    int part = 0;
    if (P.queue) {
        __syncthreads();                       // every wave is done with the previous item (ring reads, the slot)
        if (threadIdx.x == 0) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            const int total = P.nchunks * P.tiles_per_xcd;
            int it = -1;
            for (int v = 0; v < 8 && it < 0; ++v) {
                const int x = (int)((xcc + v) & 7u);
                for (;;) {
                    const int i2 = atomicAdd(P.queue + x * 16, 1);
                    const int i = i2 >> P.parts_log2;            // many-rank plans: 2 or 4 queue items per tile
                    if (i >= total) break;
                    item_slot[2] = i2 & ((1 << P.parts_log2) - 1);
                    const int t = (int)P.tilemap[(size_t)i * 8 + x];
                    if (t >= 0 && t < P.nxt) {
                        it = ((i / P.tiles_per_xcd) << 16) | t;
                        break;
                    }
                }
            }
            item_slot[0] = it;
        }
        __syncthreads();
        const int it = item_slot[0];
        part = item_slot[2];
        if (it < 0) break;
        chunk = it >> 16;
        xt = it & 0xffff;
    } else {
        const int b = blockIdx.x;
        const int xcd = b & 7, r = b >> 3;
        chunk = r / P.tiles_per_xcd;
        const int qx = r - chunk * P.tiles_per_xcd;
        if (chunk >= P.nchunks) return;
        xt = P.tilemap ? (int)P.tilemap[((size_t)chunk * P.tiles_per_xcd + qx) * 8 + xcd]
                       : ((qx / P.G) * 8 + xcd) * P.G + (qx % P.G);
        if (xt < 0 || xt >= P.nxt) return;
    }
    const int s0 = chunk * KF_THREADS;
    const int x0w = (P.xlo & ~7) + xt * (XB * NH);   // workgroup's first output trace: a multiple of 8 (outputs left of xlo are not stored)
    const int x0 = x0w + half * XB;
    const int snum = P.snum, tnum = P.tnum;
    // ds_read_b128 is serviced in four groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31},
    // {32-35,44-47,52-59}, {36-43,48-51,60-63}).  Give every group 16 CONSECUTIVE samples: their
    // picks then span < 16 rows and row*11 mod 16 is a bijection, i.e. no bank conflicts.
    const int lane = tid & 63;
    const int l32 = lane & 31;
    const int grp = ((l32 >= 4 && l32 < 12) || (l32 >= 16 && l32 < 20) || l32 >= 28) ? 1 : 0;
    const int idx = grp ? (l32 < 12 ? l32 - 4 : (l32 < 20 ? l32 - 8 : l32 - 16))
                        : (l32 < 4 ? l32 : (l32 < 16 ? l32 - 8 : l32 - 12));
    const int sigma = (lane & 32) + grp * 16 + idx;
    const int ti_raw = s0 + (tid & ~63) + sigma;
    const int ti = min(ti_raw, snum - 1);

    const int hmax = P.hmax[chunk];
    // first offset: the 8 traces a step block adds must be one aligned 8-row group of the image
    // (ring-relative trace q0 = 8 blk + XB - 1  ->  x0 + nlo = 1 mod 8, i.e. nlo = 1 mod 8); the up to
    // 7 extra leading steps lie outside every aperture of the chunk and pick the all-zero row
    // (the offsets below are tile 0's: the ring clock.  Tile h is at offset n - h XB and needs the clock to run
    // until its own last offset, (NH - 1) XB steps longer)
    int nlo_ = max(-hmax, -(x0w + XB - 1));
    nlo_ -= (nlo_ - 1) & 7;
    int nhi_ = min(hmax + (NH - 1) * XB, tnum - 1 - x0w);
    bool empty_part = false;
    if (P.parts_log2) {
        // piece boundaries (all = 1 mod 8): ..., 1 - U k | 1 - U k .. 0 | 1 .. U k | 1 + U k, ...   (2 pieces: n <= 0 | n >= 1)
        const int uk = (NB * S) * max(1, (hmax + NB * S) / (2 * NB * S));
        const int inner = P.parts_log2 == 2 ? uk : (1 << 28);
        const int side = P.parts_log2 == 2 ? (part >> 1) : part;          // 0: n <= 0, 1: n >= 1
        const bool far = P.parts_log2 == 2 && ((part & 1) ^ side) == 0;    // pieces 0 and 3 are the outer ones
        const int lo = side == 0 ? (far ? -(1 << 28) : 1 - inner) : (far ? 1 + inner : 1);
        const int hi = side == 0 ? (far ? -inner : 0) : (far ? (1 << 28) : inner);
        nlo_ = max(nlo_, lo);
        nhi_ = min(nhi_, hi);
        empty_part = nhi_ < nlo_;
    }
    if (empty_part) continue;                    // uniform; the partial images are zeroed before the launch
    const int nlo = nlo_;
    const int nhi = nhi_;
    const int nsteps = nhi - nlo + 1;
    const int nblocks = (nsteps + S - 1) / S;
    // the main loop always runs whole ring revolutions (NB blocks); steps past the
    // aperture pick the table's all-zero row, so they add nothing
    const int nrev = (nblocks + NB - 1) / NB;
    const int jbase = x0w + nlo;
    // table row of block 0 (block b covers n = nlo + 8 b .. + 7); rows past the tables' end (only the
    // padding blocks of the last revolution can get there) are clamped to the last, all-dropped row
    const int mrow = ((nlo - 1) >> 3) + P.mrow0;
    auto row_of = [&](int blk) { return min(mrow + blk, P.nrows - 1); };
    // this tile's pick rows: half * XB offsets behind the clock (the table keeps 8 all-dropped rows at its start)
    const int mrow_h = mrow - half * (XB / 8);
    // (blocks past the walk -- the padding of the last ring revolution -- read the table's last, all-dropped row: after
    // a piece of a walk they would otherwise be offsets that the next piece owns)
    auto prow_of = [&](int blk) { return blk >= nblocks ? P.nrows - 1 : max(min(mrow_h + blk, P.nrows - 1), 0); };

    // Staging window of the 8 traces block `blk_for` adds: one 8-byte lookup, issued ONE BLOCK EARLIER
    // than its use: s_waitcnt vmcnt counts in order, so waiting for a lookup issued in the same block
    // would also wait for the pick loads just in front of it (a full miss latency per block).
    const int2 *WIN = P.WIN + (size_t)chunk * P.nrows;
    auto fetch_for = [&](int blk_for, int &a, int &b) {
        const int2 w = WIN[max(row_of(blk_for), 0)];
        a = w.x;
        b = w.y;
    };
    // The image is stored in groups of 8 traces, sample-major inside a group (see PrepParams::i8):
    // sample k of the 8 traces of a step block is 32 contiguous bytes.  Raw buffers based at this
    // workgroup's first group: scalar group offset + per-lane sample offset, no address arithmetic
    // in vector registers; traces outside the profile are zero rows of the padded image.
    const unsigned grp_bytes = (unsigned)snum * 32u;
    // buffer descriptors as four scalar words (the DMA below is inline asm): base, no stride, 2 GiB range, raw dword format
    auto make_desc = [&](const float *img) {
        const unsigned long long a = (unsigned long long)(img + (ptrdiff_t)(jbase - 1) * snum);
        kq_u4 d;
        d.x = __builtin_amdgcn_readfirstlane((unsigned)a);
        d.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
        d.z = 0x7fffffffu;
        d.w = 0x00020000u;
        return d;
    };
    const kq_u4 gdesc = make_desc(P.GT);
    const kq_u4 ddesc = make_desc(NEAR ? P.DT : P.GT);
    // Pick table entry = LDS offset of the picked sample's ring row (bytes >> SH); pairs the reference
    // drops (t > t_max, or the 0/0 apex of a t = 0 sample) point at the all-zero row, so they need no
    // compare/select.  The 8 picks of a step block are 16 contiguous bytes per lane: one raw-buffer load
    // (scalar row offset + lane offset) per block.
    const __amdgpu_buffer_rsrc_t tkres =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(P.TKB), 0, 0x7fffffff, 0x00020000);
    const unsigned tioff = (unsigned)ti * 16u;
    const unsigned rowbytes = (unsigned)kq_tkb_stride(snum) * 16u;
    auto picks = [&](int blk) -> kq_u4 {
        return __builtin_amdgcn_raw_buffer_load_b128(tkres, tioff, (unsigned)prow_of(blk) * rowbytes, 0);
    };
#define KQ_TK(q, s) (((q)[(s) >> 1] >> (16 * ((s) & 1))) & 0xffffu)
    const float c1 = P.c1[ti], c2 = NEAR ? P.c2[ti] : 0.f, fin = P.fin[ti];
    // n^2 of a step: scalar multiply + one conversion (in-aperture offsets are < 65536, so the product
    // fits 32 bits; padding steps may wrap, they only ever meet the all-zero row)
    const int nlo_h = nlo - half * XB;
    auto n2_of = [&](int step) {
        const int n = nlo_h + step;
        return (float)((unsigned)n * (unsigned)n);
    };

    if (first_item) {
        // (later items of a persistent workgroup keep the image: every row a pick can point at is re-staged before
        // it is read, and the all-zero spare row is never written)
        if ((unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)lds != 0u) __builtin_trap();
        for (int e = threadIdx.x; e < (int)(img_bytes / 4) * (NEAR ? 2 : 1); e += KF_THREADS * NH) lds[e] = 0.f;
        __syncthreads();
        first_item = false;
    }
    // Pick rows are requested TWO blocks ahead (KQ_PICK_AHEAD = 2): a row that misses the XCD's L2 (the table is
    // 61 MB, a chunk's slice 3.8 MB) takes longer than one block (~1.3 us) to arrive, and with one block of
    // lookahead that latency sat in front of every barrier (a build whose pick rows were always L2-hot ran
    // 0.74 ms faster at config 3).  The wait in front of the barrier then leaves the two youngest loads (the
    // pick row and the staging window issued at the top of the block) in flight: s_waitcnt vmcnt counts in
    // order, so vmcnt(2) still retires the staging DMA issued behind the previous barrier.
    kq_u4 tkc = picks(0);                          // picks of the current block
    kq_u4 tkp[PD + 1];                             // ... of the next one (tkp[0]) and of those behind it
#pragma unroll
    for (int j = 0; j < PD; ++j) tkp[j] = picks(1 + j);

    // accumulators in quads: the ordering pin below takes them as XB/4 operands of ONE asm statement
    kq_f4 acc4[XB / 4];
#pragma unroll
    for (int i = 0; i < XB / 4; ++i) acc4[i] = kq_f4{0.f, 0.f, 0.f, 0.f};
#define KQ_ACC(i) acc4[(i) >> 2][(i) & 3]

    // ---- staging by LDS-DMA: the 8 traces a block adds (one 8-row group of the image) go straight from
    // memory into their ring group, 1 KiB (32 rows) per wave instruction, no staging registers and no
    // ds_write.  Lane l of a piece writes bytes [16 l, 16 l + 16) of the piece: row 32 piece + l/2,
    // half (l & 1); its SOURCE is the sample congruent to that row in [kmin, kmin + W) and the half
    // the row's swap puts there.  Pieces are dealt to the four waves round robin.
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // wave index in the workgroup, in a scalar register
    const int rl = lane >> 1;                                  // row of this lane inside a piece
    const unsigned hsel = ((unsigned)(lane & 1) ^ ((unsigned)(rl >> 3) & 1u)) * 16u;
    // The DMA is inline asm on purpose: hipcc treats its own LDS-DMA as a pending LDS write and drains it
    // (s_waitcnt vmcnt) in front of the next ds_read, which would expose the whole load latency in
    // every block.  Here the order is ours: each wave retires its DMA with the vmcnt(0) in front of the
    // block's barrier, and the new traces are only read after that barrier.  M0 (the LDS destination)
    // is saved and restored inside the statement.
    auto dma16 = [&](unsigned lds_dst, unsigned vo, kq_u4 desc, unsigned so) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(lds_dst), "v"(vo), "s"(desc), "s"(so)
                     : "memory");
    };
    auto dma_issue = [&](int blk_for, int wa) {     // traces that block `blk_for` adds to the ring
        const int kmin = wa & 0xffff, kmod = (int)((unsigned)wa >> 16);
        const int gidx = ((blk_for + G0) % NB + NB) % NB;      // ring group of these traces
        const unsigned so = (unsigned)(blk_for + G0) * grp_bytes;     // image group of trace jbase + q0
        for (int pc = wv; pc < npieces; pc += 4 * NH) {
            int t = pc * 32 + rl - kmod;
            t += (t < 0) ? W : 0;
            const int c = min(kmin + t, snum - 1);
            const unsigned vo = (unsigned)c * 32u + hsel;
            dma16((unsigned)(pc * KQ_PS + gidx * KQ_GS), vo, gdesc, so);
            if (NEAR) dma16(img_bytes + (unsigned)(pc * KQ_PS + gidx * KQ_GS), vo, ddesc, so);
        }
    };
    // ring position of ring-relative trace q is (q + 1) % RG: the 8 traces a block adds are one ring
    // group; the ring starts with the four groups q = -1 .. 30
    int wa, wb;
    for (int pb = -G0; pb <= 0; ++pb) {
        fetch_for(pb, wa, wb);
        dma_issue(pb, wa);
    }
    int wl[LK + 1], wp[PD];                        // windows of the DMAs issued in the prologue / in the first PD blocks
#pragma unroll
    for (int j = 0; j <= LK; ++j) fetch_for(1 + j, wl[j], wb);
#pragma unroll
    for (int j = 0; j < PD; ++j) fetch_for(2 + LK + j, wp[j], wb);
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
    __syncthreads();
#pragma unroll
    for (int j = 0; j <= LK; ++j) dma_issue(1 + j, wl[j]);     // block 1's (and 2's) traces land during block 0
    // DMA instructions this wave issues per block (its share of the pieces): the in-order wait in front of the
    // barrier leaves one block's worth of them in flight when LK = 1
    const int dma_c = __builtin_amdgcn_readfirstlane((npieces - wv + 4 * NH - 1) / (4 * NH));

    // ---- the resident ring is read software pipelined: the ds_read_b128 of step s+1 are in flight
    // while the FMAs of step s run (two statically indexed quad buffers), also across block ends
    kq_f4 va[NQ], vb[NQ], ua[NEAR ? NQ : 1], ub[NEAR ? NQ : 1];
    auto needed = [](int pm, int qd) {
        bool any = false;
        for (int c = 0; c < 4; ++c) any = any || ((4 * qd + c - pm - 1 + 2 * RG) % RG) < XB;
        return any;
    };
    // pm = step index mod RG (a compile-time constant after unrolling), tk = the step's table entry =
    // byte offset of half 0 of the picked row; quad qd = ring slots 4 qd .. 4 qd + 3 = half qd & 1 of group qd >> 1
    // position of quad qd in the step's run of 6-7 needed quads (they are consecutive, cyclically)
    auto run_pos = [](int pm, int qd) { return (qd - ((pm + 1) % RG) / 4 + NQ) % NQ; };
    // `part` selects a slice of the run (KQ_PARTS slices; -1 = all): reads and FMAs are interleaved slice
    // by slice so that a wave's ds_read_b128 do not arrive at the LDS queue as one burst of 7
    auto load_step = [&](int pm, unsigned tk, kq_f4 (&v)[NQ], kq_f4 (&u)[NEAR ? NQ : 1], int part) {
        // LDS pointers built from the table's byte offsets directly: the dynamic LDS block of this kernel
        // starts at LDS address 0 (no static __shared__; checked once in the prologue), so no base is added
        typedef const __attribute__((address_space(3))) kq_f4 *lds_f4p;
        const unsigned a0 = tk << SH, a1 = a0 ^ 16u;
#pragma unroll
        for (int qd = 0; qd < NQ; ++qd)
            if (needed(pm, qd) && (part < 0 || run_pos(pm, qd) / KQ_PER == part)) {
                // must stay a whole ds_read_b128 also for the partly used quads at the ends of
                // the window: fma_step marks the unused components as used (empty asm)
                const unsigned src = ((qd & 1) ? a1 : a0) + (qd >> 1) * KQ_GS;
                v[qd] = *(lds_f4p)(uintptr_t)src;
                if (NEAR) u[qd] = *(lds_f4p)(uintptr_t)(src + img_bytes);
            }
    };
    auto fma_step = [&](int pm, float w, float w2, const kq_f4 (&v)[NQ], const kq_f4 (&u)[NEAR ? NQ : 1], int part) {
#pragma unroll
        for (int qd = 0; qd < NQ; ++qd) {
            if (part >= 0 && run_pos(pm, qd) / KQ_PER != part) continue;
            // outputs served by slots 4qd .. 4qd+3 at this step
            const int i0 = (4 * qd + 0 - pm - 1 + 2 * RG) % RG;
            const int i1 = (4 * qd + 1 - pm - 1 + 2 * RG) % RG;
            const int i2 = (4 * qd + 2 - pm - 1 + 2 * RG) % RG;
            const int i3 = (4 * qd + 3 - pm - 1 + 2 * RG) % RG;
            const bool any = i0 < XB || i1 < XB || i2 < XB || i3 < XB;
            if (any) {
                // components of an edge quad that serve no output are handed to an empty asm: the
                // load stays a whole ds_read_b128 (split into b32 pieces it bank-conflicts 4-way)
                // at no instruction cost
#define KQ_COMP(ix, c)                                                              \
    if (ix < XB) {                                                                  \
        KQ_ACC(ix < XB ? ix : 0) = fmaf(w, v[qd].c, KQ_ACC(ix < XB ? ix : 0));            \
        if (NEAR) KQ_ACC(ix < XB ? ix : 0) = fmaf(w2, u[qd].c, KQ_ACC(ix < XB ? ix : 0)); \
    } else {                                                                        \
        asm volatile("" ::"v"(v[qd].c));                                            \
        if (NEAR) asm volatile("" ::"v"(u[qd].c));                                  \
    }
                KQ_COMP(i0, x)
                KQ_COMP(i1, y)
                KQ_COMP(i2, z)
                KQ_COMP(i3, w)
#undef KQ_COMP
            }
        }
    };
    // The FMAs carry no chain, so instruction selection is free to sink a whole block's
    // FMAs below all of its LDS reads (which then spill).  Passing the accumulators
    // through an empty volatile asm (with a memory clobber) after every group of reads and every
    // step's FMAs pins the order reads(s+1) -> FMAs(s) without consuming any load result early.
#define KQ_PIN()                                                                                          \
    do {                                                                                                  \
        static_assert(XB == 24 || XB == 32 || XB == 40, "KQ_PIN lists XB/4 accumulator quads");           \
        if (XB == 24)                                                                                     \
            asm volatile("" : "+v"(acc4[0]), "+v"(acc4[1]), "+v"(acc4[2]), "+v"(acc4[3]), "+v"(acc4[4]), \
                              "+v"(acc4[5]) :: "memory");                                                 \
        else if (XB == 32)                                                                                \
            asm volatile("" : "+v"(acc4[0]), "+v"(acc4[1]), "+v"(acc4[2]), "+v"(acc4[3]), "+v"(acc4[4]), \
                              "+v"(acc4[5]), "+v"(acc4[XB >= 32 ? 6 : 0]), "+v"(acc4[XB >= 32 ? 7 : 0])   \
                         :: "memory");                                                                    \
        else                                                                                              \
            asm volatile("" : "+v"(acc4[0]), "+v"(acc4[1]), "+v"(acc4[2]), "+v"(acc4[3]), "+v"(acc4[4]), \
                              "+v"(acc4[5]), "+v"(acc4[XB >= 32 ? 6 : 0]), "+v"(acc4[XB >= 32 ? 7 : 0]),  \
                              "+v"(acc4[XB >= 40 ? 8 : 0]), "+v"(acc4[XB >= 40 ? 9 : 0]) :: "memory");    \
    } while (0)

    float n2c[S];                                  // n^2 of the current block's steps
#pragma unroll
    for (int s = 0; s < S; ++s) n2c[s] = n2_of(s);
    load_step(0, KQ_TK(tkc, 0), va, ua, -1);       // step 0 of block 0
    for (int rev = 0; rev < nrev; ++rev) {
#pragma clang loop unroll(full)
        for (int bb = 0; bb < NB; ++bb) {              // block index within the ring revolution
            const int blk = rev * NB + bb;
            const int pm0 = bb * S;                    // step index mod RG of the block's first step
            // ---- loads: the next block's table entries and, by DMA, its 8 new traces.  Their ring group
            // held traces last read at step 6 of the previous block, and every wave is past that block's
            // barrier, which sits after step 6: no wave can still be reading them.
            tkp[PD] = picks(blk + PD + 1);
            int wn = 0;
            fetch_for(blk + 2 + LK + PD, wn, wb);      // staging window of a DMA issued PD blocks from now
            // obliquity cos(theta) of this block's steps
            float twc[S], tw2c[NEAR ? S : 1];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float y = __builtin_amdgcn_rsqf(fmaf(c1, n2c[s], 1.0f));
                twc[s] = y;
                if (NEAR) tw2c[s] = (y * c2) * (y * y);
            }
            {
#pragma unroll
                for (int s = 0; s < S; ++s) n2c[s] = n2_of((blk + 1) * S + s);
            }
#define KQ_W2(s) (NEAR ? tw2c[NEAR ? (s) : 0] : 0.f)
#define KQ_UNPACK(...) __VA_ARGS__
#define KQ_STEP(L, F)                                                                          \
    do {                                                                                       \
        _Pragma("unroll") for (int part = 0; part < KQ_PARTS; ++part) {                      \
            load_step(KQ_UNPACK L, part); KQ_PIN();                                            \
            fma_step(KQ_UNPACK F, part); KQ_PIN();                                             \
        }                                                                                      \
    } while (0)
            KQ_STEP((pm0 + 1, KQ_TK(tkc, 1), vb, ub), (pm0 + 0, twc[0], KQ_W2(0), va, ua));
            KQ_STEP((pm0 + 2, KQ_TK(tkc, 2), va, ua), (pm0 + 1, twc[1], KQ_W2(1), vb, ub));
            KQ_STEP((pm0 + 3, KQ_TK(tkc, 3), vb, ub), (pm0 + 2, twc[2], KQ_W2(2), va, ua));
            KQ_STEP((pm0 + 4, KQ_TK(tkc, 4), va, ua), (pm0 + 3, twc[3], KQ_W2(3), vb, ub));
            KQ_STEP((pm0 + 5, KQ_TK(tkc, 5), vb, ub), (pm0 + 4, twc[4], KQ_W2(4), va, ua));
            KQ_STEP((pm0 + 6, KQ_TK(tkc, 6), va, ua), (pm0 + 5, twc[5], KQ_W2(5), vb, ub));
            KQ_STEP((pm0 + 7, KQ_TK(tkc, 7), vb, ub), (pm0 + 6, twc[6], KQ_W2(6), va, ua));
            // ---- barrier after step 6.  The new traces are first read by step 0 of the next block, whose
            // reads are issued below; each wave retires its own DMA (and the pick load) first.  The LDS
            // reads of step 7 stay in flight across the barrier, so the read pipeline never drains.
            // a builtin so that hipcc's own wait counting sees it
            if (LK == 0) {
                __builtin_amdgcn_s_waitcnt(0x0F72);    // vmcnt(2): all but this block's pick row + window lookup
            } else if (blk < LK || dma_c == 0) {
                __builtin_amdgcn_s_waitcnt(0x0F70);    // pipeline not yet full (or a wave without a DMA share): everything
            } else if (dma_c == 1) {
                __builtin_amdgcn_s_waitcnt(0x0F75);    // vmcnt(5): two blocks of pick + window lookups and one DMA
            } else {
                __builtin_amdgcn_s_waitcnt(0x0F76);    // vmcnt(6): ... and two DMAs
            }
            asm volatile("s_barrier" ::: "memory");
            // every wave is past step 6 of this block: the ring group of the traces last read there is free
            dma_issue(blk + 2 + LK, wp[0]);
#pragma unroll
            for (int j = 0; j + 1 < PD; ++j) wp[j] = wp[j + 1];
            wp[PD - 1] = wn;
#define KQ_NEXT tkp[0]
            KQ_STEP(((pm0 + 8) % RG, KQ_TK(KQ_NEXT, 0), va, ua), (pm0 + 7, twc[7], KQ_W2(7), vb, ub));
#undef KQ_W2
#undef KQ_STEP
#undef KQ_UNPACK
            tkc = KQ_NEXT;
#undef KQ_NEXT
#pragma unroll
            for (int j = 0; j < PD; ++j) tkp[j] = tkp[j + 1];
        }
    }
#undef KQ_PIN
#undef KQ_TK
    asm volatile("" ::"v"(va[0].x), "v"(va[NQ - 1].w));   // the look-ahead reads of the step after the last
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // the look-ahead DMA must land before the LDS is released

    if (ti_raw < snum) {
        float *o = (P.parts_log2 ? reinterpret_cast<float *>(P.partial) + (size_t)part * P.part_stride : P.out) +
                   (size_t)ti_raw * P.ldo + (x0 - P.xlo);
#pragma unroll
        for (int i = 0; i < XB; ++i)
            if (x0 + i >= P.xlo && x0 + i < P.xhi) o[i] = KQ_ACC(i) * fin;
    }
#undef KQ_ACC
    if (!P.queue) break;
    }   // item loop
}

// ---------------------------------------------------------------------------
// dquad kernel: the quad kernel's ring in float64 -- the float64 parity path on uniform grids (what
// RadarData.migrate('kirch') runs on a float64 radargram, mig_python.py:35-60,118).
//
// Same LDS image byte for byte (a ring row is 32 bytes: here 4 traces x 8 bytes, so the image is kept in groups
// of FOUR traces and a step block is 4 steps), same LDS-DMA staging, same fp64 pick table (the reference's
// picks, t > t_max test and 0/0 skip, kirch_tabled_kernel below); a ds_read_b128 now carries the samples of two
// traces.  What differs from the float32 kernel:
//   * XB = 20 (or 16) outputs per lane in float64 accumulators (XB/2 register quads, as many as the 40 floats);
//   * the obliquity factor cos(theta) = zs/rs = rsqrt(1 + (n dx / zs)^2) is evaluated in float64 per (lane,
//     step): a float32 v_rsq seed and ONE third-order correction y0 (1 + r/2 + 3 r^2/8), r = 1 - x y0^2, which
//     leaves < 1e-20 of truncation on top of float64 rounding -- far inside the 1e-12 bar, without the 113 MB
//     float64 weight table of kirch_exact_tab_kernel or an IEEE sqrt + divide per step;
//   * the factor sign(zs)/(2 pi v) is applied once at the end, in float64 (the near-field weight cos/rs^2 is
//     carried in those units: y^3 v / zs^2);
//   * non-finite input: only NaNs are zeroed by prep (they are what nansum drops, :53,:58); infinities travel
//     through the sum as in the reference.  Rows with zs = 0 are written as exact zeros (:60 with cos = 0).
// The kernel is bound by the float64 vector rate (20 v_fma_f64 + ~10 for the weight per lane and step against
// 11 ds_read_b128), not by the LDS port.
// ---------------------------------------------------------------------------
__host__ __device__ constexpr int kd_ring_slots(int xb) { return ((xb + 7 + 3) / 4) * 4; }     // XB + 2 S - 1 live traces, whole groups
__host__ __device__ constexpr int kd_piece_bytes(int xb) { return ((kd_ring_slots(xb) / 4 * KQ_GS + 255) / 256) * 256; }
static_assert(kd_ring_slots(20) == 28 && kd_piece_bytes(20) == 7424 && kd_ring_slots(16) == 24 && kd_piece_bytes(16) == 6400, "");
typedef double kd_d2 __attribute__((ext_vector_type(2)));

// picks for the dquad kernel: rows of FOUR offsets, n = 4 (r - mrow0) + 1 + s, 16 bits each (see kirch_tableq_kernel)
__global__ __launch_bounds__(256) void kirch_tabled_kernel(TableQParams P)
{
    const int ti = blockIdx.x * 256 + threadIdx.x;
    const int a = blockIdx.y;
    if (ti >= P.snum) return;
    const unsigned zero_row = 1024u >> P.sh;         // KQ_ZERO
    unsigned pk[5];
    bool out = false;
    const double zs = P.zs[ti], zs2 = P.zs2[ti];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int n = 4 * a + s;
        unsigned kb = zero_row;
        if (n < P.nmax && !out) {
            const double dx = (double)n * P.dx;
            const double q = dx * dx + zs2;                       // mig_python.py:44
            const double rs = sqrt(q);
            const double cost = zs / rs;                          // :47
            const double t = 2.0 * rs / P.vel;                    // :49
            if (t > P.tmax) {                                     // :52
                out = true;
            } else if (cost == cost) {
                const int ns = P.snum;
                int k0 = (int)floor((t - P.tt0) * P.inv_dt);
                k0 = min(max(k0, 0), ns - 1);
                while (k0 < ns - 1 && P.tt[k0 + 1] <= t) ++k0;
                while (k0 > 0 && P.tt[k0] > t) --k0;
                const int k1 = min(k0 + 1, ns - 1);
                const int k = (fabs(P.tt[k1] - t) < fabs(P.tt[k0] - t)) ? k1 : k0;
                kb = kq_row_offset(k % P.wmod, P.ps) >> P.sh;
            }
        }
        pk[s] = kb;
    }
    uint2 *T = reinterpret_cast<uint2 *>(P.TKB);
    const int rp = a + P.mrow0, rn = P.mrow0 - a - 1;
    if (rp < P.nrows) T[(size_t)rp * kq_tkb_stride(P.snum) + ti] = make_uint2(pk[1] | (pk[2] << 16), pk[3] | (pk[4] << 16));
    if (rn >= 0) T[(size_t)rn * kq_tkb_stride(P.snum) + ti] = make_uint2(pk[3] | (pk[2] << 16), pk[1] | (pk[0] << 16));
}

template <int XB, bool NEAR, int OCC, int SH, int NH>
__global__ __launch_bounds__(KF_THREADS * NH, OCC) void kirch_dquad_kernel(FastParams P, int W)
{   // NH tiles of XB traces per workgroup on one ring, tile h walking h XB offsets behind tile 0: see kirch_quad_kernel
    constexpr int S = 4;
    constexpr int RG = kd_ring_slots(XB);     // ring slots (traces)
    constexpr int KQ_PS = kd_piece_bytes(XB);
    constexpr int G0 = XB / 4;                // a block's new traces belong to image / ring group blk + G0
    constexpr int KQ_PARTS = (XB / 2 + 1 + KQ_PER - 1) / KQ_PER;   // interleave slices per step
    constexpr int NB = RG / S;                // step blocks per ring revolution (unroll length)
    constexpr int NQ = RG / 2;                // 16-byte slot pairs
    static_assert(XB + 2 * S - 1 <= RG && RG % S == 0 && XB % 4 == 0, "ring too small");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int npieces = W >> 5;
    const unsigned img_bytes = (unsigned)npieces * KQ_PS;

    const int tid = threadIdx.x & (KF_THREADS - 1);
    const int half = NH > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : 0;
    // persistent workgroups pulling (chunk, tile) items from per-XCD queues: see kirch_quad_kernel
    bool first_item = true;
    int *item_slot = reinterpret_cast<int *>(lds) + (size_t)(img_bytes / 4) * (NEAR ? 2 : 1);
    for (;;) {
    int chunk, xt;
    int part = 0;            // see kirch_quad_kernel: which piece of the tile's aperture walk (plans of 4+ ranks)
    if (P.queue) {
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            const int total = P.nchunks * P.tiles_per_xcd;
            int it = -1;
            for (int v = 0; v < 8 && it < 0; ++v) {
                const int x = (int)((xcc + v) & 7u);
                for (;;) {
                    const int i2 = atomicAdd(P.queue + x * 16, 1);
                    const int i = i2 >> P.parts_log2;            // many-rank plans: 2 or 4 queue items per tile
                    if (i >= total) break;
                    item_slot[2] = i2 & ((1 << P.parts_log2) - 1);
                    const int t = (int)P.tilemap[(size_t)i * 8 + x];
                    if (t >= 0 && t < P.nxt) {
                        it = ((i / P.tiles_per_xcd) << 16) | t;
                        break;
                    }
                }
            }
            item_slot[0] = it;
        }
        __syncthreads();
        const int it = item_slot[0];
        part = item_slot[2];
        if (it < 0) break;
        chunk = it >> 16;
        xt = it & 0xffff;
    } else {
        const int b = blockIdx.x;
        const int xcd = b & 7, r = b >> 3;
        chunk = r / P.tiles_per_xcd;
        const int qx = r - chunk * P.tiles_per_xcd;
        if (chunk >= P.nchunks) return;
        xt = P.tilemap ? (int)P.tilemap[((size_t)chunk * P.tiles_per_xcd + qx) * 8 + xcd]
                       : ((qx / P.G) * 8 + xcd) * P.G + (qx % P.G);
        if (xt < 0 || xt >= P.nxt) return;
    }
    const int s0 = chunk * KF_THREADS;
    const int x0w = (P.xlo & ~3) + xt * (XB * NH);   // workgroup tiles start at a multiple of 4 (outputs left of xlo are not stored)
    const int x0 = x0w + half * XB;
    const int snum = P.snum, tnum = P.tnum;
    // lane -> sample permutation: the 16 lanes a ds_read_b128 services together hold 16 consecutive samples
    // (see kirch_quad_kernel)
    const int lane = tid & 63;
    const int l32 = lane & 31;
    const int grp = ((l32 >= 4 && l32 < 12) || (l32 >= 16 && l32 < 20) || l32 >= 28) ? 1 : 0;
    const int idx = grp ? (l32 < 12 ? l32 - 4 : (l32 < 20 ? l32 - 8 : l32 - 16))
                        : (l32 < 4 ? l32 : (l32 < 16 ? l32 - 8 : l32 - 12));
    const int sigma = (lane & 32) + grp * 16 + idx;
    const int ti_raw = s0 + (tid & ~63) + sigma;
    const int ti = min(ti_raw, snum - 1);

    const int hmax = P.hmax[chunk];
    int nlo_ = max(-hmax, -(x0w + XB - 1));      // tile 0's offsets are the ring clock
    nlo_ -= (nlo_ - 1) & 3;                      // x0w + nlo = 1 mod 4: a block's 4 new traces are one image group
    int nhi_ = min(hmax + (NH - 1) * XB, tnum - 1 - x0w);
    bool empty_part = false;
    if (P.parts_log2) {                          // piece boundaries: see kirch_quad_kernel
        const int uk = (NB * S) * max(1, (hmax + NB * S) / (2 * NB * S));
        const int inner = P.parts_log2 == 2 ? uk : (1 << 28);
        const int side = P.parts_log2 == 2 ? (part >> 1) : part;
        const bool far = P.parts_log2 == 2 && ((part & 1) ^ side) == 0;
        const int lo = side == 0 ? (far ? -(1 << 28) : 1 - inner) : (far ? 1 + inner : 1);
        const int hi = side == 0 ? (far ? -inner : 0) : (far ? (1 << 28) : inner);
        nlo_ = max(nlo_, lo);
        nhi_ = min(nhi_, hi);
        empty_part = nhi_ < nlo_;
    }
    if (empty_part) continue;                    // uniform; the partial images are zeroed before the launch
    const int nlo = nlo_;
    const int nhi = nhi_;
    const int nsteps = nhi - nlo + 1;
    const int nblocks = (nsteps + S - 1) / S;
    const int nrev = (nblocks + NB - 1) / NB;
    const int jbase = x0w + nlo;
    const int mrow = ((nlo - 1) >> 2) + P.mrow0;
    auto row_of = [&](int blk) { return min(mrow + blk, P.nrows - 1); };
    const int mrow_h = mrow - half * (XB / 4);   // this tile's pick rows: half * XB offsets behind the clock
    auto prow_of = [&](int blk) { return blk >= nblocks ? P.nrows - 1 : max(min(mrow_h + blk, P.nrows - 1), 0); };

    const int2 *WIN = P.WIN + (size_t)chunk * P.nrows;
    auto fetch_for = [&](int blk_for, int &a, int &b) {
        const int2 w = WIN[max(row_of(blk_for), 0)];
        a = w.x;
        b = w.y;
    };
    const unsigned grp_bytes = (unsigned)snum * 32u;
    auto make_desc = [&](const double *img) {
        const unsigned long long a = (unsigned long long)(img + (ptrdiff_t)(jbase - 1) * snum);
        kq_u4 d;
        d.x = __builtin_amdgcn_readfirstlane((unsigned)a);
        d.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
        d.z = 0x7fffffffu;
        d.w = 0x00020000u;
        return d;
    };
    const kq_u4 gdesc = make_desc(reinterpret_cast<const double *>(P.GT));
    const kq_u4 ddesc = make_desc(reinterpret_cast<const double *>(NEAR ? P.DT : P.GT));
    const __amdgpu_buffer_rsrc_t tkres =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(P.TKB), 0, 0x7fffffff, 0x00020000);
    const unsigned tioff = (unsigned)ti * 8u;
    const unsigned rowbytes = (unsigned)kq_tkb_stride(snum) * 8u;
    typedef unsigned kd_u2 __attribute__((ext_vector_type(2)));
    auto picks = [&](int blk) -> kd_u2 {
        return __builtin_amdgcn_raw_buffer_load_b64(tkres, tioff, (unsigned)prow_of(blk) * rowbytes, 0);
    };
#define KD_TK(q, s) (((q)[(s) >> 1] >> (16 * ((s) & 1))) & 0xffffu)
    const double c1 = P.c1d[ti], c2 = NEAR ? P.c2d[ti] : 0.0, fin = P.find[ti];
    const int nlo_h = nlo - half * XB;
    auto n2_of = [&](int step) {
        const int n = nlo_h + step;
        return (double)((unsigned)n * (unsigned)n);
    };
    // cos(theta) = rsqrt(1 + c1 n^2) in float64: float32 seed + one third-order correction (see the header)
    auto cosine = [&](double n2) {
        const double x = fma(c1, n2, 1.0);
        const double y0 = (double)__builtin_amdgcn_rsqf((float)x);
        const double h = x * y0;
        const double rr = fma(-h, y0, 1.0);
        const double pp = fma(rr, 0.375, 0.5) * rr;
        return fma(y0, pp, y0);
    };

    if (first_item) {
        if ((unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)lds != 0u) __builtin_trap();
        for (int e = threadIdx.x; e < (int)(img_bytes / 4) * (NEAR ? 2 : 1); e += KF_THREADS * NH) lds[e] = 0.f;
        __syncthreads();
        first_item = false;
    }
    kd_u2 tkc = picks(0);

    kd_d2 acc2[XB / 2];
#pragma unroll
    for (int i = 0; i < XB / 2; ++i) acc2[i] = kd_d2{0.0, 0.0};
#define KD_ACC(i) acc2[(i) >> 1][(i) & 1]

    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rl = lane >> 1;
    const unsigned hsel = ((unsigned)(lane & 1) ^ ((unsigned)(rl >> 3) & 1u)) * 16u;
    auto dma16 = [&](unsigned lds_dst, unsigned vo, kq_u4 desc, unsigned so) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(lds_dst), "v"(vo), "s"(desc), "s"(so)
                     : "memory");
    };
    auto dma_issue = [&](int blk_for, int wa) {     // the 4 traces that block `blk_for` adds to the ring
        const int kmin = wa & 0xffff, kmod = (int)((unsigned)wa >> 16);
        const int gidx = ((blk_for + G0) % NB + NB) % NB;
        const unsigned so = (unsigned)(blk_for + G0) * grp_bytes;
        for (int pc = wv; pc < npieces; pc += 4 * NH) {
            int t = pc * 32 + rl - kmod;
            t += (t < 0) ? W : 0;
            const int c = min(kmin + t, snum - 1);
            const unsigned vo = (unsigned)c * 32u + hsel;
            dma16((unsigned)(pc * KQ_PS + gidx * KQ_GS), vo, gdesc, so);
            if (NEAR) dma16(img_bytes + (unsigned)(pc * KQ_PS + gidx * KQ_GS), vo, ddesc, so);
        }
    };
    int wa, wb, wn = 0;
    for (int pb = -G0; pb <= 0; ++pb) {
        fetch_for(pb, wa, wb);
        dma_issue(pb, wa);
    }
    fetch_for(1, wa, wb);
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
    __syncthreads();
    dma_issue(1, wa);
    fetch_for(2, wa, wb);

    kq_f4 va[NQ], vb[NQ], ua[NEAR ? NQ : 1], ub[NEAR ? NQ : 1];
    auto needed = [](int pm, int qd) {
        bool any = false;
        for (int c = 0; c < 2; ++c) any = any || ((2 * qd + c - pm - 1 + 2 * RG) % RG) < XB;
        return any;
    };
    auto run_pos = [](int pm, int qd) { return (qd - ((pm + 1) % RG) / 2 + NQ) % NQ; };
    auto load_step = [&](int pm, unsigned tk, kq_f4 (&v)[NQ], kq_f4 (&u)[NEAR ? NQ : 1], int part) {
        typedef const __attribute__((address_space(3))) kq_f4 *lds_f4p;
        const unsigned a0 = tk << SH, a1 = a0 ^ 16u;
#pragma unroll
        for (int qd = 0; qd < NQ; ++qd)
            if (needed(pm, qd) && (part < 0 || run_pos(pm, qd) / KQ_PER == part)) {
                const unsigned src = ((qd & 1) ? a1 : a0) + (qd >> 1) * KQ_GS;
                v[qd] = *(lds_f4p)(uintptr_t)src;
                if (NEAR) u[qd] = *(lds_f4p)(uintptr_t)(src + img_bytes);
            }
    };
    auto fma_step = [&](int pm, double w, double w2, const kq_f4 (&v)[NQ], const kq_f4 (&u)[NEAR ? NQ : 1], int part) {
#pragma unroll
        for (int qd = 0; qd < NQ; ++qd) {
            if (part >= 0 && run_pos(pm, qd) / KQ_PER != part) continue;
            const int i0 = (2 * qd + 0 - pm - 1 + 2 * RG) % RG;
            const int i1 = (2 * qd + 1 - pm - 1 + 2 * RG) % RG;
            if (i0 < XB || i1 < XB) {
                const kd_d2 dv = __builtin_bit_cast(kd_d2, v[qd]);
                const kd_d2 du = __builtin_bit_cast(kd_d2, u[NEAR ? qd : 0]);
#define KD_FMA(a, b, c_) fma(a, b, c_)
#define KD_COMP(ix, c)                                                              \
    if (ix < XB) {                                                                  \
        KD_ACC(ix < XB ? ix : 0) = KD_FMA(w, dv[c], KD_ACC(ix < XB ? ix : 0));         \
        if (NEAR) KD_ACC(ix < XB ? ix : 0) = fma(w2, du[c], KD_ACC(ix < XB ? ix : 0)); \
    } else {                                                                        \
        asm volatile("" ::"v"(dv[c]));                                              \
        if (NEAR) asm volatile("" ::"v"(du[c]));                                    \
    }
                KD_COMP(i0, 0)
                KD_COMP(i1, 1)
#undef KD_COMP
            }
        }
    };
#define KD_PIN()                                                                                          \
    do {                                                                                                  \
        static_assert(XB == 16 || XB == 20, "KD_PIN lists XB/2 accumulator pairs");                       \
        if (XB == 16)                                                                                     \
            asm volatile("" : "+v"(acc2[0]), "+v"(acc2[1]), "+v"(acc2[2]), "+v"(acc2[3]), "+v"(acc2[4]), \
                              "+v"(acc2[5]), "+v"(acc2[6]), "+v"(acc2[7]) :: "memory");                   \
        else                                                                                              \
            asm volatile("" : "+v"(acc2[0]), "+v"(acc2[1]), "+v"(acc2[2]), "+v"(acc2[3]), "+v"(acc2[4]), \
                              "+v"(acc2[5]), "+v"(acc2[6]), "+v"(acc2[7]), "+v"(acc2[XB >= 20 ? 8 : 0]),  \
                              "+v"(acc2[XB >= 20 ? 9 : 0]) :: "memory");                                  \
    } while (0)

    double n2c[S];
#pragma unroll
    for (int s = 0; s < S; ++s) n2c[s] = n2_of(s);
    load_step(0, KD_TK(tkc, 0), va, ua, -1);
    for (int rev = 0; rev < nrev; ++rev) {
#pragma clang loop unroll(full)
        for (int bb = 0; bb < NB; ++bb) {
            const int blk = rev * NB + bb;
            const int pm0 = bb * S;
            const kd_u2 tkn = picks(blk + 1);
            fetch_for(blk + 3, wn, wb);
            double twc[S], tw2c[NEAR ? S : 1];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const double y = cosine(n2c[s]);
                twc[s] = y;
                if (NEAR) tw2c[s] = (y * c2) * (y * y);        // near-field weight in units of fin: c2 = v / zs^2
            }
#pragma unroll
            for (int s = 0; s < S; ++s) n2c[s] = n2_of((blk + 1) * S + s);
#define KD_W2(s) (NEAR ? tw2c[NEAR ? (s) : 0] : 0.0)
#define KD_UNPACK(...) __VA_ARGS__
#define KD_STEP(L, F)                                                                          \
    do {                                                                                       \
        _Pragma("unroll") for (int part = 0; part < KQ_PARTS; ++part) {                      \
            load_step(KD_UNPACK L, part); KD_PIN();                                            \
            fma_step(KD_UNPACK F, part); KD_PIN();                                             \
        }                                                                                      \
    } while (0)
            KD_STEP((pm0 + 1, KD_TK(tkc, 1), vb, ub), (pm0 + 0, twc[0], KD_W2(0), va, ua));
            KD_STEP((pm0 + 2, KD_TK(tkc, 2), va, ua), (pm0 + 1, twc[1], KD_W2(1), vb, ub));
            KD_STEP((pm0 + 3, KD_TK(tkc, 3), vb, ub), (pm0 + 2, twc[2], KD_W2(2), va, ua));
            // barrier after step S - 2: the ring group the next DMA overwrites was last read there
            __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0)
            asm volatile("s_barrier" ::: "memory");
            dma_issue(blk + 2, wa);
            wa = wn;
            KD_STEP(((pm0 + 4) % RG, KD_TK(tkn, 0), va, ua), (pm0 + 3, twc[3], KD_W2(3), vb, ub));
#undef KD_W2
#undef KD_STEP
#undef KD_UNPACK
            tkc = tkn;
        }
    }
#undef KD_PIN
#undef KD_TK
    asm volatile("" ::"v"(va[0].x), "v"(va[NQ - 1].w));
    __builtin_amdgcn_s_waitcnt(0x0F70);

    if (ti_raw < snum) {
        double *o = (P.parts_log2 ? reinterpret_cast<double *>(P.partial) + (size_t)part * P.part_stride
                                  : reinterpret_cast<double *>(P.out)) + (size_t)ti_raw * P.ldo + (x0 - P.xlo);
#pragma unroll
        for (int i = 0; i < XB; ++i)
            if (x0 + i >= P.xlo && x0 + i < P.xhi) o[i] = (fin == 0.0) ? 0.0 : KD_ACC(i) * fin;
    }
#undef KD_ACC
    if (!P.queue) break;
    }   // item loop
}

// sum of the pieces of every aperture walk, in piece order (many-rank plans, see kirch_quad_kernel)
template <typename T, int K>
__global__ __launch_bounds__(256) void kirch_combine_kernel(const T *__restrict__ partial, T *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    T v = partial[i] + partial[n + i];
    if (K == 4) v = (v + partial[2 * n + i]) + partial[3 * n + i];
    out[i] = v;
}

// ===========================================================================
// host side
// ===========================================================================
// set by mig_kirch_loop around its plan creation: the time limit t > t_max drops a pair at (mig_python.py:52) is the
// caller's argument there, not max(tt)
static thread_local const double *g_tmax_override = nullptr;

extern "C" int impdar_kirch_plan_create(impdar_ctx *ctx, int dtype, int snum, int tnum,
                                        const double *dist_m, const double *tt_sec, double vel,
                                        int nearfield, int grad_uniform, double grad_h,
                                        const double *ga, const double *gb, const double *gc,
                                        int mode, int nranks, impdar_kirch_plan **out)
{
    IMPDAR_ARG_CHECK(ctx && out, "null context/plan pointer");
    IMPDAR_ARG_CHECK(dtype == IMPDAR_F32 || dtype == IMPDAR_F64, "dtype must be 0 (f32) or 1 (f64)");
    IMPDAR_ARG_CHECK(snum >= 2 && tnum >= 1, "need snum >= 2 and tnum >= 1 (got %d, %d)", snum, tnum);
    IMPDAR_ARG_CHECK(dist_m && tt_sec, "dist/travel_time must not be null");
    IMPDAR_ARG_CHECK(vel > 0, "vel must be positive");
    IMPDAR_ARG_CHECK(nranks >= 1, "nranks must be >= 1");
    IMPDAR_ARG_CHECK(grad_uniform || (ga && gb && gc), "non-uniform gradient needs ga/gb/gc");

    // (the geometry analysis and the kernel choice below are pure host code and run before the first HIP call,
    // so their argument errors do not need a device: tests/test_sanitizer.py drives them under ASan/UBSan)
    impdar_kirch_plan *p = new impdar_kirch_plan();
    p->ctx = ctx;
    p->dtype = dtype;
    p->snum = snum;
    p->tnum = tnum;
    p->nranks = nranks;
    // equal input shards of whole 8-row groups (the grouped image layout keeps a shard contiguous)
    p->tnum_pad = ((tnum + 8 * nranks - 1) / (8 * nranks)) * 8 * nranks;
    p->nearfield = nearfield ? 1 : 0;
    p->grad_uniform = grad_uniform;
    p->grad_h = grad_h;
    p->vel = vel;

    // ---- geometry analysis ------------------------------------------------
    double tmax = tt_sec[0];
    bool increasing = true;
    for (int k = 1; k < snum; ++k) {
        tmax = std::max(tmax, tt_sec[k]);
        if (!(tt_sec[k] > tt_sec[k - 1])) increasing = false;
    }
    if (g_tmax_override) tmax = *g_tmax_override;      // mig_kirch_loop: the caller's own time limit (see there)
    p->tmax = tmax;
    if (!increasing) {
        delete p;
        impdar_set_error("travel_time must be strictly increasing");
        return IMPDAR_ERR_ARG;
    }
    const double dt = (tt_sec[snum - 1] - tt_sec[0]) / (snum - 1);
    bool uni_t = dt > 0;
    for (int k = 0; k < snum && uni_t; ++k)
        if (std::fabs(tt_sec[k] - (tt_sec[0] + k * dt)) > 1e-9 * dt) uni_t = false;
    double dx = 1.0;
    bool uni_x = true;
    if (tnum >= 2) {
        dx = (dist_m[tnum - 1] - dist_m[0]) / (tnum - 1);
        uni_x = dx > 0;
        for (int j = 0; j < tnum && uni_x; ++j)
            if (std::fabs(dist_m[j] - (dist_m[0] + j * dx)) > 1e-9 * dx) uni_x = false;
    }
    p->dist_sorted = true;
    for (int j = 1; j < tnum; ++j)
        if (!(dist_m[j] >= dist_m[j - 1])) p->dist_sorted = false;
    // position noise of a pair's dist[j] - dist[xi] against n * dx, in units of dx, measured on the profile: twice the
    // largest deviation from the fitted grid plus the rounding of the largest |dist| (never less than 4.5e-16 tnum)
    {
        double dev = 0.0, amax = 0.0;
        for (int j = 0; j < tnum; ++j) {
            dev = std::max(dev, std::fabs(dist_m[j] - (dist_m[0] + j * dx)));
            amax = std::max(amax, std::fabs(dist_m[j]));
        }
        p->xnoise = std::max(4.5e-16 * (double)tnum, (2.0 * dev + 4.5e-16 * amax) / (dx > 0 ? dx : 1.0));
    }
    p->dt = dt;
    p->dx = dx;
    p->tt0 = tt_sec[0];
    p->uniform = uni_t && uni_x;
    const double sa = 2.0 * dx / (vel * dt);       // samples of moveout per trace at far offset
    p->alpha = sa * sa;
    // fast kernels need the moveout 2dx/(v dt) (samples per trace) small enough for their
    // LDS windows: quad (sample-major ring, 24 traces x 8-step blocks, up to ~6.7 samples/trace)
    // or, for steeper moveout, tab (trace-major ring of 16 traces, 512-sample slots)
    // quad kernel's output-trace tile: 24 traces (40-slot ring, three workgroups per CU) or 40 traces
    // (56-slot ring, two workgroups per CU).  The wider tile stages and picks 40 % less per pair and is
    // 2.5-5 % faster on a whole radargram (same-box A/B at config 3; 32 traces: 1.3 %); with the short
    // launches of a many-rank run (under ~2000 output traces per rank) its fewer, longer workgroups
    // balance worse (-7 % at 8 ranks), so those keep 24.  IMPDAR_KIRCH_XB = 24 / 32 / 40 overrides.
    int xbq = 24;
    {
        auto rows_for = [&](int xb) { return ((KF_THREADS + (int)std::ceil(sa * (xb + 8 - 2)) + 8 + 31) / 32) * 32; };
        const char *xe = getenv("IMPDAR_KIRCH_XB");
        if (xe && (atoi(xe) == 24 || atoi(xe) == 32 || atoi(xe) == 40))
            xbq = atoi(xe);
        else if ((size_t)(rows_for(40) / 32) * kq_piece_bytes(40) <= 80 * 1024)
            xbq = 40;       // (round 1 kept 24 for the short launches of a many-rank run; with the balanced tile map and
                            // the work queues 40 is ahead there too: 1.31 vs 1.41 ms for an 8-rank block of config 3)
    }
    // Whole radargrams by default: TWO tiles of 32 traces per workgroup on one ring (kirch_quad_kernel, NH = 2),
    // when that ring fits half a CU's LDS: two workgroups of 8 waves per CU (four waves per SIMD at 128 VGPRs)
    // instead of two of 4.  Same-box A/B at config 3: 2 % faster than one 40-trace tile per workgroup (the same
    // tile pair with one workgroup per CU, 40 x 2, is 1 % slower), and the fabric traffic roughly halves.
    bool pair32 = false;
    // (a rank's block of under ~8000 traces gives the 64-trace workgroup tiles too few items per slot: 40 x 1 there;
    // emulated per-rank steps at config 3, profiles/r02_rank_steps_tiles.txt: 4.28 / 2.32 / 1.31 ms at 2 / 4 / 8 ranks
    // against 4.33 / 2.39 / 1.65 with the tile pair)
    if (!getenv("IMPDAR_KIRCH_XB") && !getenv("IMPDAR_KIRCH_NH") && !nearfield && xbq == 40 &&
        (long long)tnum >= 8000LL * nranks) {
        const int rows = ((KF_THREADS + (int)std::ceil(sa * (32 * 2 + 8 - 2)) + 8 + 31) / 32) * 32;
        if ((size_t)(rows / 32) * kq_piece_bytes(32) <= 80 * 1024 && (size_t)(rows / 32) * kq_piece_bytes(32) > 65535) {
            pair32 = true;
            xbq = 32;
        }
    }
    int kq_ps = kq_piece_bytes(xbq);
    // tiles per workgroup on one ring (see kirch_quad_kernel, NH): the staging window grows by the moveout over
    // the (nh - 1) xb traces the later tiles lag behind; one workgroup per CU then (up to the whole 160 KB)
    int nhq = 1;
    {
        const char *ne = getenv("IMPDAR_KIRCH_NH");         // tuning knob: 1 | 2 | 3
        const int want = ne ? atoi(ne) : (pair32 ? 2 : KQ_DEFAULT_NH);
        auto rows_nh = [&](int nh) { return ((KF_THREADS + (int)std::ceil(sa * (xbq * nh + 8 - 2)) + 8 + 31) / 32) * 32; };
        if ((want == 2 || want == 3) && (xbq == 40 || (xbq == 32 && want == 2)) && !nearfield &&
            (ne || pair32) &&
            (size_t)(rows_nh(want) / 32) * kq_ps <= 160 * 1024 && (size_t)(rows_nh(want) / 32) * kq_ps > 65535)
            nhq = want;
    }
    p->nh = nhq;
    int lkq = 0;
    {
        const char *le = getenv("IMPDAR_KIRCH_LK");         // tuning knob: 0 | 1
        const int want = le ? atoi(le) : KQ_DEFAULT_LK;
        const int rows = ((KF_THREADS + (int)std::ceil(sa * (xbq * nhq + 8 - 2)) + 8 + 31) / 32) * 32;
        // every wave must own at least one DMA piece per block (the in-order wait counts on it)
        if (want == 1 && nhq >= 2 && rows / 32 >= 4 * nhq && (size_t)(rows / 32) * kq_piece_bytes_lk(xbq, 1) <= 160 * 1024)
            lkq = 1;
    }
    p->lk = lkq;
    kq_ps = kq_piece_bytes_lk(xbq, lkq);
    const int wq = ((KF_THREADS + (int)std::ceil(sa * (xbq * nhq + 8 - 2)) + 8 + 31) / 32) * 32;   // whole 32-row pieces
    // (two workgroups per CU: 80 KB of LDS each; table entries are 16-bit byte offsets up to 12 pieces,
    // 16-byte units beyond)
    const bool quad_ok = (size_t)(wq / 32) * kq_ps <= (nhq > 1 ? 160 : 80) * 1024;
    const bool tab_ok = (KF_THREADS + sa * (16 - 1) + 8.0) <= (double)KF_W;
    const bool window_ok = quad_ok || tab_ok;
    // aperture half width in traces (upper bound): below 65536 (the kernel squares trace offsets in 32
    // bits), and the span of image groups one workgroup walks must stay inside its 2 GiB raw buffer
    const double hest = std::min(std::fabs(tmax / dt) / sa + 2.0, (double)tnum + 128.0);
    const bool span_ok = (2.0 * hest + 400.0) / 8.0 * (double)snum * 32.0 < 2147483648.0;
    const bool fast_ok = dtype == IMPDAR_F32 && p->uniform && window_ok && snum < 65536 &&
                         std::fabs(tmax / dt) / sa < 65000.0 && span_ok;

    // float32 data on a profile whose spacing is NOT uniform (mig_python.py:44 takes any dist[]): kirch_gen_kernel
    // computes every pair's pick from the positions.  It needs a uniform time axis, a sorted dist[] (the staging windows
    // and the input range of a tile come from bisections) and, per 32 consecutive output traces, a moveout that fits
    // its LDS slots: W >= 264 + (extent of the 32 traces in samples).  IMPDAR_KIRCH_IMPL=gen takes it on uniform
    // profiles too (A/B against the ring kernels).
    int gen_w = 0;
    bool gen_ok = false;
    bool uni_t11 = uni_t;          // the float64 re-decision of a pick takes tt[k] = tt[0] + k dt: to 1e-11 dt here
    for (int k = 0; k < snum && uni_t11; ++k)
        if (std::fabs(tt_sec[k] - (tt_sec[0] + k * dt)) > 1e-11 * dt) uni_t11 = false;
    if (dtype == IMPDAR_F32 && uni_t11 && p->dist_sorted && tnum >= 2 && snum >= 4 && snum < (1 << 22) && !g_tmax_override &&
        (double)tnum * snum * 4.0 < 2147483648.0) {
        double ext = 0.0;
        for (int j = 0; j < tnum; ++j) ext = std::max(ext, dist_m[std::min(j + 31, tnum - 1)] - dist_m[j]);
        const double need = 264.0 + std::ceil(ext * 2.0 / (vel * dt));
        if (need <= 1024.0) {
            gen_w = ((int)need + 255) / 256 * 256;
            if (gen_w < 512) gen_w = 512;
            gen_ok = true;
        }
    }
    const char *impl_env = getenv("IMPDAR_KIRCH_IMPL");
    const bool gen_forced = gen_ok && impl_env && !strcmp(impl_env, "gen") && mode != IMPDAR_KIRCH_EXACT;
    const int requested_mode = mode;
    const bool gen = gen_forced || (gen_ok && !fast_ok && mode != IMPDAR_KIRCH_EXACT);
    if (mode == IMPDAR_KIRCH_AUTO) mode = (fast_ok || gen) ? IMPDAR_KIRCH_FAST : IMPDAR_KIRCH_EXACT;
    if (mode == IMPDAR_KIRCH_FAST && !fast_ok && !gen) {
        delete p;
        impdar_set_error("the float32 Kirchhoff kernels need float32 data on a uniform travel_time axis and either a "
                         "uniform dist with moveout 2dx/(v dt) <= %.1f samples per trace (got %.2f) or a sorted dist "
                         "whose 32-trace windows span <= 760 samples of moveout",
                         (KF_W - KF_THREADS - 8.0) / 15.0, sa);
        return IMPDAR_ERR_UNSUPPORTED;
    }
    p->mode = mode;
    p->gen = gen && mode == IMPDAR_KIRCH_FAST;
    p->genW = gen_w;
    {
        // One full-aperture walk of a shallow chunk takes ~1.2 ms at config 3 -- as long as the whole step of a rank of
        // an 8-GPU run should be, and such a rank's block has fewer items than the chip has workgroup slots.  Plans
        // of 4+ ranks cut every walk in 2, of 8+ ranks in 4 pieces (kirch_quad_kernel); the pieces are summed in a
        // fixed order, so launches stay bit-reproducible (against whole walks the sum differs by rounding).
        const char *pe = getenv("IMPDAR_KIRCH_PARTS");      // tuning knob: 1 | 2 | 4
        const int parts = pe ? atoi(pe) : (nranks >= 8 ? 4 : (nranks >= 4 ? 2 : 1));
        p->walk_parts_log2 = parts == 4 ? 2 : (parts == 2 ? 1 : 0);
    }
    {
        const char *ie = getenv("IMPDAR_KIRCH_IMPL");       // tuning knob: "tab" forces the b32 ring
        p->quadW = wq;
        p->quadSH = ((size_t)(wq / 32) * kq_ps <= 65535) ? 0 : 4;
        p->quad = (mode == IMPDAR_KIRCH_FAST) && !p->gen && quad_ok && !(ie && !strcmp(ie, "tab") && tab_ok);
        p->xb = p->gen ? 32 : (p->quad ? xbq : 16);
    }
    // float64 data in exact mode on uniform grids: the same ring in float64 (20 or 16 output traces per lane,
    // step blocks of 4) when its window fits; otherwise (and for IMPDAR_KIRCH_EXACT_IMPL = tab | pair) the
    // global-memory kernels
    // The table-driven float64 kernels weight a pair by its trace OFFSET (n dx); the reference by dist[j] - dist[xi].
    // On a profile whose positions are noisy against the grid (a first trace tens of kilometres along the line:
    // ulp(dist) / dx ~ 1e-10; 100000 traces from 0: 4.5e-11) the two weights differ by that much relative, and the
    // result differs from the reference's by up to ~0.03 xnoise of the image maximum (measured 2.5e-12 at 1e-10 with
    // the near-field term).  Rounds 2-3 sent every profile with xnoise > 3e-11 to the per-pair kernel to hold a flat
    // 1e-12 -- 45-100x slower on ordinary long traverses.  The ring (and the tabulated kernel) now stay; the stated bar
    // of the float64 path is  max(1e-12, 0.1 xnoise)  of the image maximum (impdar_kirch_plan_xnoise reports xnoise;
    // picks are not affected: every pick within the noise of a tie is re-done pair by pair, kirch_tiefix_kernel).
    // IMPDAR_KIRCH_EXACT_IMPL=pair still forces the reference's arithmetic pair by pair.
    if (mode == IMPDAR_KIRCH_EXACT && dtype == IMPDAR_F64 && p->uniform && snum < 65536 &&
        std::fabs(tmax / dt) / sa < 65000.0 && (2.0 * hest + 400.0) / 4.0 * (double)snum * 32.0 < 2147483648.0 &&
        !getenv("IMPDAR_KIRCH_EXACT_IMPL")) {
        auto rows_for = [&](int xb, int nh) { return ((KF_THREADS + (int)std::ceil(sa * (xb * nh + 4 - 2)) + 8 + 31) / 32) * 32; };
        auto fits = [&](int xb, int nh) {
            const size_t b = (size_t)(rows_for(xb, nh) / 32) * kd_piece_bytes(xb);
            return b <= 80 * 1024 && (nh == 1 || b > 65535);
        };
        // Whole radargrams: TWO tiles of 20 traces per workgroup on one ring when that ring fits half a CU's LDS, as the
        // float32 kernel does (round 4, same-box A/B at config 3: 17.71 -> 16.91 ms; two tiles of 16: 19.2; a rank's
        // block of under ~8000 traces keeps one tile, as there)
        const char *ne = getenv("IMPDAR_KIRCH_NHD");        // tuning knob: tiles per workgroup, 1 | 2
        const char *xe = getenv("IMPDAR_KIRCH_XBD");        // tuning knob: 16 | 20
        int nhd = ne ? atoi(ne) : ((!xe && fits(20, 2) && (long long)tnum >= 8000LL * nranks) ? 2 : KD_DEFAULT_NH);
        if (nhd != 2 || nearfield) nhd = 1;
        int xbd = (xe && atoi(xe) == 16) ? 16 : ((xe && atoi(xe) == 20) ? 20 : ((nhd == 2 && ne) ? KD_DEFAULT_XB2 : 20));
        if (nhd == 2 && !fits(xbd, 2)) nhd = 1;
        if (!fits(xbd, nhd)) xbd = 16;
        if (fits(xbd, nhd)) {
            p->dquad = true;
            p->xb = xbd;
            p->nh = nhd;
            p->quadW = rows_for(xbd, nhd);
            p->quadSH = ((size_t)(p->quadW / 32) * kd_piece_bytes(xbd) <= 65535) ? 0 : 4;
        }
    }

    int rc = IMPDAR_OK;
    auto fail = [&](int code) {
        delete p;
        return code;
    };
    if (hipSetDevice(ctx->device) != hipSuccess) {
        impdar_set_error("hipSetDevice(%d) failed: %s", ctx->device, hipGetErrorString(hipGetLastError()));
        return fail(IMPDAR_ERR_HIP);
    }
    const size_t esz = impdar_dtype_size(dtype);
    const size_t img = (size_t)(p->tnum_pad + 2 * KF_PAD_ROWS) * snum * esz;   // zero rows on both sides
    for (int b = 0; b < 2; ++b) {
        if (p->GT[b].ensure(img) != hipSuccess || (p->nearfield && p->DT[b].ensure(img) != hipSuccess)) {
            impdar_set_error("hipMalloc of %zu-byte image failed", img);
            return fail(IMPDAR_ERR_HIP);
        }
        (void)hipMemsetAsync(p->GT[b].p, 0, img, ctx->stream);
        if (p->nearfield) (void)hipMemsetAsync(p->DT[b].p, 0, img, ctx->stream);
    }
    (void)hipStreamSynchronize(ctx->stream);
    if (!grad_uniform) {
        if ((rc = upload(p->d_ga, ga, snum * 8)) || (rc = upload(p->d_gb, gb, snum * 8)) ||
            (rc = upload(p->d_gc, gc, snum * 8)))
            return fail(rc);
    }
    // exact-kernel tables (also used by count_pairs)
    {
        std::vector<double> zs(snum), zs2(snum);
        for (int k = 0; k < snum; ++k) {
            zs[k] = vel * tt_sec[k] / 2.0;          // mig_python.py:101
            zs2[k] = zs[k] * zs[k];                 // :102
        }
        std::vector<double> dpad(dist_m, dist_m + tnum);
        dpad.resize((size_t)tnum + 64, dist_m[tnum - 1]);      // kirch_gen_kernel reads up to 31 entries past a tile's end
        if ((rc = upload(p->d_dist, dpad.data(), dpad.size() * 8)) || (rc = upload(p->d_tt, tt_sec, (size_t)snum * 8)) ||
            (rc = upload(p->d_zs, zs.data(), (size_t)snum * 8)) || (rc = upload(p->d_zs2, zs2.data(), (size_t)snum * 8)))
            return fail(rc);
    }
    // aperture half width per sample (uniform grids): largest n with t <= tmax
    if (p->uniform) {
        p->h_half.resize(snum);
        const double um = tmax / dt;
        for (int k = 0; k < snum; ++k) {
            const double a = tt_sec[k] / dt;
            const double rem = um * um - a * a;
            p->h_half[k] = rem < 0 ? -1 : (int)std::floor(std::sqrt(rem / p->alpha) + 1e-12);
        }
    }
    // ---- picks that rounding noise decides (see kirch_tiescan_kernel): the table-driven kernels would break those
    // ties one way per offset, the reference breaks them pair by pair
    // (kirch_gen_kernel needs no list, on a uniform profile either: it re-does every pair on a half-way point in the
    // reference's own arithmetic by itself, kg_ref_upper; rounds 3-4 sent uniform profiles with ties to the float64 kernels)
    if (p->uniform && !p->gen && (mode == IMPDAR_KIRCH_FAST || p->dquad || !getenv("IMPDAR_KIRCH_EXACT_IMPL") ||
                                  strcmp(getenv("IMPDAR_KIRCH_EXACT_IMPL"), "pair"))) {
        int hg = 0;
        for (int k = 0; k < snum; ++k) hg = std::max(hg, p->h_half[k] + 1);
        hg = std::min(hg, tnum) + 1;
        constexpr int TIE_CAP = 1 << 20;
        DevBuf d_flag, d_list;
        if (d_flag.ensure(64) != hipSuccess || d_list.ensure((size_t)TIE_CAP * sizeof(int2)) != hipSuccess) {
            impdar_set_error("hipMalloc failed");
            return fail(IMPDAR_ERR_HIP);
        }
        (void)hipMemsetAsync(d_flag.p, 0, 64, ctx->stream);
        TableXParams T;
        T.XK = nullptr;
        T.XW = T.XW2 = nullptr;
        T.zs = p->d_zs.as<double>();
        T.zs2 = p->d_zs2.as<double>();
        T.tt = p->d_tt.as<double>();
        T.dx = dx;
        T.vel = vel;
        T.tmax = tmax;
        T.inv_dt = 1.0 / dt;
        T.tt0 = tt_sec[0];
        T.snum = snum;
        T.ntab = hg;
        T.near = 0;
        const double xnoise = p->xnoise;     // position noise of a pair's dist[j] - dist[xi] in units of dx, from the profile itself
        hipLaunchKernelGGL(kirch_tiescan_kernel, dim3((snum + 255) / 256, hg), dim3(256), 0, ctx->stream, T, xnoise,
                           d_flag.as<int>(), d_list.as<int2>(), TIE_CAP);
        int count = 0;
        if (hipMemcpyAsync(&count, d_flag.p, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
            impdar_set_error("tie scan failed: %s", hipGetErrorString(hipGetLastError()));
            return fail(IMPDAR_ERR_HIP);
        }
        const bool no_fix = getenv("IMPDAR_KIRCH_TIEFIX") && !strcmp(getenv("IMPDAR_KIRCH_TIEFIX"), "0");   // diagnostic
        if (count > 0 && count <= TIE_CAP && !no_fix) {
            // group the flagged offsets by sample (sorted: the correction adds them in a fixed order)
            std::vector<int2> list(count);
            if (hipMemcpy(list.data(), d_list.p, (size_t)count * sizeof(int2), hipMemcpyDeviceToHost) != hipSuccess) {
                impdar_set_error("tie list download failed");
                return fail(IMPDAR_ERR_HIP);
            }
            std::sort(list.begin(), list.end(), [](const int2 &a, const int2 &b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
            std::vector<int> g_ti, g_off, g_n(count);
            for (int i = 0; i < count; ++i) {
                if (i == 0 || list[i].x != list[i - 1].x) {
                    g_ti.push_back(list[i].x);
                    g_off.push_back(i);
                }
                g_n[i] = list[i].y;
            }
            g_off.push_back(count);
            p->ntie_groups = (int)g_ti.size();
            if ((rc = upload(p->d_tie_ti, g_ti.data(), g_ti.size() * 4)) || (rc = upload(p->d_tie_off, g_off.data(), g_off.size() * 4)) ||
                (rc = upload(p->d_tie_n, g_n.data(), g_n.size() * 4)))
                return fail(rc);
        }
        p->tie_ambiguous = count > TIE_CAP || (count > 0 && no_fix);
        if (p->tie_ambiguous) {
            // more ties than the list holds (or the correction switched off): the float32 ring kernels stay available
            // when asked for by name; everything the library chooses itself goes per pair
            if (mode == IMPDAR_KIRCH_FAST && requested_mode == IMPDAR_KIRCH_AUTO) {
                mode = p->mode = IMPDAR_KIRCH_EXACT;
                p->quad = false;
                p->xb = 16;
            }
            p->dquad = false;
            p->xtab_off = true;
        }
    }
    if (p->gen) {
        // per-sample float32 factors, the squared half-way radii of the float64 re-decision, per-chunk bounds
        const int nch = (snum + KF_THREADS - 1) / KF_THREADS;
        p->nchunks = nch;
        std::vector<float> a(snum), a2(snum), alo2(nch, 3.0e38f);
        p->h_zs2min.assign(nch, 1e300);
        for (int k = 0; k < snum; ++k) {
            const double ak = tt_sec[k] / dt;
            a[k] = (float)ak;
            a2[k] = (float)(ak * ak / ((double)(gen_w - 1) * (double)(gen_w - 1)));    // normalised to the slot (kirch_gen_kernel)
            alo2[k / KF_THREADS] = std::min(alo2[k / KF_THREADS], (float)(ak * ak) * (1.0f - 1.0e-6f));
            const double zs = vel * tt_sec[k] / 2.0;
            p->h_zs2min[k / KF_THREADS] = std::min(p->h_zs2min[k / KF_THREADS], zs * zs);
        }
        p->h_dist.assign(dist_m, dist_m + tnum);
        if ((rc = upload(p->d_ga32, a.data(), snum * 4)) || (rc = upload(p->d_ga2_32, a2.data(), snum * 4)) ||
            (rc = upload(p->d_alo2, alo2.data(), nch * 4)))
            return fail(rc);
    }
    const int ringS = p->dquad ? 4 : 8;            // steps per block of the ring kernels (traces per 32-byte row)
    if ((mode == IMPDAR_KIRCH_FAST && !p->gen) || p->dquad) {
        const int nch = (snum + KF_THREADS - 1) / KF_THREADS;
        p->nchunks = nch;
        std::vector<int> hmax(nch, 0);
        if (p->dquad) {
            // float64 per-sample factors, from the reference's own zs = v t / 2 (mig_python.py:101):
            //   cos(theta) = zs / sqrt(zs^2 + (n dx)^2) = sign(zs) * rsqrt(1 + c1 n^2),   c1 = (dx / zs)^2
            //   far field  : cos / (2 pi v)     = fin * |cos|,           fin = sign(zs) / (2 pi v)
            //   near field : cos / (2 pi rs^2)  = fin * c2 * |cos|^3,    c2 = v / zs^2
            // zs = 0: the whole output row is 0 (cos = 0 off the apex, the apex is 0/0 and dropped): fin = 0
            std::vector<double> c1(snum), c2(snum), fin(snum);
            for (int k = 0; k < snum; ++k) {
                const double zs = vel * tt_sec[k] / 2.0;
                if (zs == 0.0) {
                    c1[k] = c2[k] = fin[k] = 0.0;
                    continue;
                }
                c1[k] = std::min((dx / zs) * (dx / zs), 1e300);
                c2[k] = std::min(vel / (zs * zs), 1e300);
                fin[k] = (zs > 0 ? 1.0 : -1.0) / (2.0 * M_PI * vel);
            }
            if ((rc = upload(p->d_c1d, c1.data(), snum * 8)) || (rc = upload(p->d_c2d, c2.data(), snum * 8)) ||
                (rc = upload(p->d_find, fin.data(), snum * 8)))
                return fail(rc);
        } else {
            // with a = tt/dt (samples) and rs = half * sqrt(a^2 + alpha n^2):
            //   cos(theta)        = a / sqrt(a^2 + alpha n^2) = sign(a) * rsq(1 + c1 n^2),  c1 = alpha / a^2
            //   far-field weight  = cos / (2 pi v)            = fin * |cos|,                 fin = sign(a) / (2 pi v)
            //   near-field weight = cos / (2 pi rs^2)         = fin * c2 * |cos|^3,          c2 = v / (half a)^2
            // a = 0 (a sample at t = 0): cos = 0 for every n != 0 and the apex is 0/0 (dropped) -> fin = 0
            std::vector<float> c1(snum), c2(snum), fin(snum);
            const double half = vel * dt / 2.0;         // metres per sample of two-way time
            for (int k = 0; k < snum; ++k) {
                const double a = tt_sec[k] / dt;
                if (a == 0.0) {
                    c1[k] = c2[k] = fin[k] = 0.f;
                    continue;
                }
                c1[k] = (float)std::min(p->alpha / (a * a), 1e30);
                c2[k] = (float)std::min(vel / (half * half * a * a), 1e30);
                fin[k] = (float)((a > 0 ? 1.0 : -1.0) / (2.0 * M_PI * vel));
            }
            if ((rc = upload(p->d_c1, c1.data(), snum * 4)) || (rc = upload(p->d_c2, c2.data(), snum * 4)) ||
                (rc = upload(p->d_fin, fin.data(), snum * 4)))
                return fail(rc);
        }
        int hglob = 0;
        std::vector<double> cmin(nch), cmax(nch);
        for (int c = 0; c < nch; ++c) {
            double amin = 1e300, amax = 0;
            int h = 0;
            for (int k = c * KF_THREADS; k < std::min(snum, (c + 1) * KF_THREADS); ++k) {
                const double a = tt_sec[k] / dt;
                amin = std::min(amin, a * a);
                amax = std::max(amax, a * a);
                h = std::max(h, p->h_half[k] + 1);
            }
            hmax[c] = h;
            hglob = std::max(hglob, h);
            cmin[c] = amin;
            cmax[c] = amax;
        }
        // offsets beyond the profile length can only meet traces outside the profile (zero
        // rows), so the tables need not extend past tnum even when the aperture does
        hglob = std::min(hglob, tnum + 128);
        for (int c = 0; c < nch; ++c) hmax[c] = std::min(hmax[c], hglob);
        const int nb = hglob + 64;
        p->nb = nb;
        p->ntab = hglob + 1;       // offsets 0..hglob-1 (hmax carries a guard) + one all-zero row
        // ring kernels: tables by step block, row r <-> offsets n = S (r - mrow0) + 1 .. + S
        p->mrow0 = hglob / ringS + 8;
        p->nrows = 2 * (hglob / ringS) + 64;
        {
            const size_t tkbytes = (p->quad || p->dquad) ? (size_t)p->nrows * kq_tkb_stride(snum) * 2 * ringS
                                                         : (size_t)p->ntab * snum * 2;
            if (tkbytes >= ((size_t)1 << 31)) {      // the kernels address it as one raw buffer
                impdar_set_error("fast Kirchhoff pick table of %zu bytes exceeds 2 GiB; use the exact mode", tkbytes);
                return fail(IMPDAR_ERR_UNSUPPORTED);
            }
            const size_t ent = (size_t)p->ntab * snum;
            bool ok = true;
            for (int b = 0; b < 2; ++b) {
                ok = ok && p->d_TK[b].ensure(tkbytes) == hipSuccess;
                if (!p->quad && !p->dquad) ok = ok && p->d_TW[b].ensure(ent * 4) == hipSuccess;
                if (!p->quad && !p->dquad && p->nearfield) ok = ok && p->d_TW2[b].ensure(ent * 4) == hipSuccess;
            }
            if (!ok) {
                impdar_set_error("hipMalloc of the %zu-byte pick table failed", tkbytes);
                return fail(IMPDAR_ERR_HIP);
            }
        }
        // staging windows: smallest / largest sample index any lane of chunk c
        // can pick at offset |n| (one guard sample each side)
        std::vector<int> klo((size_t)nch * nb), khi((size_t)nch * nb);
        const double u0 = tt_sec[0] / dt;
        for (int c = 0; c < nch; ++c)
            for (int n = 0; n < nb; ++n) {
                const double bn = p->alpha * (double)n * (double)n;
                const double ulo = std::sqrt(cmin[c] + bn) - u0, uhi = std::sqrt(cmax[c] + bn) - u0;
                klo[(size_t)c * nb + n] = std::max(0, (int)std::floor(ulo) - 1);
                khi[(size_t)c * nb + n] = std::min(snum - 1, (int)std::ceil(uhi) + 1);
            }
        if (p->quad || p->dquad) {
            // the S traces block r adds are read by the steps n = S (r - mrow0) + 1 .. + XB + S - 2 (they enter
            // the XB-trace window of a lane at its last slot and leave it XB - 1 steps later)
            std::vector<int> win((size_t)nch * p->nrows * 2);
            for (int r = 0; r < p->nrows; ++r) {
                // (with nh tiles on one ring the later tiles read the same traces (nh - 1) xb offsets earlier)
                const long long nz = (long long)ringS * (r - p->mrow0) + 1 + p->xb + ringS - 2;
                const long long na = (long long)ringS * (r - p->mrow0) + 1 - (long long)((p->quad || p->dquad) ? p->nh - 1 : 0) * p->xb;
                const long long lo = (na <= 0 && nz >= 0) ? 0 : std::min(std::llabs(na), std::llabs(nz));
                const long long hi = std::max(std::llabs(na), std::llabs(nz));
                for (int c = 0; c < nch; ++c) {
                    const int kmin = klo[(size_t)c * nb + std::min<long long>(lo, nb - 1)];
                    const int kmax = khi[(size_t)c * nb + std::min<long long>(hi, nb - 1)];
                    win[((size_t)c * p->nrows + r) * 2 + 0] = kmin | ((kmin % p->quadW) << 16);
                    win[((size_t)c * p->nrows + r) * 2 + 1] = kmax;
                }
            }
            if ((rc = upload(p->d_WIN, win.data(), win.size() * 4))) return fail(rc);
        }
        p->h_hmax = hmax;
        if ((rc = upload(p->d_hmax, hmax.data(), nch * 4)) || (rc = upload(p->d_klo, klo.data(), klo.size() * 4)) ||
            (rc = upload(p->d_khi, khi.data(), khi.size() * 4)))
            return fail(rc);
    }
    for (int s = 0; s < impdar_kirch_plan::NSLOT; ++s)
        for (int i = 0; i < 6; ++i)
            if (hipEventCreate(&p->evs[s][i]) != hipSuccess) {
                impdar_set_error("hipEventCreate failed");
                return fail(IMPDAR_ERR_HIP);
            }
    if (hipEventCreateWithFlags(&p->ev_ready[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p->ev_ready[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p->ev_free[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p->ev_free[1], hipEventDisableTiming) != hipSuccess) {
        impdar_set_error("hipEventCreate failed");
        return fail(IMPDAR_ERR_HIP);
    }
    *out = p;
    return IMPDAR_OK;
}

extern "C" void impdar_kirch_plan_destroy(impdar_kirch_plan *p)
{
    if (!p) return;
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->aux);
    (void)hipStreamSynchronize(p->ctx->stream);
    delete p;
}

extern "C" int impdar_kirch_plan_mode(const impdar_kirch_plan *p) { return p ? p->mode : IMPDAR_ERR_ARG; }
extern "C" double impdar_kirch_plan_xnoise(const impdar_kirch_plan *p) { return p ? p->xnoise : -1.0; }
extern "C" int impdar_kirch_plan_tnum_pad(const impdar_kirch_plan *p) { return p ? p->tnum_pad : IMPDAR_ERR_ARG; }

extern "C" int impdar_kirch_plan_kernel(const impdar_kirch_plan *p)
{
    if (!p) return IMPDAR_ERR_ARG;
    if (p->gen) return IMPDAR_KERNEL_GEN;
    if (p->mode == IMPDAR_KIRCH_FAST) return p->quad ? IMPDAR_KERNEL_QUAD : IMPDAR_KERNEL_TAB;
    if (p->dquad) return IMPDAR_KERNEL_DQUAD;
    const char *e = getenv("IMPDAR_KIRCH_EXACT_IMPL");
    if (p->uniform && !p->xtab_off && !(e && !strcmp(e, "pair"))) return IMPDAR_KERNEL_EXACT_TAB;
    return IMPDAR_KERNEL_EXACT_PAIR;
}

// `more`: further input traces of the radargram whose first traces an earlier prep took, AFTER a migrate that reads
// only those has been enqueued (the pipelined one-shot call): same buffer set, and no wait for that migrate -- it does
// not read the rows written here.
static int kirch_prep_impl(impdar_kirch_plan *p, const void *d_data, int ld, int jlo, int nloc, int precomputed, bool more = false)
{
    IMPDAR_ARG_CHECK(p && d_data, "null plan/data");
    IMPDAR_ARG_CHECK(jlo >= 0 && nloc >= 0 && jlo + nloc <= p->tnum_pad && ld >= nloc,
                     "column block [%d,%d) with ld %d does not fit the plan (tnum_pad %d)", jlo, jlo + nloc, ld,
                     p->tnum_pad);
    IMPDAR_HIP_CHECK(hipSetDevice(p->ctx->device));
    hipStream_t st = p->ctx->aux;
    if (more) {
        p->migrated_since_prep = false;
    } else if (p->migrated_since_prep) {
        // first prep after a migrate: a new radargram -> other buffer set, next event slot
        p->migrated_since_prep = false;
        p->buf ^= 1;
        p->slot = (p->slot + 1) % impdar_kirch_plan::NSLOT;
        p->haves[p->slot][0] = p->haves[p->slot][1] = p->haves[p->slot][2] = false;
    }
    const int b = p->buf;
    // the buffer set may still be read by the migrate of the radargram before last.  (The
    // input itself must already be complete: the upload entry points are blocking, and the
    // one-shot paths below synchronise their own copy before calling prep.)
    if (p->free_recorded[b] && !more) IMPDAR_HIP_CHECK(hipStreamWaitEvent(st, p->ev_free[b], 0));
    // d_data may have been produced by a *_dev step (band pass, re-spacing, cast, another migration).  Those only
    // enqueue on the compute stream and mark the context (impdar_ctx_mark_produced): the producer stream waits
    // for the last such mark.  The mark sits in FRONT of any diffraction sum enqueued since, so the prep of
    // radargram s+1 still overlaps the migrate of radargram s; only when the input IS the previous migrate's
    // output does prep wait for that kernel.
    if (p->ctx->produced) IMPDAR_HIP_CHECK(hipStreamWaitEvent(st, p->ctx->ev_produced, 0));
    if (p->last_out && d_data == p->last_out && p->free_recorded[p->last_out_buf])
        IMPDAR_HIP_CHECK(hipStreamWaitEvent(st, p->ev_free[p->last_out_buf], 0));
    hipEvent_t *ev = p->evs[p->slot];
    if (!p->haves[p->slot][0]) IMPDAR_HIP_CHECK(hipEventRecord(ev[0], st));
    if (nloc > 0) {
        PrepParams P;
        P.data = d_data;
        P.ld = ld;
        P.snum = p->snum;
        P.nloc = nloc;
        P.jlo = jlo;
        P.GT = img_row0(p, p->GT[b]);
        P.DT = p->nearfield ? img_row0(p, p->DT[b]) : nullptr;
        P.grad_uniform = p->grad_uniform;
        P.precomputed = precomputed;
        P.grad_h = p->grad_h;
        P.ga = p->d_ga.as<double>();
        P.gb = p->d_gb.as<double>();
        P.gc = p->d_gc.as<double>();
        P.clean = (p->mode == IMPDAR_KIRCH_FAST) ? 1 : (p->dquad ? 2 : 0);
        P.i8 = (p->quad || p->dquad) ? 1 : 0;
        dim3 grid((nloc + 63) / 64, (p->snum + 63) / 64);
        if (p->dtype == IMPDAR_F32 && P.i8)
            hipLaunchKernelGGL((kirch_prep_direct_kernel<float, 8>), dim3((nloc + 255) / 256, p->snum), dim3(256), 0, st, P);
        else if (p->dquad)
            hipLaunchKernelGGL((kirch_prep_direct_kernel<double, 4>), dim3((nloc + 255) / 256, p->snum), dim3(256), 0, st, P);
        else if (p->dtype == IMPDAR_F32)
            hipLaunchKernelGGL((kirch_prep_kernel<float, float>), grid, dim3(256), 0, st, P);
        else
            hipLaunchKernelGGL((kirch_prep_kernel<double, double>), grid, dim3(256), 0, st, P);
        IMPDAR_HIP_CHECK(hipGetLastError());
    }
    // The pick table depends on the plan's geometry only (like an FFT plan's twiddles): it is built by the first prep
    // into each of the two buffer sets and kept.  (Round 1 rebuilt it with every prep -- 0.06-0.13 ms on the producer
    // stream, hidden behind a whole-radargram diffraction sum but a tenth of the step of an 8-rank block.)
    if (((p->mode == IMPDAR_KIRCH_FAST && p->quad) || p->dquad) && p->table_built[b]) {
        // this buffer set's table is in place
    } else if ((p->mode == IMPDAR_KIRCH_FAST && p->quad) || p->dquad) {
        p->table_built[b] = true;
        ++p->diag_tables_built;
        TableQParams T;
        T.TKB = p->d_TK[b].as<uint4>();
        T.zs = p->d_zs.as<double>();
        T.zs2 = p->d_zs2.as<double>();
        T.tt = p->d_tt.as<double>();
        T.dx = p->dx;
        T.vel = p->vel;
        T.tmax = p->tmax;
        T.inv_dt = 1.0 / p->dt;
        T.tt0 = p->tt0;
        T.snum = p->snum;
        T.nrows = p->nrows;
        T.mrow0 = p->mrow0;
        T.nmax = p->ntab - 1;
        T.wmod = p->quadW;
        T.sh = p->quadSH;
        T.ps = p->dquad ? kd_piece_bytes(p->xb) : kq_piece_bytes_lk(p->xb, p->lk);
        // rows a + mrow0 (a >= 0) and mrow0 - a - 1: a runs over the larger of the two sides
        const int na = std::max(p->nrows - p->mrow0, p->mrow0);
        if (p->dquad)
            hipLaunchKernelGGL(kirch_tabled_kernel, dim3((p->snum + 255) / 256, na), dim3(256), 0, st, T);
        else
            hipLaunchKernelGGL(kirch_tableq_kernel, dim3((p->snum + 255) / 256, na), dim3(256), 0, st, T);
        IMPDAR_HIP_CHECK(hipGetLastError());
    } else if (p->mode == IMPDAR_KIRCH_FAST && !p->gen) {
        // geometry-only pick/weight table, rebuilt with every prep (counted in prep time)
        TableParams T;
        T.TK = p->d_TK[b].as<unsigned short>();
        T.TW = p->d_TW[b].as<float>();
        T.TW2 = p->d_TW2[b].as<float>();
        T.zs = p->d_zs.as<double>();
        T.zs2 = p->d_zs2.as<double>();
        T.tt = p->d_tt.as<double>();
        T.dx = p->dx;
        T.vel = p->vel;
        T.tmax = p->tmax;
        T.inv_dt = 1.0 / p->dt;
        T.tt0 = p->tt0;
        T.snum = p->snum;
        T.ntab = p->ntab;
        T.near = p->nearfield;
        T.wmod = KF_W;
        T.kscale = 4;
        T.sentinel = 0;
        T.write_w = 1;
        hipLaunchKernelGGL(kirch_table_kernel, dim3((p->snum + 255) / 256, p->ntab), dim3(256), 0, st, T);
        IMPDAR_HIP_CHECK(hipGetLastError());
    }
    IMPDAR_HIP_CHECK(hipEventRecord(ev[1], st));
    IMPDAR_HIP_CHECK(hipEventRecord(p->ev_ready[b], st));
    p->haves[p->slot][0] = true;
    return IMPDAR_OK;
}

extern "C" int impdar_kirch_prep(impdar_kirch_plan *p, const void *d_data, int ld, int jlo, int nloc)
{
    return kirch_prep_impl(p, d_data, ld, jlo, nloc, 0);
}

int impdar_kirch_prep_precomputed(impdar_kirch_plan *p, const void *d_grad, int ld, int jlo, int nloc)
{
    return kirch_prep_impl(p, d_grad, ld, jlo, nloc, 1);
}

// Which output tile a workgroup takes: blocks are dealt round-robin over the 8 XCDs (block b -> XCD b & 7), so slot
// q of chunk c on XCD x is block ((c * tiles_per_xcd + q) * 8 + x).  The arithmetic rule (groups of G adjacent tiles
// per XCD in turn) leaves the XCDs up to +-3 % apart in work at config 3: tiles near the ends of the profile walk
// clipped apertures, and which XCD gets them depends on the tile count.  Here the groups of G adjacent tiles (they
// share staging lines in the XCD's L2) are handed out per chunk, longest first, each to the XCD with the least
// accumulated walk so far (steps rounded up to ring revolutions, plus a prologue's worth).  -1 = empty slot.
static int build_tilemap(impdar_kirch_plan *p, FastParams &P, int tile_w, int align_mask, int ring_blocks, int step_block,
                         hipStream_t st)
{
    P.tilemap = nullptr;
    if ((int)p->h_hmax.size() != P.nchunks) return IMPDAR_OK;
    const int key[5] = {P.xlo, P.xhi, tile_w, P.G, P.tiles_per_xcd};
    const size_t n = (size_t)P.nchunks * P.tiles_per_xcd * 8;
    if (memcmp(key, p->tm_key, sizeof(key)) != 0 || p->h_tilemap.size() != n) {
        std::vector<short> map(n, (short)-1);
        const int x00 = P.xlo & ~align_mask;
        const int G = P.G, units = (P.nxt + G - 1) / G, cap = P.tiles_per_xcd / G;
        std::vector<double> load(8, 0.0);
        std::vector<std::pair<double, int>> cost(units);
        for (int c = 0; c < P.nchunks; ++c) {
            const int hm = p->h_hmax[c];
            for (int u = 0; u < units; ++u) {
                double w = 0;
                for (int t = u * G; t < std::min((u + 1) * G, P.nxt); ++t) {
                    const int x0 = x00 + t * tile_w;
                    const int nlo = std::max(-hm, -(x0 + tile_w - 1)), nhi = std::min(hm, p->tnum - 1 - x0);
                    const int blocks = std::max(0, nhi - nlo + step_block) / step_block;
                    w += ((blocks + ring_blocks - 1) / ring_blocks) * ring_blocks + 8;
                }
                cost[u] = {w, u};
            }
            std::sort(cost.begin(), cost.end(), [](const std::pair<double, int> &a, const std::pair<double, int> &b) {
                return a.first > b.first || (a.first == b.first && a.second < b.second);
            });
            int used[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (const auto &cu : cost) {
                int best = -1;
                for (int x = 0; x < 8; ++x)
                    if (used[x] < cap && (best < 0 || load[x] < load[best])) best = x;
                if (best < 0) return IMPDAR_OK;          // cannot happen (8 cap >= units); keep the arithmetic rule
                for (int g = 0; g < G; ++g) {
                    const int t = cu.second * G + g;
                    map[((size_t)c * P.tiles_per_xcd + (size_t)used[best] * G + g) * 8 + best] = (short)(t < P.nxt ? t : -1);
                }
                ++used[best];
                load[best] += cu.first;
            }
        }
        if (P.nxt > 32767) return IMPDAR_OK;
        p->h_tilemap.swap(map);
        memcpy(p->tm_key, key, sizeof(key));
        IMPDAR_HIP_CHECK(p->d_tilemap.ensure(n * sizeof(short)));
        IMPDAR_HIP_CHECK(hipMemcpyAsync(p->d_tilemap.p, p->h_tilemap.data(), n * sizeof(short), hipMemcpyHostToDevice, st));
    }
    P.tilemap = p->d_tilemap.as<short>();
    return IMPDAR_OK;
}

// Workgroup slots the persistent ring kernels leave EMPTY in a multi-rank plan.  They launch exactly as many
// workgroups as are resident at once and keep them until the diffraction sum ends; RCCL's send/recv and all-gather
// are kernels too (ncclDevKernel_Generic: workgroups of their own, with LDS).  Measured on one GPU with a rank of an
// 8-rank plan emulated and its exchange done as a self send/recv (profiles/r03_exchange_overlap.txt, kernel trace
// profiles/r03_exchange_trace_*.txt): queued on the producer stream behind a full chip the RCCL kernel is dispatched
// 0.1 ms into the diffraction sum but ENDS with it (4.5 of 4.7 ms; 1.06 of 1.08 ms) -- the exchange of radargram
// s+1 serialises behind the sum of radargram s instead of hiding under it.  With 32 of the 512 slots left free it
// completes in 0.15-0.3 ms while the sum runs (8 and 16 free slots do not change anything); the diffraction sum pays
// 2.7-2.9 % (4.665 -> 4.800 ms, 1.081 -> 1.110 ms), 5.7 % with 64.  IMPDAR_KIRCH_RESERVE=<R> overrides (0: none).
#ifndef KQ_DEFAULT_RESERVE
#define KQ_DEFAULT_RESERVE 32
#endif
static int kirch_reserved_slots(const impdar_kirch_plan *p)
{
    const char *re = getenv("IMPDAR_KIRCH_RESERVE");
    const int r = re ? atoi(re) : (p->nranks > 1 ? KQ_DEFAULT_RESERVE : 0);
    return std::min(std::max(r, 0), 256);
}

template <int XB, int OCC, int SH, int NH = 1, int LK = 0>
static int launch_quad(impdar_kirch_plan *p, const FastParams &P0, int nx, hipStream_t st)
{
    FastParams P = P0;
    const int ntiles = (P.xhi - (P.xlo & ~7) + XB * NH - 1) / (XB * NH);      // workgroup tiles start at a multiple of 8
    P.nxt = ntiles;
    {
        // groups of 4 adjacent tiles per XCD share staging lines in L2 (same-box A/B at config 3: 1.2-1.6 %
        // faster than 1, L2 misses -36 %; 6 / 8 / 12 are slower), but the XCDs only stay balanced when each gets many
        // groups: with the 45-70 tiles of an 8-rank block groups of 4 leave half of the XCDs with twice the work (-16 %).
        P.G = ntiles >= 256 ? 4 : 1;
    }
    const int per = 8 * P.G;
    const int nxt_pad = ((ntiles + per - 1) / per) * per;
    P.tiles_per_xcd = nxt_pad / 8;
    const int nblk = P.nchunks * nxt_pad;
    const int W = p->quadW;
    {
        const int trc = build_tilemap(p, P, XB * NH, 7, (kq_ring_slots(XB) + 8 * LK) / 8, 8, st);
        if (trc) return trc;
    }
    const size_t shmem = (size_t)(W / 32) * kq_piece_bytes_lk(XB, LK) * (p->nearfield ? 2 : 1) + 16;   // + the item slot
    if (p->nearfield) {
        auto k = kirch_quad_kernel<XB, true, 1, SH, 1, 0>;
        IMPDAR_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(k, dim3(nblk), dim3(KF_THREADS), shmem, st, P, W);
    } else {
        auto k = kirch_quad_kernel<XB, false, OCC, SH, NH, LK>;
        IMPDAR_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        int grid = nblk;
        if (P.tilemap) {
            // persistent workgroups: as many as are resident at once, each pulling items from the per-XCD queues
            if (p->slots <= 0) {        // resident workgroups of this plan's kernel on this device: asked once
                int per_cu = 0, ncu = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k, KF_THREADS * NH, shmem) == hipSuccess &&
                    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, p->ctx->device) == hipSuccess &&
                    per_cu > 0 && ncu > 0)
                    p->slots = per_cu * ncu;
                else
                    (void)hipGetLastError();
            }
            // many-rank plans: every walk in 2 or 4 pieces with partial images of their own (p->walk_parts_log2)
            const int pl2 = NH == 1 ? p->walk_parts_log2 : 0;
            const size_t nout = (size_t)P.snum * P.ldo, esz = impdar_dtype_size(p->dtype);
            const bool pieces = pl2 > 0 && p->slots > 0 && p->d_partial.ensure((nout << pl2) * esz) == hipSuccess;
            if (p->slots > 0 && (p->slots < nblk || pieces) && p->d_queue.ensure(8 * 64) == hipSuccess) {
                IMPDAR_HIP_CHECK(hipMemsetAsync(p->d_queue.p, 0, 8 * 64, st));
                P.queue = p->d_queue.as<int>();
                if (pieces) {
                    P.parts_log2 = pl2;
                    P.partial = p->d_partial.p;
                    P.part_stride = nout;
                    // a piece no workgroup reaches (tiles beyond the block's end) must read as zero
                    IMPDAR_HIP_CHECK(hipMemsetAsync(p->d_partial.p, 0, (nout << pl2) * esz, st));
                }
                grid = (int)std::min<long long>(std::max(p->slots - kirch_reserved_slots(p), 1), (long long)nblk << P.parts_log2);
            }
        }
        hipLaunchKernelGGL(k, dim3(grid), dim3(KF_THREADS * NH), shmem, st, P, W);
        if (P.parts_log2) {
            const size_t nout = (size_t)P.snum * P.ldo;
            const dim3 cg((unsigned)((nout + 255) / 256));
            if (p->dtype == IMPDAR_F32) {
                if (P.parts_log2 == 2)
                    hipLaunchKernelGGL((kirch_combine_kernel<float, 4>), cg, dim3(256), 0, st, (const float *)P.partial, (float *)P.out, nout);
                else
                    hipLaunchKernelGGL((kirch_combine_kernel<float, 2>), cg, dim3(256), 0, st, (const float *)P.partial, (float *)P.out, nout);
            } else {
                if (P.parts_log2 == 2)
                    hipLaunchKernelGGL((kirch_combine_kernel<double, 4>), cg, dim3(256), 0, st, (const double *)P.partial, (double *)P.out, nout);
                else
                    hipLaunchKernelGGL((kirch_combine_kernel<double, 2>), cg, dim3(256), 0, st, (const double *)P.partial, (double *)P.out, nout);
            }
        }
    }
    IMPDAR_HIP_CHECK(hipGetLastError());
    return IMPDAR_OK;
}

template <int XB, int SH, int NH = 1>
static int launch_dquad(impdar_kirch_plan *p, const FastParams &P0, hipStream_t st)
{
    FastParams P = P0;
    const int ntiles = (P.xhi - (P.xlo & ~3) + XB * NH - 1) / (XB * NH);      // workgroup tiles start at a multiple of 4
    P.nxt = ntiles;
    {
        P.G = ntiles >= 256 ? 4 : 1;
    }
    const int per = 8 * P.G;
    const int nxt_pad = ((ntiles + per - 1) / per) * per;
    P.tiles_per_xcd = nxt_pad / 8;
    const int nblk = P.nchunks * nxt_pad;
    const int W = p->quadW;
    {
        const int trc = build_tilemap(p, P, XB * NH, 3, kd_ring_slots(XB) / 4, 4, st);
        if (trc) return trc;
    }
    const size_t shmem = (size_t)(W / 32) * kd_piece_bytes(XB) * (p->nearfield ? 2 : 1) + 16;      // + the item slot
    if (p->nearfield) {
        auto k = kirch_dquad_kernel<XB, true, 1, SH, 1>;
        IMPDAR_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(k, dim3(nblk), dim3(KF_THREADS), shmem, st, P, W);
    } else {
        auto k = kirch_dquad_kernel<XB, false, (NH > 1 ? 4 : 2), SH, NH>;
        IMPDAR_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        int grid = nblk;
        if (P.tilemap) {
            if (p->slots <= 0) {        // resident workgroups of this plan's kernel on this device: asked once
                int per_cu = 0, ncu = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k, KF_THREADS * NH, shmem) == hipSuccess &&
                    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, p->ctx->device) == hipSuccess &&
                    per_cu > 0 && ncu > 0)
                    p->slots = per_cu * ncu;
                else
                    (void)hipGetLastError();
            }
            // many-rank plans: every walk in 2 or 4 pieces with partial images of their own (p->walk_parts_log2)
            const int pl2 = NH == 1 ? p->walk_parts_log2 : 0;
            const size_t nout = (size_t)P.snum * P.ldo, esz = impdar_dtype_size(p->dtype);
            const bool pieces = pl2 > 0 && p->slots > 0 && p->d_partial.ensure((nout << pl2) * esz) == hipSuccess;
            if (p->slots > 0 && (p->slots < nblk || pieces) && p->d_queue.ensure(8 * 64) == hipSuccess) {
                IMPDAR_HIP_CHECK(hipMemsetAsync(p->d_queue.p, 0, 8 * 64, st));
                P.queue = p->d_queue.as<int>();
                if (pieces) {
                    P.parts_log2 = pl2;
                    P.partial = p->d_partial.p;
                    P.part_stride = nout;
                    // a piece no workgroup reaches (tiles beyond the block's end) must read as zero
                    IMPDAR_HIP_CHECK(hipMemsetAsync(p->d_partial.p, 0, (nout << pl2) * esz, st));
                }
                grid = (int)std::min<long long>(std::max(p->slots - kirch_reserved_slots(p), 1), (long long)nblk << P.parts_log2);
            }
        }
        hipLaunchKernelGGL(k, dim3(grid), dim3(KF_THREADS * NH), shmem, st, P, W);
        if (P.parts_log2) {
            const size_t nout = (size_t)P.snum * P.ldo;
            const dim3 cg((unsigned)((nout + 255) / 256));
            if (p->dtype == IMPDAR_F32) {
                if (P.parts_log2 == 2)
                    hipLaunchKernelGGL((kirch_combine_kernel<float, 4>), cg, dim3(256), 0, st, (const float *)P.partial, (float *)P.out, nout);
                else
                    hipLaunchKernelGGL((kirch_combine_kernel<float, 2>), cg, dim3(256), 0, st, (const float *)P.partial, (float *)P.out, nout);
            } else {
                if (P.parts_log2 == 2)
                    hipLaunchKernelGGL((kirch_combine_kernel<double, 4>), cg, dim3(256), 0, st, (const double *)P.partial, (double *)P.out, nout);
                else
                    hipLaunchKernelGGL((kirch_combine_kernel<double, 2>), cg, dim3(256), 0, st, (const double *)P.partial, (double *)P.out, nout);
            }
        }
    }
    IMPDAR_HIP_CHECK(hipGetLastError());
    return IMPDAR_OK;
}

template <int XB, int S, int OCC>
static int launch_tab(impdar_kirch_plan *p, const FastParams &P0, int nx, hipStream_t st)
{
    constexpr int R = XB + 2 * S;
    FastParams P = P0;
    const int ntiles = (nx + XB - 1) / XB;
    P.nxt = ntiles;
    P.G = 4;
    const int per = 8 * P.G;
    const int nxt_pad = ((ntiles + per - 1) / per) * per;
    P.tiles_per_xcd = nxt_pad / 8;
    const int nblk = P.nchunks * nxt_pad;
    const size_t shmem = (size_t)R * KF_W * 4 * (p->nearfield ? 2 : 1);
    if (p->nearfield) {
        auto k = kirch_tab_kernel<XB, S, true, 1>;
        IMPDAR_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(k, dim3(nblk), dim3(KF_THREADS), shmem, st, P);
    } else {
        auto k = kirch_tab_kernel<XB, S, false, OCC>;
        IMPDAR_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(k, dim3(nblk), dim3(KF_THREADS), shmem, st, P);
    }
    IMPDAR_HIP_CHECK(hipGetLastError());
    return IMPDAR_OK;
}

extern "C" int impdar_kirch_migrate(impdar_kirch_plan *p, void *d_out, int xlo, int xhi)
{
    IMPDAR_ARG_CHECK(p && d_out, "null plan/output");
    IMPDAR_ARG_CHECK(0 <= xlo && xlo <= xhi && xhi <= p->tnum, "bad output trace range [%d,%d) for tnum %d", xlo,
                     xhi, p->tnum);
    IMPDAR_HIP_CHECK(hipSetDevice(p->ctx->device));
    hipStream_t st = p->ctx->stream;
    hipEvent_t *ev = p->evs[p->slot];
    const int b = p->buf;
    IMPDAR_HIP_CHECK(hipStreamWaitEvent(st, p->ev_ready[b], 0));      // image + table of this radargram
    if (!p->haves[p->slot][2]) IMPDAR_HIP_CHECK(hipEventRecord(ev[4], st));
    const int nx = xhi - xlo;
    if (nx > 0 && p->gen) {
        const int grc = kirch_launch_gen(p, d_out, xlo, xhi, st);
        if (grc) return grc;
    } else if (nx > 0 && (p->mode == IMPDAR_KIRCH_FAST || p->dquad)) {
        FastParams P;
        P.GT = reinterpret_cast<const float *>(img_row0(p, p->GT[b]));
        P.DT = p->nearfield ? reinterpret_cast<const float *>(img_row0(p, p->DT[b])) : nullptr;
        P.out = reinterpret_cast<float *>(d_out);
        P.ldo = nx;
        P.snum = p->snum;
        P.tnum = p->tnum;
        P.xlo = xlo;
        P.xhi = xhi;
        P.hmax = p->d_hmax.as<int>();
        P.klo = p->d_klo.as<int>();
        P.khi = p->d_khi.as<int>();
        P.nb = p->nb;
        P.zero_row = p->tnum_pad;          // first zero row after the data rows
        P.nchunks = p->nchunks;
        P.TK = p->d_TK[b].as<unsigned short>();
        P.TW = p->d_TW[b].as<float>();
        P.TW2 = p->d_TW2[b].as<float>();
        P.ntab = p->ntab;
        P.c1 = p->d_c1.as<float>();
        P.c2 = p->d_c2.as<float>();
        P.fin = p->d_fin.as<float>();
        P.c1d = p->d_c1d.as<double>();
        P.c2d = p->d_c2d.as<double>();
        P.find = p->d_find.as<double>();
        P.TKB = p->d_TK[b].p;
        P.WIN = p->d_WIN.as<int2>();
        P.nrows = p->nrows;
        P.mrow0 = p->mrow0;
        P.tilemap = nullptr;
        P.queue = nullptr;
        P.parts_log2 = 0;
        P.partial = nullptr;
        P.part_stride = 0;
        int rc;
        if (p->dquad && p->nh == 2 && p->quadSH == 4)
            rc = p->xb == 20 ? launch_dquad<20, 4, 2>(p, P, st) : launch_dquad<16, 4, 2>(p, P, st);
        else if (p->dquad)
            rc = p->quadSH == 0 ? (p->xb == 20 ? launch_dquad<20, 0>(p, P, st) : launch_dquad<16, 0>(p, P, st))
                                : (p->xb == 20 ? launch_dquad<20, 4>(p, P, st) : launch_dquad<16, 4>(p, P, st));
        else if (p->quad && p->quadSH == 0)
            rc = p->xb == 40 ? launch_quad<40, 2, 0>(p, P, nx, st)
                 : p->xb == 32 ? launch_quad<32, 2, 0>(p, P, nx, st)
                               : launch_quad<24, 3, 0>(p, P, nx, st);
        else if (p->quad && p->nh == 2 && p->xb == 32)
            rc = launch_quad<32, 4, 4, 2, 0>(p, P, nx, st);      // 70 KB of LDS: two workgroups of 8 waves per CU
        else if (p->quad && p->nh == 2 && p->xb == 40)
            rc = p->lk ? launch_quad<40, 2, 4, 2, 1>(p, P, nx, st) : launch_quad<40, 2, 4, 2, 0>(p, P, nx, st);
        else if (p->quad && p->nh == 3 && p->xb == 40)
            rc = p->lk ? launch_quad<40, 3, 4, 3, 1>(p, P, nx, st) : launch_quad<40, 3, 4, 3, 0>(p, P, nx, st);
        else if (p->quad)
            rc = p->xb == 40 ? launch_quad<40, 2, 4>(p, P, nx, st)
                 : p->xb == 32 ? launch_quad<32, 2, 4>(p, P, nx, st)
                             : launch_quad<24, 2, 4>(p, P, nx, st);      // steep moveout: two workgroups per CU
        else
            rc = launch_tab<16, 4, 4>(p, P, nx, st);
        if (rc) return rc;
    } else if (nx > 0 && p->uniform && !p->xtab_off && !(getenv("IMPDAR_KIRCH_EXACT_IMPL") &&
                                                          !strcmp(getenv("IMPDAR_KIRCH_EXACT_IMPL"), "pair"))) {
        // exact arithmetic on uniform grids: tabulated fp64 picks / weights, one gather + FMA per pair
        const int nch = (p->snum + 255) / 256;
        if (!p->xtab_ready) {
            int hg = 0;
            std::vector<int> hm(nch, 0);
            for (int k = 0; k < p->snum; ++k) {
                hm[k / 256] = std::max(hm[k / 256], p->h_half[k] + 1);
                hg = std::max(hg, p->h_half[k] + 1);
            }
            hg = std::min(hg, p->tnum) + 1;
            const size_t ent = (size_t)hg * p->snum;
            if (p->d_XK.ensure(ent * 4) != hipSuccess || p->d_XW.ensure(ent * 8) != hipSuccess ||
                (p->nearfield && p->d_XW2.ensure(ent * 8) != hipSuccess) || p->d_xhmax.ensure((size_t)nch * 4) != hipSuccess) {
                impdar_set_error("hipMalloc of the %zu-entry exact pick/weight table failed", ent);
                return IMPDAR_ERR_HIP;
            }
            IMPDAR_HIP_CHECK(hipMemcpyAsync(p->d_xhmax.p, hm.data(), (size_t)nch * 4, hipMemcpyHostToDevice, st));
            IMPDAR_HIP_CHECK(hipStreamSynchronize(st));          // hm is a stack vector
            TableXParams T;
            T.XK = p->d_XK.as<int>();
            T.XW = p->d_XW.as<double>();
            T.XW2 = p->d_XW2.as<double>();
            T.zs = p->d_zs.as<double>();
            T.zs2 = p->d_zs2.as<double>();
            T.tt = p->d_tt.as<double>();
            T.dx = p->dx;
            T.vel = p->vel;
            T.tmax = p->tmax;
            T.inv_dt = 1.0 / p->dt;
            T.tt0 = p->tt0;
            T.snum = p->snum;
            T.ntab = hg;
            T.near = p->nearfield;
            hipLaunchKernelGGL(kirch_tablex_kernel, dim3(nch, hg), dim3(256), 0, st, T);
            IMPDAR_HIP_CHECK(hipGetLastError());
            p->xntab = hg;
            p->xtab_ready = true;
        }
        ExactTabParams P;
        P.GT = img_row0(p, p->GT[b]);
        P.DT = p->nearfield ? img_row0(p, p->DT[b]) : nullptr;
        P.out = d_out;
        P.ldo = nx;
        P.snum = p->snum;
        P.tnum = p->tnum;
        P.xlo = xlo;
        P.xhi = xhi;
        P.XK = p->d_XK.as<int>();
        P.XW = p->d_XW.as<double>();
        P.XW2 = p->d_XW2.as<double>();
        P.hmax = p->d_xhmax.as<int>();
        P.ntab = p->xntab;
#ifndef KX_XE
#define KX_XE 16
#endif
        constexpr int XE = KX_XE;
        dim3 grid(nch, (nx + XE - 1) / XE);
        if (p->dtype == IMPDAR_F32) {
            if (p->nearfield)
                hipLaunchKernelGGL((kirch_exact_tab_kernel<float, true, XE>), grid, dim3(256), 0, st, P);
            else
                hipLaunchKernelGGL((kirch_exact_tab_kernel<float, false, XE>), grid, dim3(256), 0, st, P);
        } else {
            if (p->nearfield)
                hipLaunchKernelGGL((kirch_exact_tab_kernel<double, true, XE>), grid, dim3(256), 0, st, P);
            else
                hipLaunchKernelGGL((kirch_exact_tab_kernel<double, false, XE>), grid, dim3(256), 0, st, P);
        }
        IMPDAR_HIP_CHECK(hipGetLastError());
    } else if (nx > 0) {
        ExactParams P;
        P.GT = img_row0(p, p->GT[b]);
        P.DT = p->nearfield ? img_row0(p, p->DT[b]) : nullptr;
        P.out = d_out;
        P.ldo = nx;
        P.snum = p->snum;
        P.tnum = p->tnum;
        P.xlo = xlo;
        P.xhi = xhi;
        P.dist = p->d_dist.as<double>();
        P.zs = p->d_zs.as<double>();
        P.zs2 = p->d_zs2.as<double>();
        P.tt = p->d_tt.as<double>();
        P.vel = p->vel;
        P.tmax = p->tmax;
        const double rlim = p->vel * p->tmax / 2.0;
        P.r2lim = rlim * rlim * (1.0 + 1e-9);
        P.inv_dt = 1.0 / p->dt;
        P.tt0 = p->tt0;
        P.dist_sorted = p->dist_sorted ? 1 : 0;
        dim3 grid((p->snum + 255) / 256, nx);
        if (p->dtype == IMPDAR_F32) {
            if (p->nearfield)
                hipLaunchKernelGGL((kirch_exact_kernel<float, true>), grid, dim3(256), 0, st, P);
            else
                hipLaunchKernelGGL((kirch_exact_kernel<float, false>), grid, dim3(256), 0, st, P);
        } else {
            if (p->nearfield)
                hipLaunchKernelGGL((kirch_exact_kernel<double, true>), grid, dim3(256), 0, st, P);
            else
                hipLaunchKernelGGL((kirch_exact_kernel<double, false>), grid, dim3(256), 0, st, P);
        }
        IMPDAR_HIP_CHECK(hipGetLastError());
    }
    if (nx > 0 && p->ntie_groups > 0 && impdar_kirch_plan_kernel(p) != IMPDAR_KERNEL_EXACT_PAIR) {
        // the (sample, offset) entries whose pick rounding noise decides, pair by pair (kirch_tiefix_kernel)
        const int kern = impdar_kirch_plan_kernel(p);
        TieFixParams F;
        F.GT = img_row0(p, p->GT[b]);
        F.DT = p->nearfield ? img_row0(p, p->DT[b]) : nullptr;
        F.out = d_out;
        F.ldo = nx;
        F.snum = p->snum;
        F.tnum = p->tnum;
        F.xlo = xlo;
        F.xhi = xhi;
        F.grp = kern == IMPDAR_KERNEL_QUAD ? 8 : (kern == IMPDAR_KERNEL_DQUAD ? 4 : 0);
        F.near = p->nearfield;
        F.dist = p->d_dist.as<double>();
        F.zs = p->d_zs.as<double>();
        F.zs2 = p->d_zs2.as<double>();
        F.tt = p->d_tt.as<double>();
        F.dx = p->dx;
        F.vel = p->vel;
        F.tmax = p->tmax;
        F.inv_dt = 1.0 / p->dt;
        F.tt0 = p->tt0;
        F.g_ti = p->d_tie_ti.as<int>();
        F.g_off = p->d_tie_off.as<int>();
        F.g_n = p->d_tie_n.as<int>();
        const bool xtab = kern == IMPDAR_KERNEL_EXACT_TAB;
        F.hmax = xtab ? p->d_xhmax.as<int>() : p->d_hmax.as<int>();
        F.nmax = xtab ? p->xntab : p->ntab - 1;
        const dim3 grid((nx + 255) / 256, p->ntie_groups);
        if (p->dtype == IMPDAR_F32)
            hipLaunchKernelGGL(kirch_tiefix_kernel<float>, grid, dim3(256), 0, st, F);
        else
            hipLaunchKernelGGL(kirch_tiefix_kernel<double>, grid, dim3(256), 0, st, F);
        IMPDAR_HIP_CHECK(hipGetLastError());
    }
    IMPDAR_HIP_CHECK(hipEventRecord(ev[5], st));
    IMPDAR_HIP_CHECK(hipEventRecord(p->ev_free[b], st));
    p->free_recorded[b] = true;
    p->last_out = d_out;
    p->last_out_buf = b;
    p->migrated_since_prep = true;
    p->haves[p->slot][2] = true;
    return IMPDAR_OK;
}

// defined in comm.hip
int impdar_allgather_rows(impdar_ctx *ctx, void *image, size_t bytes_per_rank, hipStream_t stream);

extern "C" int impdar_kirch_allgather(impdar_kirch_plan *p)
{
    IMPDAR_ARG_CHECK(p, "null plan");
    // (IMPDAR_COMM_EMULATE=1, diagnostics: a plan built for N ranks driven over a 1-rank communicator -- one rank of an
    // N-rank run emulated on a single GPU with self send/recv, profiles/tools/exchange_overlap.py)
    IMPDAR_ARG_CHECK(p->nranks == p->ctx->nranks || getenv("IMPDAR_COMM_EMULATE"),
                     "plan was built for %d ranks but the communicator has %d", p->nranks, p->ctx->nranks);
    IMPDAR_HIP_CHECK(hipSetDevice(p->ctx->device));
    hipStream_t st = p->ctx->aux;                       // behind this radargram's prep
    hipEvent_t *ev = p->evs[p->slot];
    const int b = p->buf;
    IMPDAR_HIP_CHECK(hipEventRecord(ev[2], st));
    if (p->nranks > 1 || p->ctx->comm) {        // a 1-rank communicator still runs the (trivial) collective
        const size_t per = (size_t)(p->tnum_pad / p->nranks) * p->snum * impdar_dtype_size(p->dtype);
        int rc = impdar_allgather_rows(p->ctx, img_row0(p, p->GT[b]), per, st);
        if (rc) return rc;
        if (p->nearfield && (rc = impdar_allgather_rows(p->ctx, img_row0(p, p->DT[b]), per, st))) return rc;
    }
    IMPDAR_HIP_CHECK(hipEventRecord(ev[3], st));
    IMPDAR_HIP_CHECK(hipEventRecord(p->ev_ready[b], st));
    p->haves[p->slot][1] = true;
    return IMPDAR_OK;
}

// defined in comm.hip
int impdar_exchange_ranges(impdar_ctx *ctx, void *image, int nsend, const int *speer, const size_t *soff,
                           const size_t *slen, int nrecv, const int *rpeer, const size_t *roff, const size_t *rlen,
                           hipStream_t stream);

// Halo exchange: grouped ncclSend/ncclRecv of image-row ranges instead of the all-gather, for shards whose
// aperture halo is narrower than the rest of the profile.  Row ranges are whole 8-trace groups
// (contiguous in both image layouts); rows nobody sends stay whatever the buffer set held -- the kernels never
// pick from rows beyond their block's aperture.
extern "C" int impdar_kirch_exchange(impdar_kirch_plan *p, int nsend, const int *speer, const int *slo, const int *shi,
                                     int nrecv, const int *rpeer, const int *rlo, const int *rhi)
{
    IMPDAR_ARG_CHECK(p, "null plan");
    IMPDAR_ARG_CHECK(nsend >= 0 && nrecv >= 0 && nsend <= 4096 && nrecv <= 4096, "bad range counts %d / %d", nsend, nrecv);
    IMPDAR_ARG_CHECK((nsend == 0 || (speer && slo && shi)) && (nrecv == 0 || (rpeer && rlo && rhi)), "null range arrays");
    // (IMPDAR_COMM_EMULATE=1, diagnostics: a plan built for N ranks driven over a 1-rank communicator -- one rank of an
    // N-rank run emulated on a single GPU with self send/recv, profiles/tools/exchange_overlap.py)
    IMPDAR_ARG_CHECK(p->nranks == p->ctx->nranks || getenv("IMPDAR_COMM_EMULATE"),
                     "plan was built for %d ranks but the communicator has %d", p->nranks, p->ctx->nranks);
    const size_t rowb = (size_t)p->snum * impdar_dtype_size(p->dtype);
    std::vector<size_t> soff(nsend), slen(nsend), roff(nrecv), rlen(nrecv);
    auto conv = [&](int lo, int hi, size_t &off, size_t &len) {
        if (lo < 0 || hi < lo || hi > p->tnum_pad || (lo & 7) || (hi & 7)) return false;
        off = (size_t)lo * rowb;
        len = (size_t)(hi - lo) * rowb;
        return true;
    };
    for (int i = 0; i < nsend; ++i)
        IMPDAR_ARG_CHECK(conv(slo[i], shi[i], soff[i], slen[i]), "send rows [%d,%d) are not whole 8-trace groups of the image", slo[i], shi[i]);
    for (int i = 0; i < nrecv; ++i)
        IMPDAR_ARG_CHECK(conv(rlo[i], rhi[i], roff[i], rlen[i]), "receive rows [%d,%d) are not whole 8-trace groups of the image", rlo[i], rhi[i]);
    IMPDAR_HIP_CHECK(hipSetDevice(p->ctx->device));
    hipStream_t st = p->ctx->aux;                       // behind this radargram's prep
    hipEvent_t *ev = p->evs[p->slot];
    const int b = p->buf;
    IMPDAR_HIP_CHECK(hipEventRecord(ev[2], st));
    int rc = impdar_exchange_ranges(p->ctx, img_row0(p, p->GT[b]), nsend, speer, soff.data(), slen.data(), nrecv, rpeer,
                                    roff.data(), rlen.data(), st);
    if (rc) return rc;
    if (p->nearfield && (rc = impdar_exchange_ranges(p->ctx, img_row0(p, p->DT[b]), nsend, speer, soff.data(), slen.data(),
                                                     nrecv, rpeer, roff.data(), rlen.data(), st)))
        return rc;
    IMPDAR_HIP_CHECK(hipEventRecord(ev[3], st));
    IMPDAR_HIP_CHECK(hipEventRecord(p->ev_ready[b], st));
    p->haves[p->slot][1] = true;
    return IMPDAR_OK;
}

extern "C" int impdar_kirch_history_ms(impdar_kirch_plan *p, int back, float *prep_ms, float *gather_ms,
                                       float *migrate_ms)
{
    IMPDAR_ARG_CHECK(p, "null plan");
    IMPDAR_ARG_CHECK(back >= 0 && back < impdar_kirch_plan::NSLOT, "history depth is %d steps",
                     impdar_kirch_plan::NSLOT);
    IMPDAR_HIP_CHECK(hipSetDevice(p->ctx->device));
    const int s = ((p->slot - back) % impdar_kirch_plan::NSLOT + impdar_kirch_plan::NSLOT) % impdar_kirch_plan::NSLOT;
    float *outs[3] = {prep_ms, gather_ms, migrate_ms};
    for (int i = 0; i < 3; ++i) {
        if (!outs[i]) continue;
        *outs[i] = 0.f;
        if (!p->haves[s][i]) continue;
        IMPDAR_HIP_CHECK(hipEventSynchronize(p->evs[s][2 * i + 1]));
        IMPDAR_HIP_CHECK(hipEventElapsedTime(outs[i], p->evs[s][2 * i], p->evs[s][2 * i + 1]));
    }
    return IMPDAR_OK;
}

extern "C" int impdar_kirch_last_ms(impdar_kirch_plan *p, float *prep_ms, float *gather_ms, float *migrate_ms)
{
    return impdar_kirch_history_ms(p, 0, prep_ms, gather_ms, migrate_ms);
}


extern "C" long long impdar_kirch_count_pairs(const impdar_kirch_plan *p, int xlo, int xhi)
{
    if (!p || !p->uniform || xlo < 0 || xhi > p->tnum || xlo > xhi) return -1;
    long long total = 0;
    for (int k = 0; k < p->snum; ++k) {
        const long long h = p->h_half[k];
        if (h < 0) continue;
        for (int xi = xlo; xi < xhi; ++xi) {
            const long long lo = std::max<long long>(xi - h, 0), hi = std::min<long long>(xi + h, p->tnum - 1);
            total += hi - lo + 1;
        }
    }
    return total;
}

// ---------------------------------------------------------------------------
// one-shot host-buffer entry point
// ---------------------------------------------------------------------------
// The one-shot entry point keeps its last plan and device buffers (as the Stolt and phase-shift entry points do):
// a second radargram of the same geometry skips the plan, the pick table, the allocations and their release
// (19.9 -> 18.1 ms host-to-host at config 3).
// IMPDAR_KIRCH_ONESHOT_CACHE=0 releases everything at the end of each call, as before.
namespace {
struct KirchOneShot {
    const impdar_ctx *owner = nullptr;
    impdar_kirch_plan *plan = nullptr;
    DevBuf din, dout;
    hipEvent_t ev_blk[8] = {};       // output block i of a split one-shot call summed (see impdar_kirchhoff)
    int dtype = -1, snum = 0, tnum = 0, nearfield = 0, grad_uniform = 0, mode = 0;
    double vel = 0, grad_h = 0;
    std::vector<double> dist, tt, ga, gb, gc;
    std::string knobs;       // the IMPDAR_KIRCH_* settings the plan was built under
    void drop()
    {
        if (plan) impdar_kirch_plan_destroy(plan);
        plan = nullptr;
        din.release();
        dout.release();
        for (hipEvent_t &e : ev_blk) {
            if (e) (void)hipEventDestroy(e);
            e = nullptr;
        }
        owner = nullptr;
    }
};
std::mutex g_k1_mu;
KirchOneShot *g_k1 = nullptr;
thread_local bool t_k1_busy = false, t_hook_busy = false;     // this thread is inside impdar_kirchhoff / mig_kirch_loop

std::string kirch_knobs()
{
    static const char *names[] = {"IMPDAR_KIRCH_EXACT_IMPL", "IMPDAR_KIRCH_IMPL", "IMPDAR_KIRCH_LK", "IMPDAR_KIRCH_NH",
                                  "IMPDAR_KIRCH_NHD", "IMPDAR_KIRCH_PARTS", "IMPDAR_KIRCH_TIEFIX", "IMPDAR_KIRCH_XB",
                                  "IMPDAR_KIRCH_XBD", "IMPDAR_KIRCH_MODE"};
    std::string k;
    for (const char *n : names) {
        const char *v = getenv(n);
        k += v ? v : "";
        k += ';';
    }
    return k;
}

bool same_vec(const std::vector<double> &have, const double *p, size_t n)
{
    if (!p) return have.empty();
    return have.size() == n && memcmp(have.data(), p, n * sizeof(double)) == 0;
}
}   // namespace

// called by impdar_ctx_destroy: the cached plan must not outlive the context it was created on
void impdar_kirch_forget(const impdar_ctx *ctx)
{
    std::lock_guard<std::mutex> lk(g_k1_mu);
    if (g_k1 && g_k1->owner == ctx) {
        g_k1->drop();
        delete g_k1;
        g_k1 = nullptr;
    }
}

extern "C" int impdar_kirchhoff(impdar_ctx *ctx, const void *data, int dtype, int snum, int tnum,
                                const double *dist_m, const double *tt_sec, double vel, int nearfield,
                                int grad_uniform, double grad_h, const double *ga, const double *gb,
                                const double *gc, int mode, double *out)
{
    IMPDAR_ARG_CHECK(ctx && data && out, "null context/data/output");
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t0 = now();
    std::lock_guard<std::mutex> lk(g_k1_mu);
    ImpdarBusy busy(t_k1_busy);
    if (!g_k1) g_k1 = new KirchOneShot();
    KirchOneShot &c = *g_k1;
    const char *ce = getenv("IMPDAR_KIRCH_ONESHOT_CACHE");
    const bool keep = !(ce && ce[0] == '0');
    const std::string knobs = kirch_knobs();
    const bool hit = c.plan && c.owner == ctx && c.dtype == dtype && c.snum == snum && c.tnum == tnum && c.vel == vel &&
                     c.nearfield == nearfield && c.grad_uniform == grad_uniform && c.grad_h == grad_h && c.mode == mode &&
                     dist_m && tt_sec && same_vec(c.dist, dist_m, (size_t)tnum) && same_vec(c.tt, tt_sec, (size_t)snum) &&
                     same_vec(c.ga, ga, (size_t)snum) && same_vec(c.gb, gb, (size_t)snum) &&
                     same_vec(c.gc, gc, (size_t)snum) && c.knobs == knobs;
    int rc;
    if (!hit) {
        c.drop();
        if ((rc = impdar_kirch_plan_create(ctx, dtype, snum, tnum, dist_m, tt_sec, vel, nearfield, grad_uniform, grad_h,
                                           ga, gb, gc, mode, 1, &c.plan))) {
            c.plan = nullptr;
            return rc;
        }
        c.owner = ctx;
        c.dtype = dtype, c.snum = snum, c.tnum = tnum, c.nearfield = nearfield, c.grad_uniform = grad_uniform, c.mode = mode;
        c.vel = vel, c.grad_h = grad_h;
        c.dist.assign(dist_m, dist_m + tnum);
        c.tt.assign(tt_sec, tt_sec + snum);
        auto take = [&](std::vector<double> &v, const double *p) {
            if (p) v.assign(p, p + snum);
            else v.clear();
        };
        take(c.ga, ga);
        take(c.gb, gb);
        take(c.gc, gc);
        c.knobs = knobs;
    }
    impdar_kirch_plan *p = c.plan;
    const size_t esz = impdar_dtype_size(dtype);
    const size_t bytes = (size_t)snum * tnum * esz;
    auto done = [&](int code) {
        if (code != IMPDAR_OK) {
            // the uploads of the pipelined form read the caller's array asynchronously: nothing may still be in flight
            // when an error hands the array back
            (void)hipStreamSynchronize(ctx->aux);
            (void)hipStreamSynchronize(ctx->stream);
        }
        if (code != IMPDAR_OK || !keep) c.drop();
        return code;
    };
    if (c.din.ensure(bytes) != hipSuccess || c.dout.ensure(bytes) != hipSuccess) {
        impdar_set_error("hipMalloc of %zu bytes failed", bytes);
        return done(IMPDAR_ERR_HIP);
    }
    const auto t0b = now();
    // Large radargrams on the ring kernels go through in pieces, so that PCIe and the kernels work at the same time.
    // OUTPUT traces in blocks, one launch each (output blocks equal the whole image bit for bit: every output
    // accumulates its pairs in the same order): a block crosses PCIe and is widened on the host while the next is
    // summed.  INPUT traces in column chunks: block i reads the traces up to its right edge + the aperture half width
    // only (the halo of the multi-GPU plan), so its launch starts when those are on the device and the rest of the
    // upload runs under it; later chunks are prepared into the same image (kirch_prep_impl's `more`).  Every launch
    // must drain (~0.3 ms each) -- the copies they hide are worth more.  Four blocks [0, .10, .40, .70, 1] tnum when
    // the aperture leaves something to overlap (config 3: the first launch needs 45 % of the input; a sweep of the
    // cuts is in profiles/r03_oneshot_pipeline.txt), else two (5/8 + 3/8, download overlap only).
    // IMPDAR_KIRCH_ONESHOT_SPLIT=0: one upload, one launch, one download; =2: one upload, two launches.
    const char *se = getenv("IMPDAR_KIRCH_ONESHOT_SPLIT");
    const int kern = impdar_kirch_plan_kernel(p);
    const bool split = !(se && atoi(se) == 0) && tnum >= 4096 && bytes >= ((size_t)64 << 20) &&
                       (kern == IMPDAR_KERNEL_QUAD || kern == IMPDAR_KERNEL_DQUAD);
    // the download's staging buffer, pinned by a thread of its own while this call uploads and sums (started behind the
    // plan: beside it the two contend for the runtime -- plan 18 -> 40 ms): the 64 MB ring -- of the pipelined blocks, and
    // since round 6 of the one-piece download too (api.hip)
    impdar_ctx_pinned_prefetch(ctx, std::min(bytes, IMPDAR_STAGE_RING_BYTES));
    int nlaunch = 1;
    if (split) {
        const int halo = p->ntab + 16;                 // aperture half width (+ the kernels' staging look-ahead)
        auto r8 = [](long long x) { return (int)(x / 8 * 8); };
        std::vector<int> cut;
        const bool two = se && atoi(se) == 2;          // the round-3 first form, kept for A/B
#ifndef KOS_PCT
#define KOS_PCT {10, 40, 70}
#endif
        const std::vector<int> pct = KOS_PCT;              // interior cuts in percent of tnum (sweeps: profiles/r03_oneshot_pipeline.txt, r05_oneshot_cuts.txt)
        if (!two && (long long)tnum * pct[pct.size() > 1 ? 1 : 0] / 100 + halo < (long long)tnum * 9 / 10) {
            cut = {0};
            for (int q : pct) cut.push_back(r8((long long)tnum * q / 100));
            cut.push_back(tnum);
        } else {
            cut = {0, r8((long long)tnum * 5 / 8), tnum};
        }
        const int nblk = (int)cut.size() - 1;
        nlaunch = nblk;
        for (int i = 0; i < nblk; ++i)
            if (!c.ev_blk[i] && hipEventCreateWithFlags(&c.ev_blk[i], hipEventDisableTiming) != hipSuccess) {
                c.ev_blk[i] = nullptr;
                impdar_set_error("hipEventCreate failed");
                return done(IMPDAR_ERR_HIP);
            }
        char *dout = reinterpret_cast<char *>(c.dout.p);
        int have = 0;                                  // input traces [0, have) are on the device and prepared
        for (int i = 0; i < nblk; ++i) {
            const int need = two ? tnum : (int)std::min<long long>(tnum, ((long long)cut[i + 1] + halo + 7) / 8 * 8);
            if (need > have) {
                // column block [have, need) of the (snum, tnum) host array; on the producer stream, in front of its prep
                if (hipMemcpy2DAsync(reinterpret_cast<char *>(c.din.p) + (size_t)have * esz, (size_t)tnum * esz,
                                     reinterpret_cast<const char *>(data) + (size_t)have * esz, (size_t)tnum * esz,
                                     (size_t)(need - have) * esz, (size_t)snum, hipMemcpyHostToDevice, ctx->aux) != hipSuccess) {
                    impdar_set_error("H2D copy failed");
                    return done(IMPDAR_ERR_HIP);
                }
                if ((rc = kirch_prep_impl(p, reinterpret_cast<const char *>(c.din.p) + (size_t)have * esz, tnum, have,
                                          need - have, 0, have > 0)))
                    return done(rc);
                have = need;
            }
            if ((rc = impdar_kirch_migrate(p, dout + (size_t)snum * cut[i] * esz, cut[i], cut[i + 1]))) return done(rc);
            if (hipEventRecord(c.ev_blk[i], ctx->stream) != hipSuccess) return done(IMPDAR_ERR_HIP);
        }
        // the blocks leave on the producer stream (idle after the last prep) as their launches finish: all copies
        // enqueued at once, the host widens what has arrived
        {
            std::vector<size_t> col0(nblk), width(nblk);
            std::vector<const void *> src(nblk);
            for (int i = 0; i < nblk; ++i) {
                col0[i] = (size_t)cut[i];
                width[i] = (size_t)(cut[i + 1] - cut[i]);
                src[i] = dout + (size_t)snum * cut[i] * esz;
            }
            if ((rc = impdar_download_blocks_f64(ctx, out, (size_t)tnum, (size_t)snum, dtype, nblk, col0.data(), width.data(),
                                                 src.data(), c.ev_blk, ctx->aux)))
                return done(rc);
        }
    } else {
        if (hipMemcpyAsync(c.din.p, data, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {      // prep runs on the aux stream
            impdar_set_error("H2D copy failed");
            return done(IMPDAR_ERR_HIP);
        }
        if ((rc = impdar_kirch_prep(p, c.din.p, tnum, 0, tnum))) return done(rc);
        if ((rc = impdar_kirch_migrate(p, c.dout.p, 0, tnum))) return done(rc);
        // device -> pinned staging (in pieces) -> the caller's float64 array on several host threads
        // (mig_python.py:118 returns float64); waits for the diffraction sum on the compute stream
        if ((rc = impdar_dev_download_f64(ctx, out, c.dout.p, dtype, (size_t)snum * tnum))) return done(rc);
    }
    // what this call did, for impdar_ctx_last_metrics (the downloads above have synchronised the streams)
    float kms = -1.f;
    {
        float a = 0.f, b = 0.f, c2 = 0.f;
        if (impdar_kirch_last_ms(p, &a, &b, &c2) == IMPDAR_OK) kms = c2;       // first launch's start to last launch's end
        (void)hipGetLastError();
    }
    static const char *kernel_names[] = {"kirch_exact_kernel", "kirch_exact_tab_kernel", "kirch_dquad_kernel", "kirch_quad_kernel",
                                         "kirch_tab_kernel", "kirch_gen_kernel"};
    ctx->m_entry = "impdar_kirchhoff";
    ctx->m_kernel = (kern >= 0 && kern < 6) ? kernel_names[kern] : "";
    ctx->m_kernel_ms = kms;
    ctx->timed = false;
    const auto t2 = now();
    // xnoise: the position noise of the profile against its fitted grid, in trace spacings (impdar_kirch_plan_xnoise); the
    // float64 ring / tabulated kernels weight a pair by its trace OFFSET, so their stated bar against the reference is
    // max(1e-12, 0.1 xnoise) of the image maximum -- parity_bar says which one this call ran under (the per-pair kernel,
    // IMPDAR_KIRCH_EXACT_IMPL=pair, holds 1e-12 on any profile)
    const bool offset_weighted = kern == IMPDAR_KERNEL_DQUAD || kern == IMPDAR_KERNEL_EXACT_TAB;
    snprintf(ctx->m_extra, sizeof ctx->m_extra,
             "\"plan\": \"%s\", \"launches\": %d, \"plan_ms\": %.3f, \"call_ms\": %.3f, \"xnoise\": %.3g, \"parity_bar\": %.3g",
             hit ? "cached" : "new", nlaunch, ms(t0, t0b), ms(t0, t2), p->xnoise,
             dtype == IMPDAR_F32 && p->mode == IMPDAR_KIRCH_FAST ? 1e-4 : (offset_weighted ? std::max(1e-12, 0.1 * p->xnoise) : 1e-12));
    rc = done(IMPDAR_OK);
    return rc;
}

// ---------------------------------------------------------------------------
// reference-compatible native hook (mig_cython.h:11)
// ---------------------------------------------------------------------------
// void return, no error channel (_mig_cython.pyx:19-20): whatever goes wrong, migdata is filled with NaN -- the
// wrapper returns it as the migrated image, and an untouched (all-zero) array would read as a result.
//
// The caller hands over its own depth tables and time limit (mig_python.py:98-102 computes them).  When they are what
// the reference's driver computes -- zs = vel tt / 2 and zs2 = zs^2 bit for bit, max_travel_time = max(tt) up to the
// single-precision rounding _mig_cython.pyx:30-32 applies to it -- on a uniform profile, the float64 LDS-ring kernel
// runs (kirch_dquad_kernel, the library's own float64 default; its plan is created with the caller's time limit); any
// other tables go through the per-pair kernel, which reads them as given.  The plan and the device buffers are kept
// between calls of one geometry, as the one-shot entry point keeps its own.
int impdar_kirch_prep_precomputed(impdar_kirch_plan *p, const void *d_grad, int ld, int jlo, int nloc);

namespace {
struct KirchHook {
    impdar_ctx *ctx = nullptr;
    impdar_kirch_plan *plan = nullptr;
    DevBuf din, dout;
    int snum = 0, tnum = 0;
    double vel = 0, tmax = 0;
    bool standard = false;
    std::vector<double> dist, tt, zs, zs2;
    std::string knobs;
    void drop()
    {
        if (plan) impdar_kirch_plan_destroy(plan);
        plan = nullptr;
        din.release();
        dout.release();
    }
};
std::mutex g_hook_mu;
KirchHook g_hook;
}   // namespace

// impdar_release_caches: out of device memory somewhere -- drop the one-shot and the hook caches unless their entry
// point is the one running
void impdar_kirch_trim()
{
    if (!t_k1_busy) {
        std::unique_lock<std::mutex> lk(g_k1_mu, std::try_to_lock);
        if (lk.owns_lock() && g_k1) g_k1->drop();
    }
    if (!t_hook_busy) {
        std::unique_lock<std::mutex> lk(g_hook_mu, std::try_to_lock);
        if (lk.owns_lock()) g_hook.drop();
    }
}

extern "C" void mig_kirch_loop(double *migdata, int tnum, int snum, double *dist, double *zs, double *zs2,
                               double *tt_sec, double vel, double *gradD, double max_travel_time, int nearfield)
{
    auto fail = [&](const char *why) {
        fprintf(stderr, "impdar mig_kirch_loop: %s -- migdata filled with NaN\n", why);
        if (migdata && snum > 0 && tnum > 0) {
            const double nan = std::numeric_limits<double>::quiet_NaN();
            for (size_t i = 0, n = (size_t)snum * tnum; i < n; ++i) migdata[i] = nan;
        }
    };
    if (!migdata || !dist || !zs || !zs2 || !tt_sec || !gradD || snum < 1 || tnum < 1) return fail("null pointer or empty array");
    if (nearfield)
        return fail("the reference prototype carries no data pointer, so the near-field term cannot be formed "
                    "(use impdar_kirchhoff)");
    std::lock_guard<std::mutex> lk(g_hook_mu);
    ImpdarBusy busy(t_hook_busy);
    KirchHook &c = g_hook;
    if (!c.ctx && impdar_ctx_create(0, &c.ctx) != IMPDAR_OK) {
        c.ctx = nullptr;
        return fail(impdar_last_error());
    }
    impdar_ctx *ctx = c.ctx;
    double ttmax = tt_sec[0];
    bool standard = true;
    for (int k = 0; k < snum; ++k) {
        ttmax = std::max(ttmax, tt_sec[k]);
        standard = standard && zs[k] == vel * tt_sec[k] / 2.0 && zs2[k] == zs[k] * zs[k];     // mig_python.py:101-102
    }
    standard = standard && std::fabs(max_travel_time - ttmax) <= 1e-6 * std::fabs(ttmax);
    const std::string knobs = kirch_knobs();
    const bool hit = c.plan && c.snum == snum && c.tnum == tnum && c.vel == vel && c.tmax == max_travel_time &&
                     c.standard == standard && same_vec(c.dist, dist, (size_t)tnum) && same_vec(c.tt, tt_sec, (size_t)snum) &&
                     same_vec(c.zs, zs, (size_t)snum) && same_vec(c.zs2, zs2, (size_t)snum) && c.knobs == knobs;
    if (!hit) {
        c.drop();
        g_tmax_override = standard ? &max_travel_time : nullptr;
        const int rc = impdar_kirch_plan_create(ctx, IMPDAR_F64, snum, tnum, dist, tt_sec, vel, 0, 1, 1.0, nullptr, nullptr,
                                                nullptr, IMPDAR_KIRCH_EXACT, 1, &c.plan);
        g_tmax_override = nullptr;
        if (rc) {
            c.plan = nullptr;
            return fail(impdar_last_error());
        }
        if (!standard) {
            // the caller's own tables: per-pair kernel (the table-driven ones derive picks and apertures from the
            // plan's tables)
            c.plan->tmax = max_travel_time;
            c.plan->xtab_off = true;
            c.plan->dquad = false;
            c.plan->ntie_groups = 0;
            if (hipMemcpy(c.plan->d_zs.p, zs, (size_t)snum * 8, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(c.plan->d_zs2.p, zs2, (size_t)snum * 8, hipMemcpyHostToDevice) != hipSuccess) {
                c.drop();
                return fail("upload of the depth tables failed");
            }
        }
        c.snum = snum, c.tnum = tnum, c.vel = vel, c.tmax = max_travel_time, c.standard = standard;
        c.dist.assign(dist, dist + tnum);
        c.tt.assign(tt_sec, tt_sec + snum);
        c.zs.assign(zs, zs + snum);
        c.zs2.assign(zs2, zs2 + snum);
        c.knobs = knobs;
    }
    impdar_kirch_plan *p = c.plan;
    const size_t bytes = (size_t)snum * tnum * 8;
    const bool ok = c.din.ensure(bytes) == hipSuccess && c.dout.ensure(bytes) == hipSuccess &&
                    hipMemcpyAsync(c.din.p, gradD, bytes, hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
                    hipStreamSynchronize(ctx->stream) == hipSuccess &&          // prep runs on the producer stream
                    impdar_kirch_prep_precomputed(p, c.din.p, tnum, 0, tnum) == IMPDAR_OK &&
                    impdar_kirch_migrate(p, c.dout.p, 0, tnum) == IMPDAR_OK &&
                    impdar_download(ctx, migdata, c.dout.p, bytes, ctx->stream) == IMPDAR_OK;
    if (!ok) {
        (void)hipGetLastError();
        c.drop();
        return fail(impdar_last_error());
    }
    const char *ce = getenv("IMPDAR_KIRCH_ONESHOT_CACHE");
    if (ce && ce[0] == '0') c.drop();
}
