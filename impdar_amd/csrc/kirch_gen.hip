// Kirchhoff diffraction sum on a NON-UNIFORM trace spacing: float32 data, uniform travel_time, sorted dist[].
//
// Reference: src/impdar/lib/migrationlib/mig_python.py:35-60 sums over ANY dist[] (:44 takes dist[j] - dist[xi]).
// The ring kernels of kirchhoff.hip tabulate a pair's pick per (sample, trace OFFSET), which only exists on a
// uniform profile; everything else fell to the per-pair float64 kernel (0.77 s at 10000 x 4096).  This kernel keeps
// what makes the ring kernels fast -- input traces staged once per workgroup in LDS, many outputs per lane -- and
// computes the pick of every pair from the positions:
//
//   a workgroup owns 256 output samples (one per lane) x XB output traces (XB float accumulators per lane) and
//   walks the input traces j of the tile's aperture in blocks of 8; a block's traces sit in LDS as 8 slots of W
//   consecutive samples [kmin_j, kmin_j + W) (trace-major: consecutive lanes read consecutive or equal words of a
//   slot, no bank conflicts).  In units of samples (x 2 / (v dt)): a = tt/dt, D = |dist[j] - dist[xi]|,
//       s   = sqrt(a^2 + D^2) - tt[0]/dt        position of the pair's travel time on the sample axis (:49)
//       k   = rint(s)                            nearest sample (ties: below)
//       cos = a / sqrt(a^2 + D^2)                (:47)
//   per pair: q = a^2 + D^2 (D^2 is the same for all lanes: read from a small LDS table the workgroup fills in
//   float64 a block ahead), r = v_rsq(q), the position normalised to the slot p = clamp01((q r - u0 - kmin) / (W - 1))
//   in one fma, rounded by adding 1.5 * 2^23 (the integer lands in the mantissa), one shift-add makes the LDS
//   address, ds_read_b32, acc += r * g; the factor a / (2 pi v) is applied once at the end.  9 vector instructions
//   per pair, chosen by their measured issue cost (see kirch_gen_kernel).  A wave skips the traces that reach none of
//   its 64 samples.
//
// Parity.  A float32 position is good to ~4 ulp (2.4e-7 s) -- a pick within that of a half-way point could round
// either way, and 0.2 % of all picks are that close: on white noise that alone would be 6 % of error.  Every pair
// whose rounded position lies within the bound of a tie is therefore decided again in float64 WITHOUT a square root:
// the pair (m, m + 1) of candidate samples is known, and  k = m + 1  <=>  D^2 + a^2 > (m + 1/2 + tt[0]/dt)^2  in
// units of samples (:44,:49 squared; tt[m] = tt[0] + m dt, which the plan checks to 1e-11 dt).  One uniform branch
// per pair step, out of line, taken by ~10 % of the wave steps.  The result then has the reference's picks except
// where the reference's own float64 rounding of sqrt / divide decides (relative 1e-16: never on a jittered profile;
// a UNIFORM profile with a rational moveout has such pairs -- the plan's tie scan finds them and keeps those profiles
// on the float64 kernels).
//
// The end of the time axis.  The reference drops a pair with t > max(tt) (:52); sample snum - 1 therefore owns only
// the lower half of its cell.  Rather than testing every pair, the staged copy of every trace holds zeros from
// sample snum - 1 on -- the loop needs no time-limit test at all -- and kirch_gen_shell_kernel adds the pairs that
// pick sample snum - 1 (a shell half a sample thick: a few traces per output sample, found by a guess-and-gallop
// search on the sorted dist[]) in the reference's float64 arithmetic afterwards -- including its rounding of
// 2 rs / vel against max(tt), which decides the pairs at zero offset in the last row.
//
// Measured at 10000 x 4096, +-0.3 dx jitter (profiles/r04_gen_*.txt; 0.77 s on the per-pair kernel): 89.5 ms for the
// first correct version; re-decision path without global loads and out of line 72.5; issue-cost-aware step + per-wave
// aperture skip + sums pinned behind their reads (the compiler had sunk all 32 below the last read and spilled) 61.7.
// Then: a^2 / (W-1)^2 rounded once (error bound 5 -> 4 ulp: fewer flagged steps) and the re-decision in units of
// samples (21 -> 18 instructions): 58.8-61.0 ms by box.  Timing-only runs with the flag threshold moved
// (profiles/r04_gen_flag_cost.txt): never flagged 50.2 ms, always flagged 129 ms -- the re-decision path (a whole
// wave for ~1 flagged lane in 11 % of the steps) costs 8.6 ms; the vector-issue model of the step (28 cycles) gives
// 36 ms, the per-block work (tables, staging, barrier: ~250 instructions per 256 pair steps) and the launch's tail
// are the rest.
#include "kirch_plan.h"
#include <algorithm>
#include <cmath>

typedef float kq_f4 __attribute__((ext_vector_type(4)));
typedef unsigned kq_u4 __attribute__((ext_vector_type(4)));

#define KG_S 8                      // input traces per block
#define KG_MAGIC 12582912.0f        // 1.5 * 2^23: x + KG_MAGIC has rint(x) in its low mantissa bits (0 <= x < 2^22)
#define KG_MAGIC_BITS 0x4B400000u

struct GenParams {
    const float *GT, *DT;       // trace-major images [trace][snum]; non-finite values already mapped to 0
    float *out;
    int ldo;
    int snum, tnum, xlo, xhi;
    const double *dist;         // metres [tnum + 64], non-decreasing; the last 64 repeat dist[tnum - 1]
    const double *zs, *zs2;     // [snum] (mig_python.py:101-102)
    const double *tt;           // [snum] travel time, seconds (the reference's own axis: kg_ref_upper)
    const float *a, *a2;        // [snum] tt/dt, and (tt/dt)^2 / (W - 1)^2 rounded once
    const int2 *jr;             // [nchunks][ntiles] first / last input trace inside any aperture of (chunk, tile)
    const float *alo2;          // [nchunks] smallest a^2 of the chunk's samples
    double cscale;              // 2 / (vel dt): metres -> samples
    double r2lim;               // (vel max(tt) / 2)^2
    double tmax;                // max(tt), seconds
    double tt0, dt;             // the time axis: tt[k] = tt0 + k dt (to 1e-11 dt, checked by the plan)
    double hh;                  // 1/2 + tt0 / dt
    double tie2;                // see kg_near_tie
    double vel;
    float nu0;                  // -tt[0]/dt
    float fin;                  // 1 / (2 pi vel)
    float cn;                   // near field: vel / (vel dt / 2)^2
    float e0, e1;               // bound on the error of the float32 position: e0 + e1 * (slot's last sample)
    int W;                      // samples per LDS slot (a multiple of 256)
    int nchunks, ntiles, tiles_per_xcd, G;
};

// (Position of the time half-way between samples m and m + 1, in samples from t = 0)^2: a pair picks m + 1 iff
// D^2 + a^2 exceeds it.
__device__ static inline double kg_halfway_s2(double m, const GenParams &P)
{
    const double t = m + P.hh;
    return t * t;
}

// A pick that sits ON a half-way point (to KG_TIE_EPS samples -- far wider than any float64 rounding, far rarer than
// the float32 flags): the reference decides it by the rounding noise of its own sqrt / divide / subtract
// (mig_python.py:44,:49: argmin |tt[k] - 2 rs / vel|, first minimum), pair by pair.  Lattice positions with a rational
// moveout (an evenly spaced survey with dropped traces) are full of such pairs; a jittered profile has none.  Those
// pairs repeat the reference's operations literally -- metres, seconds, IEEE sqrt and divide, its own tt[] -- between
// the two candidate samples (m, m + 1): true = m + 1.
#define KG_TIE_EPS 1.0e-9
#ifndef KG_TIE_FN
#define KG_TIE_FN __device__ static __attribute__((noinline))
#endif
KG_TIE_FN bool kg_ref_upper(const double *__restrict__ dist, const double *__restrict__ zs2,
                            const double *__restrict__ tt, double vel, int j, int xi, int ti, int m)
{
    const double dx = dist[j] - dist[xi];
    const double rs = sqrt(dx * dx + zs2[ti]);                         // :44
    const double t = 2.0 * rs / vel;                                   // :49
    return fabs(tt[m + 1] - t) < fabs(tt[m] - t);                      // argmin keeps the first of two equal distances
}
// |s^2 - h^2| <= tie2  with  tie2 = 2 KG_TIE_EPS (largest |h| of the record + 1): at least KG_TIE_EPS samples around every
// half-way point (wider around the shallow ones; a false positive costs one literal evaluation)
__device__ static inline bool kg_near_tie(double q, double h2, double tie2) { return fabs(q - h2) <= tie2; }

__host__ __device__ constexpr unsigned kg_lds_bytes(int w, bool near)
{
    return 2u * KG_S * (unsigned)w * 4u * (near ? 2u : 1u) + 3u * KG_S * 32u * 4u + 3u * KG_S * 16u + 32u * 8u + 3u * KG_S * 4u +
           3u * KG_S * 4u;
}

// LDS: [stage 2 x 8 x W floats (x 2 with the data image)] [d2 3 x 8 x XB floats] [meta 3 x 8 x 16 bytes]
//      [positions of the tile's 32 output traces, float64] [dmin^2 3 x 8 floats] [kmin 3 x 8 ints]
//
// The pair step is written for the issue costs measured on this part (profiles/tools/valu_rates.hip,
// profiles/r04_valu_rates.txt): v_add / v_sub / v_mul / v_fma / v_fmaak_f32 on three distinct VGPRs issue every 2
// cycles per SIMD; anything with an SGPR operand, v_min / v_max, every compare, every shift-left and conversion take 4;
// v_rsq 8.  So the position is formed NORMALISED to the slot, p = (s - u0 - kmin) / (W - 1) = fma(q', r', c_j) with
// q' = q / (W - 1)^2 and the clamp modifier doing what v_min did, every operand a VGPR:
//     v_add q'   v_rsq r'   v_fma p (clamp)   v_fmaak f = p (W-1) + 1.5 2^23   v_add kf = f - 1.5 2^23
//     v_fma df = p (W-1) - kf   v_lshl_add addr   v_cmp |df| > thr   ds_read_b32   v_fma acc += r' g
// = 6 x 2 + 2 x 4 + 8 = 28 cycles (31 before).
template <int XB, int WR, bool NEAR>
__global__ __launch_bounds__(256, 4) void kirch_gen_kernel(GenParams P)
{
    static_assert(XB == 32, "thread (i, jj) roles below assume 256 = 32 x 8");
    constexpr int S = KG_S;
    constexpr int W = WR * 256;
    constexpr float WM = (float)(W - 1);
    constexpr unsigned IMG = 2u * S * W * 4u;              // bytes of one image's two stage buffers
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef const __attribute__((address_space(3))) float *lds_fp;
    typedef const __attribute__((address_space(3))) kq_f4 *lds_f4p;
    typedef const __attribute__((address_space(3))) kq_u4 *lds_u4p;
    typedef const __attribute__((address_space(3))) double *lds_dp;
    constexpr unsigned D2_OFF = IMG * (NEAR ? 2u : 1u);
    constexpr unsigned META_OFF = D2_OFF + 3u * S * XB * 4u;
    constexpr unsigned XT_OFF = META_OFF + 3u * S * 16u;
    constexpr unsigned DM_OFF = XT_OFF + XB * 8u;
    constexpr unsigned KM_OFF = DM_OFF + 3u * S * 4u;
    static_assert(KM_OFF + 3u * S * 4u == kg_lds_bytes(W, NEAR), "host and kernel disagree on the LDS layout");
    char *ldsb = reinterpret_cast<char *>(lds);
    float *d2tab = reinterpret_cast<float *>(ldsb + D2_OFF);
    kq_u4 *meta = reinterpret_cast<kq_u4 *>(ldsb + META_OFF);
    double *xtile = reinterpret_cast<double *>(ldsb + XT_OFF);
    float *dmin2tab = reinterpret_cast<float *>(ldsb + DM_OFF);
    int *kmintab = reinterpret_cast<int *>(ldsb + KM_OFF);

    // block -> (chunk, tile): blocks are dealt round robin to the 8 XCDs; groups of G adjacent tiles (their staging
    // streams overlap) stay on one XCD's L2.  Chunk 0 (the shallowest samples: the widest apertures) comes first.
    const int b = blockIdx.x;
    const int xcd = b & 7, rr = b >> 3;
    const int chunk = rr / P.tiles_per_xcd;
    const int qx = rr - chunk * P.tiles_per_xcd;
    const int xt = ((qx / P.G) * 8 + xcd) * P.G + (qx % P.G);
    if (chunk >= P.nchunks || xt >= P.ntiles) return;
    if ((unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)lds != 0u) __builtin_trap();   // raw LDS addresses below

    const int tid = threadIdx.x;
    const int snum = P.snum, tnum = P.tnum;
    const int ti_raw = chunk * 256 + tid;
    const int ti = min(ti_raw, snum - 1);
    const int x0 = P.xlo + xt * XB;
    const int2 jr = P.jr[(size_t)chunk * P.ntiles + xt];
    const int jb = jr.x, jhi = jr.y;
    const int nblocks = jhi >= jb ? (jhi - jb + S) / S : 0;

    constexpr double INV_WM2 = 1.0 / ((double)(W - 1) * (double)(W - 1));
    float a2n = P.a2[ti];                            // a^2 / (W - 1)^2
    float wm_v = WM;
    asm volatile("" : "+v"(a2n), "+v"(wm_v));        // VGPRs, not literals / scalars (an SGPR operand halves the issue rate)
    const double a2s = P.zs2[ti] * (P.cscale * P.cscale);      // a^2 in float64, for the re-decision
    float cnn = P.cn * (float)INV_WM2;                      // near field: r'^2 = (W - 1)^2 r^2
    asm volatile("" : "+v"(cnn));
    float acc[XB];
#pragma unroll
    for (int i = 0; i < XB; ++i) acc[i] = 0.f;
    // this wave's aperture: a trace whose CLOSEST output trace of the tile is further than sqrt(rad2w) samples away
    // from it reaches none of the wave's 64 samples (mig_python.py:52) -- the wave skips its 32 pair steps
    float rad2w;
    {
        const float a_first = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(P.a[ti])));
        const float a_last = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(P.a[ti]), 63));
        const float a2min = (a_first * a_last <= 0.f) ? 0.f : fminf(a_first * a_first, a_last * a_last);
        const float amax = (float)(P.tmax / P.dt);
        rad2w = (amax * amax - a2min) * (1.0f + 1.0e-5f) + 4.0f;
    }

    if (nblocks > 0) {
        // ---- roles for the tables of a block: thread (i, jj) = (tid / 8, tid % 8) computes D^2 of (output x0 + i,
        // trace jj of the block) in float64; threads 0..7 also the slot window of trace jj
        const int ri = tid >> 3, rj = tid & 7;
        const double dxi = P.dist[x0 + ri];          // (dist[] carries 64 copies of its last entry: no clamp needed)
        const double xL = P.dist[min(x0, tnum - 1)], xR = P.dist[min(min(x0 + XB, P.xhi), tnum) - 1];
        const float alo2 = P.alo2[chunk];
        const double u0 = P.tt0 / P.dt;
        auto tables = [&](int blk) {
            const int j = jb + blk * S + rj;
            const bool valid = j <= jhi && j < tnum;
            const double xj = P.dist[min(j, tnum - 1)];
            const double D = (xj - dxi) * P.cscale;
            const int slot3 = blk % 3;
            d2tab[(slot3 * S + rj) * XB + ri] = valid ? (float)(D * D * INV_WM2) : 1.0e30f;
            if (ri == 0) {
                // smallest position any pair of (chunk, tile, j) can have: closest output trace, shallowest sample
                const double dmin = fmax(fmax(xL - xj, xj - xR), 0.0) * P.cscale;
                const float slo = sqrtf((float)(dmin * dmin) + alo2) - (float)u0;
                int kmin = valid ? max((int)floorf(slo) - 2, 0) : 0;
                kmin = min(kmin, max(snum - 2, 0));
                const float cj = (float)(-(u0 + (double)kmin) / (double)WM);
                // e1 (position of the slot's last word): see kirch_launch_gen
                const float thr = 0.5f - (P.e0 + P.e1 * ((float)(kmin + W) + fabsf((float)u0)));
                const unsigned slot_bytes = (unsigned)((blk & 1) * S + rj) * (unsigned)(W * 4);
                const unsigned bias = slot_bytes - (KG_MAGIC_BITS << 2);
                meta[slot3 * S + rj] = kq_u4{bias, __float_as_uint(cj), __float_as_uint(thr), slot_bytes - 4u * (unsigned)kmin};
                dmin2tab[slot3 * S + rj] = valid ? (float)(dmin * dmin) * (1.0f - 1.0e-6f) : 3.0e38f;
                kmintab[slot3 * S + rj] = kmin;
            }
        };
        // ---- staging of a block's 8 traces: thread t holds samples kmin + t + 256 w of every trace
        float pf[S][WR], pd[NEAR ? S : 1][NEAR ? WR : 1];
        const __amdgpu_buffer_rsrc_t gres =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.GT), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t dres =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(NEAR ? P.DT : P.GT), 0, 0x7fffffff, 0x00020000);
        auto stage_load = [&](int blk) {
#pragma unroll
            for (int jj = 0; jj < S; ++jj) {
                const int kmin = kmintab[(blk % 3) * S + jj];
                const int j = min(jb + blk * S + jj, tnum - 1);
                const unsigned rowoff = (unsigned)j * (unsigned)snum * 4u;
#pragma unroll
                for (int w = 0; w < WR; ++w) {
                    const int k = kmin + w * 256 + tid;
                    // zeros from sample snum - 1 on (see the header) and in the slot's last word (what clamped picks read)
                    const bool live = k < snum - 1 && !(w == WR - 1 && tid == 255);
                    const unsigned vo = (unsigned)min(k, snum - 1) * 4u;
                    const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gres, vo, rowoff, 0));
                    pf[jj][w] = live ? v : 0.f;
                    if (NEAR) {
                        const float u = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dres, vo, rowoff, 0));
                        pd[jj][w] = live ? u : 0.f;
                    }
                }
            }
        };
        auto stage_store = [&](int blk) {
#pragma unroll
            for (int jj = 0; jj < S; ++jj)
#pragma unroll
                for (int w = 0; w < WR; ++w) {
                    const int e = ((blk & 1) * S + jj) * W + w * 256 + tid;
                    lds[e] = pf[jj][w];
                    if (NEAR) lds[IMG / 4 + e] = pd[jj][w];
                }
        };

        if (tid < XB) xtile[tid] = P.dist[x0 + tid] * P.cscale;      // positions in samples, float64
        tables(0);
        tables(1);
        __syncthreads();
        stage_load(0);
        stage_store(0);
        __syncthreads();

        for (int blk = 0; blk < nblocks; ++blk) {
            const bool more = blk + 1 < nblocks;
            if (more) stage_load(blk + 1);         // in flight during this block's pairs
            tables(blk + 2);                       // (blocks past the walk: every trace invalid, D^2 = 1e30)
            const int slot3 = blk % 3;
#pragma unroll 1
            for (int jj = 0; jj < S; ++jj) {
                const float dmin2 = *(lds_fp)(uintptr_t)(DM_OFF + (unsigned)(slot3 * S + jj) * 4u);
                if (__builtin_amdgcn_readfirstlane(__float_as_uint(dmin2)) > __float_as_uint(rad2w)) continue;   // (positive floats order like their bits)
                const kq_u4 m = *(lds_u4p)(uintptr_t)(META_OFF + (unsigned)(slot3 * S + jj) * 16u);
                const unsigned bias = m.x, base_k = m.w;                 // base_k = slot base - 4 kmin
                float cj = __uint_as_float(m.y), thr = __uint_as_float(m.z);
                const unsigned d2row = D2_OFF + (unsigned)((slot3 * S + jj) * XB) * 4u;
                const int kmin_j = kmintab[slot3 * S + jj];
                const double xj_s = P.dist[min(jb + blk * S + jj, tnum - 1)] * P.cscale;
                const float kmin_f = (float)kmin_j, smax2_f = (float)(snum - 2);
                float g_prev = 0.f, r_prev = 0.f, gd_prev = 0.f;
                // a flagged pair: within the float32 error of a half-way point.  Decide between the two candidate samples
                // (m, m + 1) in float64, by squares (see the header); the other lanes keep their address.  Out of line:
                // the branch is uniform and rarely taken.
                auto fix = [&](int i, bool flag, float kf, float df, unsigned addr) -> unsigned {
                    if (flag) {
                        // kf counts from the slot's first sample; candidates (m, m + 1), m clamped like argmin's index
                        const float mlo = fminf(fmaxf(kf + (df > 0.f ? kmin_f : kmin_f - 1.0f), 0.f), smax2_f);
                        const double ds = xj_s - *(lds_dp)(uintptr_t)(XT_OFF + 8u * (unsigned)i);
                        const double qs = ds * ds + a2s;                              // mig_python.py:44 in samples^2
                        const double dq = qs - kg_halfway_s2((double)mlo, P);
                        bool up = dq > 0.0;                                           // :49
                        if (__builtin_expect(fabs(dq) <= P.tie2, 0))                  // ... a tie: the reference's own rounding decides
                            up = kg_ref_upper(P.dist, P.zs2, P.tt, P.vel, min(jb + blk * S + jj, tnum - 1), min(x0 + i, tnum - 1), ti, (int)mlo);
                        const float pick = mlo + (up ? 1.0f : 0.0f);
                        addr = base_k + 4u * (unsigned)pick;
                    }
                    return addr;
                };
                auto sum_prev = [&](int ip) {
                    if (NEAR) {
                        const float u = (r_prev * r_prev) * gd_prev;
                        acc[ip] = fmaf(r_prev, fmaf(cnn, u, g_prev), acc[ip]);
                    } else {
                        acc[ip] = fmaf(r_prev, g_prev, acc[ip]);
                    }
                };
#pragma unroll
                for (int iq = 0; iq < XB / 4; ++iq) {
                    const kq_f4 d4 = *(lds_f4p)(uintptr_t)(d2row + iq * 16u);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int i = iq * 4 + c;
                        const float q = a2n + d4[c];
                        const float r = __builtin_amdgcn_rsqf(q);
                        // position in the slot / (W - 1), saturated to [0, 1]: the compiler folds the median into the
                        // fma's clamp modifier (an inline-asm fma here would sit behind v_rsq without the wait state
                        // the transcendental unit needs -- stale r in some lanes)
                        const float pn = __builtin_amdgcn_fmed3f(fmaf(q, r, cj), 0.0f, 1.0f);
                        const float f = fmaf(pn, wm_v, KG_MAGIC);
                        const float kf = f - KG_MAGIC;
                        const float df = fmaf(pn, wm_v, -kf);
                        unsigned addr = (__float_as_uint(f) << 2) + bias;
                        const bool flag = fabsf(df) > thr;
                        if (__builtin_expect(__builtin_amdgcn_ballot_w64(flag) != 0, 0)) addr = fix(i, flag, kf, df, addr);
                        const float g = *(lds_fp)(uintptr_t)addr;
                        float gd = 0.f;
                        if (NEAR) gd = *(lds_fp)(uintptr_t)(addr + IMG);
                        // the previous pair's sum is formed while this pair's read is in flight -- HERE: left alone, the
                        // compiler sinks all 32 sums below the last read and keeps 64 operands live for it (scratch
                        // spills in the re-decision path: three 500-cycle reloads per flagged step)
                        if (i > 0) {
                            sum_prev(i > 0 ? i - 1 : 0);
                            asm volatile("" : "+v"(acc[i > 0 ? i - 1 : 0]));
                        }
                        g_prev = g;
                        gd_prev = gd;
                        r_prev = r;
                    }
                }
                sum_prev(XB - 1);
            }
            if (more) stage_store(blk + 1);
            __syncthreads();
        }
    }

    if (ti_raw < snum) {
        const float a = P.a[ti];
        const float scale = a * P.fin * (1.0f / WM);          // r' = (W - 1) r
        float *o = P.out + (size_t)ti_raw * P.ldo + (x0 - P.xlo);
#pragma unroll
        for (int i = 0; i < XB; ++i)
            if (x0 + i < P.xhi) o[i] = a == 0.f ? 0.f : acc[i] * scale;      // a = 0: cos = 0 or 0/0 (dropped): exactly 0
    }
}

// The pairs that pick the LAST sample (see the header): one thread per output sample, the few candidate traces on
// both sides found by bisection on the sorted dist[], every candidate decided and weighted in float64 in the
// reference's operation order (mig_python.py:44-58), with the same pick rule as the flagged pairs of the main kernel.
template <bool NEAR>
__global__ __launch_bounds__(256) void kirch_gen_shell_kernel(GenParams P)
{
    const int ti = blockIdx.x * 256 + threadIdx.x;
    const int xi = P.xlo + blockIdx.y;
    if (ti >= P.snum || xi >= P.xhi) return;
    const int snum = P.snum, tnum = P.tnum;
    const double z = P.zs[ti], z2 = P.zs2[ti];
    const double qhi = P.r2lim * (1.0 + 1e-9);            // candidates only: the reference's own test decides below
    if (z == 0.0 || !(qhi >= z2)) return;
    // the main kernel's own rule for picking the last sample, in metres^2
    const double cs2 = P.cscale * P.cscale;
    const double qlo = kg_halfway_s2((double)(snum - 2), P) / cs2;
    const double dhi = sqrt(qhi - z2) * (1.0 + 1e-12);
    const double dlo = qlo > z2 ? sqrt(qlo - z2) * (1.0 - 1e-12) : -1.0;
    const double x = P.dist[xi];
    // first j with dist[j] >= v (GT = false) or > v (GT = true; stationary stretches repeat a position).  A guess from
    // the profile's mean spacing, a galloping bracket around it, then bisection inside the bracket: 3-5 loads on a
    // jittered grid instead of the 14 of a bisection over 10000 traces.
    const double d0 = P.dist[0], inv_span = (double)(tnum - 1) / fmax(P.dist[tnum - 1] - d0, 1e-300);
    auto locate = [&](double v, bool gt) {
        auto below = [&](int j) { return gt ? P.dist[j] <= v : P.dist[j] < v; };
        int j = (int)fmin(fmax((v - d0) * inv_span, 0.0), (double)(tnum - 1));
        int lo, hi, step = 1;                 // invariant: everything <= lo is below, everything >= hi is not
        if (below(j)) {
            lo = j;
            hi = j + 1;
            while (hi < tnum && below(hi)) {
                lo = hi;
                step *= 2;
                hi = min(hi + step, tnum);
            }
        } else {
            hi = j;
            lo = j - 1;
            while (lo >= 0 && !below(lo)) {
                hi = lo;
                step *= 2;
                lo = max(lo - step, -1);
            }
        }
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (below(mid)) lo = mid; else hi = mid;
        }
        return hi;
    };
    auto first_ge = [&](double v) { return locate(v, false); };
    auto first_gt = [&](double v) { return locate(v, true); };
    double far = 0.0, near = 0.0;
    auto span = [&](int j0, int j1) {
        for (int j = max(j0, 0); j < min(j1, tnum); ++j) {
            const double dx = P.dist[j] - x;
            const double q = dx * dx + z2;                     // :44
            bool last = q > qlo;                               // picks the last sample (the main kernel's rule ...
            if (kg_near_tie(q * cs2, qlo * cs2, P.tie2)) last = kg_ref_upper(P.dist, P.zs2, P.tt, P.vel, j, xi, ti, snum - 2);      // ... ties included)
            if (!last) continue;
            const double rs = sqrt(q);
            // the time limit in the reference's own arithmetic (:49,:52): in the last row the pair (xi, xi) and every
            // trace at the same position sit ON the limit, and the rounding of 2 rs / vel decides them
            if (2.0 * rs / P.vel > P.tmax) continue;
            const double cost = z / rs;                        // :47
            const size_t o = (size_t)j * snum + (snum - 1);
            far += (double)P.GT[o] * cost / P.vel;             // :53
            if (NEAR) near += (double)P.DT[o] * cost / (rs * rs);   // :58
        }
    };
    if (dlo < 0.0) {
        span(first_ge(x - dhi) - 1, first_gt(x + dhi) + 1);
    } else {
        const int a0 = first_ge(x - dhi) - 1, a1 = first_gt(x - dlo) + 1;
        const int b0 = first_ge(x + dlo) - 1, b1 = first_gt(x + dhi) + 1;
        if (b0 < a1) {
            span(a0, b1);
        } else {
            span(a0, a1);
            span(b0, b1);
        }
    }
    const double add = (far + near) * (1.0 / (2.0 * 3.141592653589793));
    if (add != 0.0) {
        float *o = P.out + (size_t)ti * P.ldo + (xi - P.xlo);
        *o = (float)((double)*o + add);
    }
}

// ===========================================================================
// host side
// ===========================================================================
// kirch_gen_kernel + kirch_gen_shell_kernel (kirch_gen.h) on output traces [xlo, xhi)
int kirch_launch_gen(impdar_kirch_plan *p, void *d_out, int xlo, int xhi, hipStream_t st)
{
    constexpr int XB = 32;
    const int b = p->buf, snum = p->snum, tnum = p->tnum;
    const int ntiles = (xhi - xlo + XB - 1) / XB, nch = p->nchunks;
    const double rlim = p->vel * p->tmax / 2.0;
    if (p->jr_key[0] != xlo || p->jr_key[1] != xhi || p->h_jr.size() != (size_t)nch * ntiles) {
        // input traces within reach of (chunk, tile): the aperture radius of the chunk's shallowest sample around the
        // tile's output traces, one guard trace each side
        const std::vector<double> &d = p->h_dist;
        p->h_jr.assign((size_t)nch * ntiles, make_int2(0, -1));
        for (int c = 0; c < nch; ++c) {
            const double rad2 = rlim * rlim * (1.0 + 1e-9) - p->h_zs2min[c];
            if (rad2 < 0.0) continue;
            const double rad = std::sqrt(rad2) * (1.0 + 1e-12);
            for (int t = 0; t < ntiles; ++t) {
                const int x0 = xlo + t * XB, x1 = std::min(x0 + XB, xhi) - 1;
                const int lo = (int)(std::lower_bound(d.begin(), d.end(), d[x0] - rad) - d.begin()) - 1;
                const int hi = (int)(std::upper_bound(d.begin(), d.end(), d[x1] + rad) - d.begin());
                p->h_jr[(size_t)c * ntiles + t] = make_int2(std::max(lo, 0), std::min(hi, tnum - 1));
            }
        }
        IMPDAR_HIP_CHECK(p->d_jr.ensure(p->h_jr.size() * sizeof(int2)));
        // a blocking copy (tens of KB, only when the block of output traces changes): the next call with another block
        // re-fills h_jr, which an asynchronous copy from that pageable vector might still be reading
        IMPDAR_HIP_CHECK(hipStreamSynchronize(st));
        IMPDAR_HIP_CHECK(hipMemcpy(p->d_jr.p, p->h_jr.data(), p->h_jr.size() * sizeof(int2), hipMemcpyHostToDevice));
        p->jr_key[0] = xlo;
        p->jr_key[1] = xhi;
    }
    GenParams P;
    P.GT = reinterpret_cast<const float *>(img_row0(p, p->GT[b]));
    P.DT = p->nearfield ? reinterpret_cast<const float *>(img_row0(p, p->DT[b])) : nullptr;
    P.out = reinterpret_cast<float *>(d_out);
    P.ldo = xhi - xlo;
    P.snum = snum;
    P.tnum = tnum;
    P.xlo = xlo;
    P.xhi = xhi;
    P.dist = p->d_dist.as<double>();
    P.zs = p->d_zs.as<double>();
    P.zs2 = p->d_zs2.as<double>();
    P.tt = p->d_tt.as<double>();
    P.a = p->d_ga32.as<float>();
    P.a2 = p->d_ga2_32.as<float>();
    P.jr = p->d_jr.as<int2>();
    P.alo2 = p->d_alo2.as<float>();
    P.cscale = 2.0 / (p->vel * p->dt);
    P.r2lim = rlim * rlim;
    P.tmax = p->tmax;
    P.tt0 = p->tt0;
    P.dt = p->dt;
    P.hh = 0.5 + p->tt0 / p->dt;
    P.tie2 = 2.0 * KG_TIE_EPS * ((double)snum + std::fabs(P.hh) + 1.0);
    P.vel = p->vel;
    P.nu0 = (float)(-p->tt0 / p->dt);
    P.fin = (float)(1.0 / (2.0 * M_PI * p->vel));
    const double half = p->vel * p->dt / 2.0;
    P.cn = (float)(p->vel / (half * half));
    // error of the float32 position (kirch_gen_kernel: p (W - 1) = q' r' (W - 1) - u0 - kmin), with u = 2^-24: a^2 / (W-1)^2
    // and D^2 / (W-1)^2 are rounded once each, their sum once (|dq| <= 2 u q, i.e. u s), v_rsq is good to 1 ulp (2 u s),
    // the constant (u0 + kmin) / (W - 1) and the fma round once each (u (u0 + kmin) and < u W): < 4 u s = 2.4e-7 s at
    // the slot's end; e0 covers the 1e-11 dt the time axis may be off a grid
    P.e0 = 2.0e-6f;
    P.e1 = 2.65e-7f;
    P.W = p->genW;
    P.nchunks = nch;
    P.ntiles = ntiles;
    P.G = ntiles >= 256 ? 4 : 1;
    const int per = 8 * P.G;
    P.tiles_per_xcd = ((ntiles + per - 1) / per) * per / 8;
    const int nblk = nch * P.tiles_per_xcd * 8;
    const int wr = p->genW / 256;
    const size_t shmem = kg_lds_bytes(p->genW, p->nearfield != 0);
#define KG_LAUNCH(WR, NEAR)                                                                                      \
    do {                                                                                                          \
        auto k = kirch_gen_kernel<XB, WR, NEAR>;                                                                  \
        IMPDAR_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
        hipLaunchKernelGGL(k, dim3(nblk), dim3(256), shmem, st, P);                                               \
    } while (0)
    if (p->nearfield) {
        if (wr == 2) KG_LAUNCH(2, true); else if (wr == 3) KG_LAUNCH(3, true); else KG_LAUNCH(4, true);
    } else {
        if (wr == 2) KG_LAUNCH(2, false); else if (wr == 3) KG_LAUNCH(3, false); else KG_LAUNCH(4, false);
    }
#undef KG_LAUNCH
    IMPDAR_HIP_CHECK(hipGetLastError());
    const dim3 sg((snum + 255) / 256, xhi - xlo);
    if (p->nearfield)
        hipLaunchKernelGGL(kirch_gen_shell_kernel<true>, sg, dim3(256), 0, st, P);
    else
        hipLaunchKernelGGL(kirch_gen_shell_kernel<false>, sg, dim3(256), 0, st, P);
    IMPDAR_HIP_CHECK(hipGetLastError());
    return IMPDAR_OK;
}

