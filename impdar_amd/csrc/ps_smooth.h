// Phase shift (Gazdag) for velocity profiles that change at (nearly) EVERY depth step -- a firn profile, a linear
// gradient: what getVelocityProfile's 2 * gradient(z(t)) gives for anything but a few thick layers.
// (included by phaseshift.hip)
//
// Reference: mig_python.py:438-487.  Per depth step tau and frequency w (wavenumber kx):
//     coss = 1 - (0.5 v_tau kx / w)^2;   FK[w] *= exp(i w dt sqrt(coss));   FK[w] = 0 for good once coss <= thr_tau;
//     TK[tau] += FK[w]
// The runs kernels (ps_vz32_kernel / ps_vz64_kernel, ps_mfma.h) live on runs of constant velocity, where the rotation
// per step is fixed.  Without runs the per-step kernel paid a float64 divide, square root and a sincos per
// (tau, w): 917 ms (float32) / 834 ms (float64) at 8192^2 against 15 / 60 ms for a four-layer table.
//
// Work units.  ONE WAVE per (wavenumber, chunk of 64 M frequencies), a workgroup of its own: no barrier anywhere.  The
// frequencies below v kx / 2 are evanescent -- 40-55 % of the (kx, w) plane at 1 m / 10 ns / 1.69e8 m/s -- and the few
// next to that edge need an exact, expensive branch at every step (the band, below).  With a workgroup per wavenumber
// (rounds 4a-c) the wave that held the band ran twice as long as its seven neighbours, which waited at the tile's
// barrier with their compute unit: 190 ms where the instruction count said 80.  As waves of their own the all-evanescent
// chunks leave at once, the band's wave takes its time alone, and the hardware fills the slots.  Each wave writes the
// step sums of ITS frequencies to its own row of a partial image [chunk][k][tau]; ps_smooth_sum_kernel adds the
// chunks in order (deterministic) and divides by snum (:492).  9 GB of extra traffic at 8192^2: 2.5 ms.
//
// float64 data (ps_smooth_kernel): every per-step quantity is carried forward instead of recomputed:
//   * y = sqrt(coss) by three Newton steps from the previous step's y, with g = 1 / (2 y) carried by its own Newton
//     step between them: 10 float64 fma, no divide, no square root;
//   * the per-step rotation R = exp(i phi), phi = w dt y, by R *= exp(i d), d = phi - phi_previous, cos to d^6 and sin
//     to d^5: a rotation OF the rotation, no sincos.
// The band.  Newton from the previous value needs coss to move little RELATIVE to itself: rho = |d coss| / (2 coss).
// A frequency about to turn evanescent (coss -> 0) violates that, and so does every frequency at a step where the
// velocity jumps.  Lanes with coss < 60 |d(v^2)| / v^2 (all lanes at step 0) take the exact path for that step: coss
// in the reference's own rounding where it is within 1e-13 of the threshold (the evanescence test coss <= thr is
// decided there and only there), a real square root, a real sincos -- or, for narrow bands, the hardware's reciprocal
// root + Newton and the series of sine and cosine (the phase is small where coss is).
// float32 data (ps_smooth32_kernel): see there.
#pragma once

#define PSS_BAND 60.0
#define PSS_NARROW 0.01     // bands up to this take the short way through the exact branch
#define PSS_TT 16           // steps per tile


// sin / cos of a float64 angle in [-pi/2, 3 pi / 2] (the per-step phase w dt sqrt(coss) lies in [0, pi]): quadrant by
// Cody-Waite with pi/2 in two pieces, the kernels of fdlibm's __kernel_sin / __kernel_cos on [-pi/4, pi/4] (< 1 ulp).
// The library's sincos carries a large-argument path (a table in scratch memory): in ps_smooth_kernel's anchors that cost
// the 8-frequencies-per-lane kernel 372 bytes of scratch per lane and made it 60 % slower.
__device__ __forceinline__ void pss_sincos_small(double x, double *s, double *c)
{
    const double n = rint(x * 0.63661977236758134308);
    double r = fma(-n, 1.57079632673412561417e+00, x);
    r = fma(-n, 6.07710050650619224932e-11, r);
    r = fma(-n, 2.02226624879595063154e-21, r);
    const double z = r * r;
    const double sp = fma(z * r, fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                                        2.75573137070700676789e-06), -1.98412698298579493134e-04),
                                           8.33333333332248946124e-03), -1.66666666666666324348e-01), r);
    const double cp = fma(z * z, fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                                         -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                                            -1.38888888888741095749e-03), 4.16666666666666019037e-02), fma(-0.5, z, 1.0));
    const int q = (int)n;
    const bool odd = q & 1;
    const double sv = odd ? cp : sp, cv = odd ? sp : cp;
    *s = (q & 2) ? -sv : sv;
    *c = ((q + 1) & 2) ? -cv : cv;
}

// what a wave of either kernel starts with
struct PssWave {
    int k, chunk, lane;
    double kxk;
};

// [chunk][k][tau] partial images -> TK = sum over chunks / snum (:492); n = nk * snum elements
template <typename T> __global__ void ps_smooth_sum_kernel(const Cp<T> *__restrict__ part, Cp<T> *__restrict__ TK, int nchunks, size_t n, int snum)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Cp<T> s = part[i];
    for (int c = 1; c < nchunks; ++c) {
        const Cp<T> v = part[(size_t)c * n + i];
        s.x += v.x;
        s.y += v.y;
    }
    s.x = s.x / (T)snum;
    s.y = s.y / (T)snum;
    TK[i] = s;
}

// the tile's 16 step sums of this lane's frequencies (acc[2 t], acc[2 t + 1]) -> summed over the wave -> row `out`
template <typename T> __device__ __forceinline__ void pss_write_tile(T (&acc)[2 * PSS_TT], int lane, Cp<T> *out, int tau0, int snum, bool final_scale)
{
    wave_reduce_scatter<T, 2 * PSS_TT>(acc, lane);        // lane 2 i holds value i
    const int tau = tau0 + (lane >> 2);
    if ((lane & 1) == 0 && tau < snum) {
        T *dst = reinterpret_cast<T *>(out + tau) + ((lane >> 1) & 1);
        *dst = final_scale ? acc[0] / (T)snum : acc[0];
    }
}

// NMEM = 2: the wave takes the wavenumbers kx and -kx (rows k and tnum - k) together: y, g, the band and the rotation R depend
// on kx^2 only -- 26 of the ~32 instructions per (step, frequency) -- the state's rotation and the sums are per wavenumber.
template <int M, int NMEM>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void ps_smooth_kernel(PsParams P)
{
    using T = double;
    constexpr int TT = PSS_TT;
    // frequencies per group: their chains are independent (a wave that issues one dependent float64 operation after the
    // other waits ~16 cycles for each), ONE branch per group for the band, and a group whose frequencies have turned
    // evanescent in all 64 lanes is skipped
    constexpr int G = M < 2 ? M : 2;
    __shared__ Cp<T> part[NMEM][TT][64];                  // per-step partial sums of a lane's frequencies
    __shared__ double4 stepc[2][TT];                      // the steps' {c = v^2, band, threshold, v}
    const int chunk = gridDim.y - 1 - blockIdx.y;         // (the high, busy chunks first)
    const int lane = threadIdx.x;
    int km[NMEM];                                         // rows of the wavenumber axis
    km[0] = P.k0 + blockIdx.x;
    if (NMEM == 2) km[NMEM - 1] = (P.tnum - (int)blockIdx.x) % P.tnum;
    const bool has_b = NMEM == 2 && km[NMEM - 1] != km[0];    // (k = 0 and the Nyquist row are their own partners)
    const int k = km[0];
    const Cp<T> *F[NMEM];
    Cp<T> *out[NMEM];
#pragma unroll
    for (int mm = 0; mm < NMEM; ++mm) {
        F[mm] = reinterpret_cast<const Cp<T> *>(P.F) + (size_t)km[mm] * P.fstride;
        out[mm] = reinterpret_cast<Cp<T> *>(P.sm_nchunks > 1 ? P.sm_part : P.TK) +
                  ((size_t)(P.sm_nchunks > 1 ? chunk : 0) * P.nk + (km[mm] - P.k0)) * P.snum;
    }
    const bool final_scale = P.sm_nchunks == 1;
    const double kxk = P.kx[k];

    // per owned frequency: x = (kx / 2w)^2, w dt, y = sqrt(coss) and g ~ 1 / (2 y) of the last step
    double x[M], wdt[M], y[M], g[M];
    T sr_[NMEM][M], si_[NMEM][M], rc[M], rs[M];           // states FK and rotation R = exp(i phi)
    unsigned dead = 0;                                    // bit m: frequency m of this lane has turned evanescent
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int slot = (chunk * M + m) * 64 + lane;
        Cp<T> f[NMEM];
#pragma unroll
        for (int mm = 0; mm < NMEM; ++mm) f[mm].x = f[mm].y = 0;
        double w = 1.0;
        if (slot < P.nf) {
            f[0] = ps_load_slot<T>(F[0], P, slot);
            if (NMEM == 2 && has_b) f[NMEM - 1] = ps_load_slot<T>(F[NMEM - 1], P, slot);
            w = P.w[slot];
        } else {
            dead |= 1u << m;
        }
        const double a0 = 0.5 * kxk / w;
        x[m] = slot < P.nf ? a0 * a0 : 0.0;
        wdt[m] = w * P.dt;
        y[m] = 1.0;
        g[m] = 0.5;
#pragma unroll
        for (int mm = 0; mm < NMEM; ++mm) {
            sr_[mm][m] = f[mm].x;
            si_[mm][m] = f[mm].y;
        }
        rc[m] = 1;
        rs[m] = 0;
    }
    auto step_constants = [&](int tile) {                 // lanes 0..15: the constants of tile `tile`
        const int tau = tile * TT + lane;
        if (lane < TT && tau < P.snum) {
            const double vd = P.vz[tau], c = vd * vd;
            double csb = 4.0;                             // step 0: every lane from scratch
            if (tau > 0) {
                const double vp = P.vz[tau - 1];
                // band: coss below this moved too much relative to itself for the carried values
                csb = PSS_BAND * fabs(c - vp * vp) / c + 1.0e-9;
            }
            stepc[tile & 1][lane] = make_double4(c, csb, P.thr[tau], vd);
        }
    };
    step_constants(0);
    const int ntile = (P.snum + TT - 1) / TT;
    for (int tile = 0; tile < ntile; ++tile) {
        const int tau0 = tile * TT;
        __syncthreads();                                  // (one wave: orders the table's writes and reads)
        step_constants(tile + 1);
        // groups all of whose frequencies, in all 64 lanes, are out: nothing to carry, nothing to add
        unsigned gdead = 0;
        bool all_out = true;
#pragma unroll
        for (int m0 = 0; m0 < M; m0 += G) {
            bool alive = false;
#pragma unroll
            for (int j = 0; j < G; ++j) alive = alive || !((dead >> (m0 + j)) & 1u);
            if (__builtin_amdgcn_ballot_w64(alive) == 0) gdead |= 1u << m0;
            else all_out = false;
        }
        if (all_out) {                                    // uniform: zeros from here to the end of the record
            Cp<T> z;
            z.x = z.y = 0;
#pragma unroll
            for (int mm = 0; mm < NMEM; ++mm)
                if (mm == 0 || has_b)
                    for (int tau = tau0 + lane; tau < P.snum; tau += 64) out[mm][tau] = z;
            return;
        }
#pragma unroll 1
        for (int t = 0; t < TT; ++t) {
            const int tau = tau0 + t;
            T psr[NMEM], psi[NMEM];
#pragma unroll
            for (int mm = 0; mm < NMEM; ++mm) psr[mm] = psi[mm] = 0;
            if (tau < P.snum) {                                   // uniform
                const double4 sc = stepc[tile & 1][t];
                const double c = sc.x, csb = sc.y;
#define PSS_EACH for (int j = 0, m = m0; j < G; ++j, ++m)
#define PSS_PIN(f)                                                                   \
    do {                                                                             \
        if constexpr (G == 2) asm volatile("" : "+v"(q[0].f), "+v"(q[G > 1 ? 1 : 0].f)); \
        else asm volatile("" : "+v"(q[0].f));                                        \
    } while (0)
#pragma unroll
                for (int m0 = 0; m0 < M; m0 += G) {
                    if ((gdead >> m0) & 1u) continue;             // uniform
                    struct {
                        double cs, e1, y1, r, e2, y2, gn, dd;
                        T d, hd2, ur, ui, m1, m2, ncr, nsr;
                        bool band;
                    } q[G];
                    bool any = false;
                    // stage by stage across the group (the pins keep the scheduler from putting the chains back one after
                    // the other to save registers)
#pragma unroll
                    PSS_EACH q[j].cs = fma(-c, x[m], 1.0);
                    PSS_PIN(cs);
                    // Newton steps for y = sqrt(cs) from the previous step's y, g ~ 1 / (2 y).  (g is refreshed BETWEEN
                    // the steps: with the previous step's g in both, the second step only gains a factor rho -- rho^3
                    // left, 7e-8 on an image of 700 steps; refreshed from y1 it is a true Newton step: rho^4)
#pragma unroll
                    PSS_EACH q[j].e1 = fma(-y[m], y[m], q[j].cs);
                    PSS_PIN(e1);
#pragma unroll
                    PSS_EACH q[j].y1 = fma(q[j].e1, g[m], y[m]);
                    PSS_PIN(y1);
#pragma unroll
                    PSS_EACH {
                        q[j].r = fma(-(q[j].y1 + q[j].y1), g[m], 1.0);
                        q[j].e2 = fma(-q[j].y1, q[j].y1, q[j].cs);
                    }
                    PSS_PIN(r);
                    PSS_PIN(e2);
#pragma unroll
                    PSS_EACH q[j].gn = fma(g[m], q[j].r, g[m]);
                    PSS_PIN(gn);
#pragma unroll
                    PSS_EACH q[j].y2 = fma(q[j].e2, q[j].gn, q[j].y1);
                    PSS_PIN(y2);
                    // a third step (rho^4 = 2e-12 per step at the band's edge adds up over a record)
#pragma unroll
                    PSS_EACH q[j].e1 = fma(-q[j].y2, q[j].y2, q[j].cs);
                    PSS_PIN(e1);
#pragma unroll
                    PSS_EACH q[j].y2 = fma(q[j].e1, q[j].gn, q[j].y2);
                    PSS_PIN(y2);
                    // R *= exp(i d), d = phi - phi_previous = w dt (y2 - y): cos to d^6, sin to d^5 (with
                    // 1 - d^2/2 + i (d - d^3/6) the modulus is 1 - d^4/24: a systematic loss, 6e-9 over 700 steps; at the
                    // band's edge d reaches 0.03)
#pragma unroll
                    PSS_EACH q[j].dd = q[j].y2 - y[m];
                    PSS_PIN(dd);
#pragma unroll
                    PSS_EACH q[j].d = q[j].dd * wdt[m];
                    PSS_PIN(d);
#pragma unroll
                    PSS_EACH q[j].hd2 = q[j].d * q[j].d * 0.5;
                    PSS_PIN(hd2);
#pragma unroll
                    PSS_EACH {
                        // with h = d^2 / 2:  cos d = 1 - h + h^2/6 - h^3/90,  sin d = d (1 - h/3 + h^2/30)
                        const T h = q[j].hd2;
                        q[j].ur = fma(h, fma(h, fma(h, -1.0 / 90.0, 1.0 / 6.0), -1.0), 1.0);
                        q[j].ui = q[j].d * fma(h, fma(h, 1.0 / 30.0, -1.0 / 3.0), 1.0);
                    }
                    PSS_PIN(ur);
#pragma unroll
                    PSS_EACH {
                        q[j].m1 = rs[m] * q[j].ui;
                        q[j].m2 = rs[m] * q[j].ur;
                    }
                    PSS_PIN(m1);
                    PSS_PIN(m2);
#pragma unroll
                    PSS_EACH {
                        q[j].ncr = fma(rc[m], q[j].ur, -q[j].m1);
                        q[j].nsr = fma(rc[m], q[j].ui, q[j].m2);
                        q[j].band = q[j].cs < csb;
                        any = any || q[j].band;
                    }
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(any) != 0, 0)) {
                        const double thr = sc.z, vd = sc.w;
                        const bool narrow = csb <= PSS_NARROW;    // uniform
#pragma unroll
                        PSS_EACH {
                            if (q[j].band) {
                                const int slot = (chunk * M + m) * 64 + lane;
                                double cr = q[j].cs;
                                const bool cheap = narrow && fabs(cr - thr) > 1.0e-13;
                                if (!cheap) {
                                    // coss in the reference's own rounding (:456-460): what decides at the threshold
                                    const double wx = slot < P.nf ? P.w[slot] : 1.0;
                                    const double a = ((0.5 * vd) * kxk) / wx;
                                    cr = 1.0 - a * a;
                                }
                                if (cr <= thr || ((dead >> m) & 1u)) {
                                    // evanescent: zero from here on (:484-485 zero the spectrum itself: once out, out for
                                    // good -- a velocity that falls again must not revive the carried values of such a
                                    // lane, which are parked where they stay finite)
                                    dead |= 1u << m;
#pragma unroll
                                    for (int mm = 0; mm < NMEM; ++mm) sr_[mm][m] = si_[mm][m] = 0;
                                    x[m] = 0.0;
                                    q[j].y2 = 1.0;
                                    q[j].gn = 0.5;
                                    q[j].ncr = 1;
                                    q[j].nsr = 0;
                                } else if (cheap) {
                                    // y = sqrt(coss) from the hardware's reciprocal square root and two Newton steps that
                                    // carry h ~ 1 / (2 y) along (coss in [1e-13, 0.01]: no scaling); the phase w dt y is
                                    // below 0.32: sine and cosine from their series
                                    const double r0 = __builtin_amdgcn_rsq(cr);
                                    double y0 = cr * r0, h = 0.5 * r0;
#pragma unroll
                                    for (int it = 0; it < 2; ++it) {
                                        y0 = fma(fma(-y0, y0, cr), h, y0);
                                        h = fma(h, fma(-(y0 + y0), h, 1.0), h);
                                    }
                                    y0 = fma(fma(-y0, y0, cr), h, y0);
                                    q[j].y2 = y0;
                                    q[j].gn = h;
                                    const T ph = wdt[m] * y0, s2 = ph * ph;
                                    T pc = 1.0 / 87178291200.0;                          // 1/14!
                                    T ps = 1.0 / 1307674368000.0;                        // 1/15!
                                    pc = fma(s2, -pc, 1.0 / 479001600.0);                // 1/12!
                                    ps = fma(s2, -ps, 1.0 / 6227020800.0);               // 1/13!
                                    pc = fma(s2, -pc, 1.0 / 3628800.0);
                                    ps = fma(s2, -ps, 1.0 / 39916800.0);
                                    pc = fma(s2, -pc, 1.0 / 40320.0);
                                    ps = fma(s2, -ps, 1.0 / 362880.0);
                                    pc = fma(s2, -pc, 1.0 / 720.0);
                                    ps = fma(s2, -ps, 1.0 / 5040.0);
                                    pc = fma(s2, -pc, 1.0 / 24.0);
                                    ps = fma(s2, -ps, 1.0 / 120.0);
                                    pc = fma(s2, -pc, 0.5);
                                    ps = fma(s2, -ps, 1.0 / 6.0);
                                    pc = fma(s2, -pc, 1.0);
                                    ps = fma(s2, -ps, 1.0);
                                    q[j].ncr = pc;
                                    q[j].nsr = ps * ph;
                                } else {
                                    // this step from scratch (:456-464)
                                    q[j].y2 = sqrt(cr);
                                    q[j].gn = 0.5 / q[j].y2;
                                    T sn, cn;
                                    sincos_t<T>(wdt[m] * q[j].y2, &sn, &cn);
                                    q[j].ncr = cn;
                                    q[j].nsr = sn;
                                }
                            }
                        }
                    }
#pragma unroll
                    PSS_EACH {
                        y[m] = q[j].y2;
                        g[m] = q[j].gn;
                        rc[m] = q[j].ncr;
                        rs[m] = q[j].nsr;
#pragma unroll
                        for (int mm = 0; mm < NMEM; ++mm) {
                            const T nr = fma(sr_[mm][m], q[j].ncr, -(si_[mm][m] * q[j].nsr));     // FK *= exp(i phi), :464
                            const T ni = fma(sr_[mm][m], q[j].nsr, si_[mm][m] * q[j].ncr);
                            sr_[mm][m] = nr;
                            si_[mm][m] = ni;
                            psr[mm] += nr;                                          // :487
                            psi[mm] += ni;
                        }
                    }
#pragma unroll
                    for (int mm = 0; mm < NMEM; ++mm) asm volatile("" : "+v"(psr[mm]), "+v"(psi[mm]));
                }
#undef PSS_PIN
#undef PSS_EACH
            }
#pragma unroll
            for (int mm = 0; mm < NMEM; ++mm) {
                Cp<T> pv;
                pv.x = psr[mm];
                pv.y = psi[mm];
                part[mm][t][lane] = pv;                   // (each lane reads back only what it wrote)
            }
        }
#pragma unroll
        for (int mm = 0; mm < NMEM; ++mm) {
            if (mm == 1 && !has_b) break;                 // uniform
            T acc[2 * TT];
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const Cp<T> pv = part[mm][t][lane];
                acc[2 * t] = pv.x;
                acc[2 * t + 1] = pv.y;
            }
            pss_write_tile<T>(acc, lane, out[mm], tau0, P.snum, final_scale);
        }
        // R itself is re-anchored every 128 steps: R *= exp(i d) rounds once per step (1e-16), so R's phase is off by
        // ~tau 1e-16 after tau steps, and the state, which integrates R, by ~tau^2 / 2 of that: 4.4e-10 of the image
        // maximum at 8192 steps (round 5: the 8192^2 spot-wavenumber test; 4e-12 with the anchors).  Here, between two
        // tiles, not inside the step loop: there the extra live values pushed the 8-frequencies-per-lane kernel further
        // into scratch (220 -> 377 ms at 8192^2).
        if ((tile & 7) == 7) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                if (!((dead >> m) & 1u)) {
                    double sn, cn;
                    pss_sincos_small(wdt[m] * y[m], &sn, &cn);
                    rc[m] = cn;
                    rs[m] = sn;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// float32 data: expansion about one velocity per 16-step tile.
//
// The kernel above spends 14 of its ~32 instructions per (step, frequency) to carry y = sqrt(coss) in float64.  A float32
// image needs float64 only where errors ADD UP: in the phase sum Phi that the anchors rotate the original spectrum by.
// Both are taken from an expansion about the middle velocity of 32 steps (two tiles), c_a = (min + max)/2 of c = v^2 over them,
// delta_t = c_t - c_a, per frequency xi = x / coss_a, u_t = xi delta_t:
//     y_t = y_a sqrt(1 - u_t)
//   * the phase sum of the whole tile at once, in float64, at the tile's start:
//         sum_t y_t = y_a sum_p b_p xi^p S_p,   S_p = sum_t delta_t^p,  b_p the binomial series of sqrt(1 - u)
//     -- S_p is the same for every frequency and wavenumber: the HOST folds Q_p = b_p S_p into a table
//     (ps_smooth_tables), the kernel runs a 9-term Horner in xi per frequency and tile (|u| <= 0.1: 1e-11 left);
//   * the step's rotation in float32: exp(i phi_t) = R_a exp(i psi_t), R_a = exp(i w dt y_a),
//     psi_t = w dt y_a (sqrt(1 - u_t) - 1) from the hardware square root (absolute error 1e-7 w dt y_a: it does not add
//     up, every step starts again from R_a), |psi| < 0.15: its sine and cosine from the series.  19 float32 instructions
//     and the square root per (step, frequency); per (tile, frequency) ~40 float64 ones for y_a (hardware reciprocal root +
//     two Newton steps), xi, the Horner, and the turn of R_a to the new tile's velocity.
// Frequencies with |u| > 0.1 somewhere in the tile, or within reach of the threshold, are the band of this kernel:
// they take the exact branch for all steps of the tile.  State anchors every 64 steps.
#define PSS32_UMAX 0.1
#define PSS32_NQ 9              // Q_0 .. Q_8
#define PSS32_NARROW 0.02       // band lanes with coss up to this: phase < 0.45, sine and cosine from their series
#define PSS32_TILE_DOUBLES 16   // per expansion: Q_0..Q_8, [9] c_a, [10] max |delta|, [11] max threshold, [12] narrow
#define PSS32_XTILES 2          // 16-step tiles per expansion (one per tile: 15 % of the kernel's instructions went into it)

// the per-step / per-tile tables of ps_smooth32_kernel (host, float64): step[4 tau + {0,1,2,3}] = {c, thr, v, delta}
static void ps_smooth_tables(const double *vz, const double *thr, int snum, std::vector<double> &step, std::vector<double> &tile)
{
    constexpr int XS = PSS32_XTILES * PSS_TT;              // steps per expansion
    const int nint = (snum + XS - 1) / XS;
    step.assign((size_t)nint * XS * 4, 0.0);
    tile.assign((size_t)nint * PSS32_TILE_DOUBLES, 0.0);
    for (int t = 0; t < nint; ++t) {
        const int t0 = t * XS, n = std::min(XS, snum - t0);
        double cmin = vz[t0] * vz[t0], cmax = cmin, thrmax = thr[t0];
        for (int i = 1; i < n; ++i) {
            const double c = vz[t0 + i] * vz[t0 + i];
            cmin = std::min(cmin, c);
            cmax = std::max(cmax, c);
            thrmax = std::max(thrmax, thr[t0 + i]);
        }
        const double ca = 0.5 * (cmin + cmax);
        double *q = &tile[(size_t)t * PSS32_TILE_DOUBLES];
        double dmax = 0.0, pw[XS];
        for (int i = 0; i < XS; ++i) {
            const int tau = t0 + std::min(i, n - 1);
            const double c = vz[tau] * vz[tau];
            double *sp = &step[(size_t)(t0 + i) * 4];
            sp[0] = c;
            sp[1] = thr[tau];
            sp[2] = vz[tau];
            sp[3] = c - ca;
            if (i < n) dmax = std::max(dmax, std::fabs(c - ca));
            pw[i] = 1.0;
        }
        double b = 1.0;
        for (int p = 0; p < PSS32_NQ; ++p) {
            if (p == 1) b = -0.5;
            if (p >= 2) b *= (2.0 * p - 3.0) / (2.0 * p);
            double sum = 0.0;
            for (int i = 0; i < n; ++i) {
                sum += pw[i];
                pw[i] *= step[(size_t)(t0 + i) * 4 + 3];
            }
            q[p] = b * sum;
        }
        q[9] = ca;
        q[10] = dmax;
        q[11] = thrmax;
        // the band's lanes: coss_a < x dmax / UMAX (+ the threshold's reach), coss_t <= coss_a + x dmax, x <= 1 / cmin while
        // the lane propagates
        q[12] = ((1.0 / PSS32_UMAX + 1.0) * dmax / cmin + 2.0 * thrmax + 1e-12 <= PSS32_NARROW) ? 1.0 : 0.0;
    }
}

// four uniform doubles through the scalar cache (load and wait in one statement, see scalar_load_2f64)
__device__ __forceinline__ void pss_scalar_load_4f64(const double *p, double *a, double *b, double *c, double *d)
{
    double x, y, z, w;
    asm volatile("s_load_dwordx2 %0, %4, 0x0\n\ts_load_dwordx2 %1, %4, 0x8\n\ts_load_dwordx2 %2, %4, 0x10\n\t"
                 "s_load_dwordx2 %3, %4, 0x18\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(x), "=&s"(y), "=&s"(z), "=&s"(w)
                 : "s"(p)
                 : "memory");
    *a = x;
    *b = y;
    *c = z;
    *d = w;
}

// NMEM = 2: the wave takes the wavenumbers kx and -kx (rows k and tnum - k) together -- coss, the expansion, the band and every
// step's rotation depend on kx^2 only (:456-460): 13 of the 19 instructions per (step, frequency) are shared, the state's
// rotation and the sums (6) are per wavenumber.  blockIdx.x = the pair.
template <int M, int NMEM>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void ps_smooth32_kernel(PsParams P)
{
    using T = float;
    constexpr int TT = PSS_TT;
    constexpr int ANCHOR_TILES = 4;
#ifndef PSS32_G
#define PSS32_G 4      // (2: 89.7 ms, 4: 87.4 ms at 8192^2 with two wavenumbers per wave)
#endif
    constexpr int G = M < PSS32_G ? M : PSS32_G;          // frequencies whose chains are interleaved, stage by stage
    __shared__ Cp<T> f0_lds[NMEM][M][64];                 // original spectra (the anchors rotate them)
    __shared__ Cp<T> part[NMEM][TT][64];                  // per-step partial sums of a lane's frequencies
    __shared__ double4 stepc[2][TT];                      // the steps' {c, thr, v, delta}
    const int chunk = gridDim.y - 1 - blockIdx.y;         // (the high, busy chunks first)
    const int lane = threadIdx.x;
    int km[NMEM];                                         // rows of the wavenumber axis
    km[0] = P.k0 + blockIdx.x;
    if (NMEM == 2) km[NMEM - 1] = (P.tnum - (int)blockIdx.x) % P.tnum;
    const bool has_b = NMEM == 2 && km[NMEM - 1] != km[0];    // (k = 0 and the Nyquist row are their own partners)
    const int k = km[0];
    const Cp<T> *F[NMEM];
    Cp<T> *out[NMEM];
#pragma unroll
    for (int mm = 0; mm < NMEM; ++mm) {
        F[mm] = reinterpret_cast<const Cp<T> *>(P.F) + (size_t)km[mm] * P.fstride;
        out[mm] = reinterpret_cast<Cp<T> *>(P.sm_nchunks > 1 ? P.sm_part : P.TK) +
                  ((size_t)(P.sm_nchunks > 1 ? chunk : 0) * P.nk + (km[mm] - P.k0)) * P.snum;
    }
    const bool final_scale = P.sm_nchunks == 1;
    const double kxk = P.kx[k];
    const double4 *gstep = reinterpret_cast<const double4 *>(P.sm_step);

    double x[M], wdt[M], Phi[M], Kp[M];                   // (kx / 2w)^2, w dt, phase at the END of the tile, w dt y_a
    T sr_[NMEM][M], si_[NMEM][M], ca[M], sa[M], xi[M], Kf[M];     // states FK, R_a = exp(i w dt y_a), x / coss_a, w dt y_a
    unsigned dead = 0;                                    // bit m: frequency m of this lane has turned evanescent
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int slot = (chunk * M + m) * 64 + lane;
        Cp<T> f[NMEM];
#pragma unroll
        for (int mm = 0; mm < NMEM; ++mm) f[mm].x = f[mm].y = 0;
        double w = 1.0;
        if (slot < P.nf) {
            f[0] = ps_load_slot<T>(F[0], P, slot);
            if (NMEM == 2 && has_b) f[NMEM - 1] = ps_load_slot<T>(F[NMEM - 1], P, slot);
            w = P.w[slot];
        } else {
            dead |= 1u << m;
        }
        const double a0 = 0.5 * kxk / w;
        x[m] = slot < P.nf ? a0 * a0 : 0.0;
        wdt[m] = w * P.dt;
        Phi[m] = 0.0;
        Kp[m] = 0.0;
#pragma unroll
        for (int mm = 0; mm < NMEM; ++mm) {
            f0_lds[mm][m][lane] = f[mm];
            sr_[mm][m] = f[mm].x;
            si_[mm][m] = f[mm].y;
        }
        ca[m] = 1;
        sa[m] = 0;
        xi[m] = 0;
        Kf[m] = 0;
    }
    if (lane < TT) stepc[0][lane] = gstep[lane];
    const int ntile = (P.snum + TT - 1) / TT;
    static_assert(ANCHOR_TILES % PSS32_XTILES == 0, "anchors fall on expansion starts");
    unsigned bandm = 0;                                   // bit m: frequency m of this lane takes the exact branch this expansion
    unsigned gband = 0;                                   // (uniform) bit m0: the group has a lane in the band
    bool narrow = false;
    for (int tile = 0; tile < ntile; ++tile) {
        const int tau0 = tile * TT;
        const bool anchor = tile % ANCHOR_TILES == 0;
        const bool expand = tile % PSS32_XTILES == 0;
        __syncthreads();                                  // (one wave: orders the table's writes and reads)
        if (lane < TT && tile + 1 < ntile) stepc[(tile + 1) & 1][lane] = gstep[(size_t)(tile + 1) * TT + lane];
        if (anchor && tile > 0) {
            // the state from the ORIGINAL spectrum and the float64 phase (which stands at the end of the last tile)
#pragma unroll
            for (int m = 0; m < M; ++m) {
                double ph = Phi[m];
                ph -= 6.283185307179586 * rint(ph * 0.15915494309189535);
                Phi[m] = ph;
                T sn, cs;
                sincos_t<T>((T)ph, &sn, &cs);
#pragma unroll
                for (int mm = 0; mm < NMEM; ++mm) {
                    const Cp<T> f0 = f0_lds[mm][m][lane];
                    sr_[mm][m] = fma(f0.x, cs, -(f0.y * sn));
                    si_[mm][m] = fma(f0.x, sn, f0.y * cs);
                    asm volatile("" : "+v"(sr_[mm][m]), "+v"(si_[mm][m]));
                }
            }
        }
        // groups all of whose frequencies, in all 64 lanes, are out
        unsigned gdead = 0;
        bool all_out = true;
#pragma unroll
        for (int m0 = 0; m0 < M; m0 += G) {
            bool alive = false;
#pragma unroll
            for (int j = 0; j < G; ++j) alive = alive || !((dead >> (m0 + j)) & 1u);
            if (__builtin_amdgcn_ballot_w64(alive) == 0) gdead |= 1u << m0;
            else all_out = false;
        }
        gdead = __builtin_amdgcn_readfirstlane(gdead);
        // ---- the expansion of the next PSS32_XTILES tiles
        if (expand && !all_out) {                         // uniform
            double tq[PSS32_TILE_DOUBLES];
            {
                const double *gt = P.sm_tile + (size_t)(tile / PSS32_XTILES) * PSS32_TILE_DOUBLES;
#pragma unroll
                for (int i = 0; i < 12; i += 4) pss_scalar_load_4f64(gt + i, &tq[i], &tq[i + 1], &tq[i + 2], &tq[i + 3]);
                tq[12] = gt[12];
            }
            const double c_a = tq[9], dmax = tq[10], thrmax = tq[11];
            narrow = tq[12] != 0.0;
            bandm = 0;
            gband = 0;
#pragma unroll
            for (int m0 = 0; m0 < M; m0 += G) {
                if ((gdead >> m0) & 1u) continue;             // uniform
                bool anyband = false;
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const int m = m0 + j;
                    const double cs_a = fma(-c_a, x[m], 1.0);
                    const bool band = !(cs_a * PSS32_UMAX > x[m] * dmax) || !(cs_a * 0.75 > thrmax + 1.0e-13);
                    if (band) bandm |= 1u << m;
                    anyband = anyband || band;
                    const double csu = band ? 1.0 : cs_a;     // (band lanes: any finite stand-in, they overwrite what it gives)
                    const double r0 = __builtin_amdgcn_rsq(csu);
                    double y0 = csu * r0, h = 0.5 * r0;
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        y0 = fma(fma(-y0, y0, csu), h, y0);
                        h = fma(h, fma(-(y0 + y0), h, 1.0), h);
                    }
                    y0 = fma(fma(-y0, y0, csu), h, y0);
                    const double t2 = h + h, xid = t2 * t2 * x[m];     // x / coss_a
                    double H = tq[PSS32_NQ - 1];
#pragma unroll
                    for (int p = PSS32_NQ - 2; p >= 0; --p) H = fma(H, xid, tq[p]);
                    const double K = wdt[m] * y0;
                    if (!band) Phi[m] = fma(K, H, Phi[m]);     // the whole expansion's phase (band lanes add theirs step by step)
                    xi[m] = (T)xid;
                    const T K32 = (T)K;
                    Kf[m] = K32;
                    // R_a: turned from the last expansion's by the series in dK, from scratch at the anchors and after jumps
                    const T dK = (T)(K - Kp[m]);
                    Kp[m] = K;
                    if (anchor || !(fabsf(dK) <= 0.2f)) {
                        const T klo = (T)(K - (double)K32);
                        T sn, cn;
                        sincos_t<T>(K32, &sn, &cn);
                        ca[m] = fma(-sn, klo, cn);
                        sa[m] = fma(cn, klo, sn);
                    } else {
                        const T s2 = dK * dK;
                        const T ec = fma(s2, fma(s2, fma(s2, (T)(-1.0 / 720), (T)(1.0 / 24)), (T)-0.5), (T)1);
                        const T es = dK * fma(s2, fma(s2, (T)(1.0 / 120), (T)(-1.0 / 6)), (T)1);
                        const T nc = fma(ca[m], ec, -(sa[m] * es));
                        const T ns = fma(ca[m], es, sa[m] * ec);
                        ca[m] = nc;
                        sa[m] = ns;
                    }
                }
                if (__builtin_amdgcn_ballot_w64(anyband) != 0) gband |= 1u << m0;
            }
            gband = __builtin_amdgcn_readfirstlane(gband);
        }
        if (all_out) {                                    // uniform: zeros from here to the end of the record
            Cp<T> z;
            z.x = z.y = 0;
#pragma unroll
            for (int mm = 0; mm < NMEM; ++mm)
                if (mm == 0 || has_b)
                    for (int tau = tau0 + lane; tau < P.snum; tau += 64) out[mm][tau] = z;
            return;
        }
#pragma unroll 1
        for (int t = 0; t < TT; ++t) {
            const int tau = tau0 + t;
            T psr[NMEM], psi[NMEM];
#pragma unroll
            for (int mm = 0; mm < NMEM; ++mm) psr[mm] = psi[mm] = 0;
            if (tau < P.snum) {                                   // uniform
                const double4 sc = stepc[tile & 1][t];
                const T dl = (T)sc.w;
#define PSS_EACH for (int j = 0, m = m0; j < G; ++j, ++m)
#define PSS_PIN(f)                                                                   \
    do {                                                                             \
        if constexpr (G == 4) asm volatile("" : "+v"(q[0].f), "+v"(q[G > 1 ? 1 : 0].f), "+v"(q[G > 2 ? 2 : 0].f), "+v"(q[G > 3 ? 3 : 0].f)); \
        else if constexpr (G == 2) asm volatile("" : "+v"(q[0].f), "+v"(q[G > 1 ? 1 : 0].f)); \
        else asm volatile("" : "+v"(q[0].f));                                        \
    } while (0)
#pragma unroll
                for (int m0 = 0; m0 < M; m0 += G) {
                    if ((gdead >> m0) & 1u) continue;             // uniform
                    struct {
                        T w, s, psi, s2, ec, es, m1, m2, ncr, nsr;
                    } q[G];
#pragma unroll
                    PSS_EACH q[j].w = fma(-xi[m], dl, (T)1);
                    PSS_PIN(w);
#pragma unroll
                    PSS_EACH q[j].s = __builtin_amdgcn_sqrtf(q[j].w);
                    PSS_PIN(s);
#pragma unroll
                    PSS_EACH q[j].psi = fma(Kf[m], q[j].s, -Kf[m]);
                    PSS_PIN(psi);
#pragma unroll
                    PSS_EACH q[j].s2 = q[j].psi * q[j].psi;
                    PSS_PIN(s2);
#pragma unroll
                    PSS_EACH {
                        q[j].ec = fma(q[j].s2, (T)(1.0 / 24), (T)-0.5);
                        q[j].es = fma(q[j].s2, (T)(1.0 / 120), (T)(-1.0 / 6));
                    }
                    PSS_PIN(ec);
                    PSS_PIN(es);
#pragma unroll
                    PSS_EACH {
                        q[j].ec = fma(q[j].s2, q[j].ec, (T)1);
                        q[j].es = fma(q[j].s2, q[j].es, (T)1);
                    }
                    PSS_PIN(ec);
                    PSS_PIN(es);
#pragma unroll
                    PSS_EACH q[j].es = q[j].es * q[j].psi;
                    PSS_PIN(es);
#pragma unroll
                    PSS_EACH {
                        q[j].m1 = sa[m] * q[j].es;
                        q[j].m2 = sa[m] * q[j].ec;
                    }
                    PSS_PIN(m1);
                    PSS_PIN(m2);
#pragma unroll
                    PSS_EACH {
                        q[j].ncr = fma(ca[m], q[j].ec, -q[j].m1);
                        q[j].nsr = fma(ca[m], q[j].es, q[j].m2);
                    }
                    if (__builtin_expect((gband >> m0) & 1u, 0)) {       // uniform
                        const double c = sc.x, thr = sc.y, vd = sc.z;
#pragma unroll
                        PSS_EACH {
                            if ((bandm >> m) & 1u) {
                                const int slot = (chunk * M + m) * 64 + lane;
                                double cr = fma(-c, x[m], 1.0);
                                const bool cheap = narrow && cr < PSS32_NARROW && fabs(cr - thr) > 1.0e-13;
                                if (!cheap) {
                                    // coss in the reference's own rounding (:456-460): what decides at the threshold
                                    const double wx = slot < P.nf ? P.w[slot] : 1.0;
                                    const double a = ((0.5 * vd) * kxk) / wx;
                                    cr = 1.0 - a * a;
                                }
                                if (cr <= thr || ((dead >> m) & 1u)) {
                                    // evanescent: zero from here on (:484-485 zero the spectrum itself: once out, out for good)
                                    dead |= 1u << m;
                                    Cp<T> z;
                                    z.x = z.y = 0;
#pragma unroll
                                    for (int mm = 0; mm < NMEM; ++mm) {
                                        sr_[mm][m] = 0;
                                        si_[mm][m] = 0;
                                        f0_lds[mm][m][lane] = z;
                                    }
                                    x[m] = 0.0;
                                    q[j].ncr = 1;
                                    q[j].nsr = 0;
                                } else if (cheap) {
                                    // the phase w dt sqrt(coss) < 0.45: float32 is enough for this step's rotation AND for the
                                    // phase sum (relative 1e-7 of a small phase, over the few dozen steps a frequency
                                    // spends in the band: 1e-6 rad)
                                    const T y32 = __builtin_amdgcn_sqrtf((T)cr);
                                    const T ph = (T)wdt[m] * y32, s2 = ph * ph;
                                    Phi[m] += (double)ph;
                                    q[j].ncr = fma(s2, fma(s2, fma(s2, fma(s2, (T)(1.0 / 40320), (T)(-1.0 / 720)), (T)(1.0 / 24)), (T)-0.5), (T)1);
                                    q[j].nsr = ph * fma(s2, fma(s2, fma(s2, fma(s2, (T)(1.0 / 362880), (T)(-1.0 / 5040)), (T)(1.0 / 120)), (T)(-1.0 / 6)), (T)1);
                                } else {
                                    const double phd = wdt[m] * sqrt(cr);            // :456-464
                                    Phi[m] += phd;
                                    T sn, cn;
                                    sincos_t<T>((T)phd, &sn, &cn);
                                    q[j].ncr = cn;
                                    q[j].nsr = sn;
                                }
                            }
                        }
                    }
#pragma unroll
                    PSS_EACH {
#pragma unroll
                        for (int mm = 0; mm < NMEM; ++mm) {
                            const T nr = fma(sr_[mm][m], q[j].ncr, -(si_[mm][m] * q[j].nsr));     // FK *= exp(i phi), :464
                            const T ni = fma(sr_[mm][m], q[j].nsr, si_[mm][m] * q[j].ncr);
                            sr_[mm][m] = nr;
                            si_[mm][m] = ni;
                            psr[mm] += nr;                                          // :487
                            psi[mm] += ni;
                        }
                    }
#pragma unroll
                    for (int mm = 0; mm < NMEM; ++mm) asm volatile("" : "+v"(psr[mm]), "+v"(psi[mm]));
                }
#undef PSS_PIN
#undef PSS_EACH
            }
#pragma unroll
            for (int mm = 0; mm < NMEM; ++mm) {
                Cp<T> pv;
                pv.x = psr[mm];
                pv.y = psi[mm];
                part[mm][t][lane] = pv;                   // (each lane reads back only what it wrote)
            }
        }
#pragma unroll
        for (int mm = 0; mm < NMEM; ++mm) {
            if (mm == 1 && !has_b) break;                 // uniform
            T acc[2 * TT];
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const Cp<T> pv = part[mm][t][lane];
                acc[2 * t] = pv.x;
                acc[2 * t + 1] = pv.y;
            }
            pss_write_tile<T>(acc, lane, out[mm], tau0, P.snum, final_scale);
        }
    }
}

// Frequencies per wave: 64 M, M <= P.sm_m = 4: what fits the registers with two wavenumbers per wave (float64, 8192^2, pairs:
// 2 -> 218 ms, 4 -> 179 ms; one wavenumber per wave with 8 spilled 200 registers: 264 ms).  A slab of a kx-sharded run (one
// wavenumber per wave) uses the SAME chunks: its rows are then bit-identical to the unsharded image's
// (tests/tools/fuzz_ps_sharded.py); chunks of 256 frequencies beyond.
static int ps_smooth_m(bool, bool) { return 4; }       // (float32 pairs with 2 per lane -- four waves per SIMD instead of two: 89.7 -> 98.6 ms)
static int ps_smooth_chunks(int nf, int m) { return nf <= 64 * m ? 1 : (nf + 64 * m - 1) / (64 * m); }

template <typename T, int M, int NMEM> static void ps_smooth_launch_one(const PsParams &P, int nchunks, hipStream_t st)
{
    const dim3 grid(NMEM == 2 ? P.tnum / 2 + 1 : P.nk, nchunks);
    if constexpr (sizeof(T) == 4) hipLaunchKernelGGL((ps_smooth32_kernel<M, NMEM>), grid, dim3(64), 0, st, P);
    else hipLaunchKernelGGL((ps_smooth_kernel<M, NMEM>), grid, dim3(64), 0, st, P);
}

// P.sm_nchunks, P.sm_m (and, beyond one chunk, P.sm_part: [chunks][nk][snum] complex) set by the caller; pairs: the whole
// wavenumber axis with kx[tnum - k] = -kx[k]
template <typename T, int NMEM> static void ps_smooth_launch_n(const PsParams &P, hipStream_t st)
{
    const int nf = P.nf, nchunks = P.sm_nchunks, mmax = P.sm_m;
    if (nf <= 64 || mmax == 1) ps_smooth_launch_one<T, 1, NMEM>(P, nchunks, st);
    else if (nf <= 128 || mmax == 2) ps_smooth_launch_one<T, 2, NMEM>(P, nchunks, st);
    else ps_smooth_launch_one<T, 4, NMEM>(P, nchunks, st);
}
template <typename T> static void ps_smooth_launch(const PsParams &P, hipStream_t st, bool pairs)
{
    if (pairs) ps_smooth_launch_n<T, 2>(P, st);
    else ps_smooth_launch_n<T, 1>(P, st);
    const int nchunks = P.sm_nchunks;
    if (nchunks > 1) {
        const size_t n = (size_t)P.nk * P.snum;
        hipLaunchKernelGGL((ps_smooth_sum_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                           reinterpret_cast<const Cp<T> *>(P.sm_part), reinterpret_cast<Cp<T> *>(P.TK), nchunks, n, P.snum);
    }
}
