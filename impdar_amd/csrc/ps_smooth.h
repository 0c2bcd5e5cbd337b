// Phase shift (Gazdag) for velocity profiles that change at (nearly) EVERY depth step -- a firn profile, a linear
// gradient: what getVelocityProfile's 2 * gradient(z(t)) gives for anything but a few thick layers.
// (included by phaseshift.hip)
//
// Reference: mig_python.py:438-487.  Per depth step tau and frequency w (wavenumber kx fixed per workgroup):
//     coss = 1 - (0.5 v_tau kx / w)^2;   FK[w] *= exp(i w dt sqrt(coss));   FK[w] = 0 for good once coss <= thr_tau;
//     TK[tau] += FK[w]
// The runs kernels (ps_vz32_kernel / ps_vz64_kernel, ps_mfma.h) live on runs of constant velocity, where the rotation
// per step is fixed.  Without runs the per-step kernel paid a float64 divide, square root and a sincos per
// (tau, w): 917 ms (float32) / 834 ms (float64) at 8192^2 against 15 / 60 ms for a four-layer table.
//
// Here the velocity is assumed to move LITTLE per step (it may still do anything: see the band below), and every
// per-step quantity is carried forward instead of recomputed:
//   * y = sqrt(coss) by two Newton steps from the previous step's y, with g = 1 / (2 y) carried by its own Newton
//     step between them: 7 float64 fma, no divide, no square root.  Two steps from a relative offset rho leave rho^4.
//   * the per-step rotation R = exp(i phi), phi = w dt y, by R *= 1 + i d - d^2 / 2 (- i d^3 / 6 + d^4 / 24 in float64) with
//     d = phi - phi_previous: a rotation OF the rotation, no sincos;
//   * float32 data: the state is re-anchored to FK0 exp(i Phi) with the float64 phase sum Phi every 64 steps, and R
//     to exp(i phi), as the runs kernels do (the recurrences drift, the anchors do not).
// The band.  Newton from the previous value needs coss to move little RELATIVE to itself: rho = |d coss| / (2 coss).
// A frequency about to turn evanescent (coss -> 0) violates that, and so does every frequency at a step where the
// velocity jumps.  Lanes with coss < 10 |d(v^2)| / v^2 (rho > 0.05; all lanes at step 0) take the exact path for
// that step: coss in the reference's own rounding (the evanescence test coss <= thr is decided there and only
// there), a real square root, a real sincos.  As v grows with depth the cut-off frequency v kx / 2 sweeps upwards:
// the frequencies in the band are a handful of NEIGHBOURS, i.e. lanes of one wave -- the exact path runs for about
// one (wave, frequency slot) pair per step.
#pragma once

// Band: coss < PSS_BAND |d(v^2)| / v^2, i.e. rho = |d coss| / (2 coss) > 1 / (2 PSS_BAND).  250 (rho > 0.002) held every
// carried value to its bar on every frequency and made the exact path a bottleneck: the frequencies in the band are
// neighbours, so ONE wave of the workgroup ran it step after step (~750 cycles a time) while seven waited at the tile
// barrier.  A frequency spends a few steps at the band's edge and is one of thousands in the sum: with rho up to 0.05
// the carried values are good to 6e-6 (float32: two Newton steps, third-order rotation) / 2e-8 (float64: three steps,
// sixth-order rotation) there and to rounding everywhere else.
#define PSS_BAND 10.0
#define PSS_NARROW 0.01     // bands up to this take the short way through the exact branch

template <typename T, int BLOCK, int M>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(2))) void ps_smooth_kernel(PsParams P)
{
    constexpr int TT = 16;
    constexpr int NW = BLOCK / 64;
    constexpr bool F32 = sizeof(T) == 4;
    constexpr int ANCHOR_TILES = 4;                       // float32: anchors every 64 steps
    // frequencies per group: their chains are independent (a wave that issues one dependent float64 operation after the
    // other waits ~16 cycles for each), ONE branch per group for the band, and a group whose 64 G lanes-and-slots have
    // all turned evanescent is skipped by its wave
    constexpr int G = M < 2 ? M : 2;
    constexpr double BAND = F32 ? PSS_BAND : 6.0 * PSS_BAND;
    extern __shared__ __attribute__((aligned(16))) char pss_smem[];
    // [M][BLOCK] original spectrum (float32: the anchors rotate it) | [TT][BLOCK] per-step partial sums of a lane's
    // frequencies | [2][NW][2 TT] wave sums
    Cp<T> *f0_lds = reinterpret_cast<Cp<T> *>(pss_smem);
    Cp<T> *part = reinterpret_cast<Cp<T> *>(pss_smem + (size_t)(F32 ? M : 0) * BLOCK * sizeof(Cp<T>));
    T(*red)[NW][2 * TT] = reinterpret_cast<T(*)[NW][2 * TT]>(reinterpret_cast<char *>(part) + (size_t)TT * BLOCK * sizeof(Cp<T>));
    // [2][TT] per-step constants {c = v^2, band, threshold, v}: written by 16 lanes a tile ahead, read by every lane as one
    // broadcast (they were ~10 uniform float64 instructions and two vector loads per step and wave)
    double4 *stepc = reinterpret_cast<double4 *>(reinterpret_cast<char *>(red) + (size_t)2 * NW * 2 * TT * sizeof(T));
    const int k = P.k0 + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Cp<T> *F = reinterpret_cast<const Cp<T> *>(P.F) + (size_t)k * P.fstride;
    Cp<T> *TK = reinterpret_cast<Cp<T> *>(P.TK) + (size_t)(k - P.k0) * P.snum;
    const double kxk = P.kx[k];

    // per owned frequency: x = (kx / 2w)^2, w dt, y = sqrt(coss) and g ~ 1 / (2 y) of the last step; float32 data also
    // the sum of y since the last anchor and the phase at that anchor
    double x[M], wdt[M], y[M], g[M], ysum[F32 ? M : 1], Phi[F32 ? M : 1];
    T sr_[M], si_[M], rc[M], rs[M];                       // state FK and rotation R = exp(i phi)
    T wdtT[F32 ? M : 1];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int slot = tid + m * BLOCK;
        Cp<T> f;
        f.x = f.y = 0;
        double w = 1.0;
        if (slot < P.nf) {
            f = ps_load_slot<T>(F, P, slot);
            w = P.w[slot];
        }
        const double a0 = 0.5 * kxk / w;
        x[m] = slot < P.nf ? a0 * a0 : 0.0;
        wdt[m] = w * P.dt;
        if (F32) wdtT[F32 ? m : 0] = (T)wdt[m];
        y[m] = 1.0;
        g[m] = 0.5;
        if (F32) {
            ysum[F32 ? m : 0] = 0.0;
            Phi[F32 ? m : 0] = 0.0;
            f0_lds[m * BLOCK + tid] = f;
        }
        sr_[m] = f.x;
        si_[m] = f.y;
        rc[m] = 1;
        rs[m] = 0;
    }
    auto step_constants = [&](int tile) {                 // lanes 0..15: the constants of tile `tile`
        const int tau = tile * TT + tid;
        if (tid < TT && tau < P.snum) {
            const double vd = P.vz[tau], c = vd * vd;
            double csb = 4.0;                             // step 0: every lane from scratch
            if (tau > 0) {
                const double vp = P.vz[tau - 1];
                // band: coss below this moved too much relative to itself for the carried values
                csb = BAND * fabs(c - vp * vp) / c + 1.0e-9;
            }
            stepc[(tile & 1) * TT + tid] = make_double4(c, csb, P.thr[tau], vd);
        }
    };
    step_constants(0);
    __syncthreads();
    unsigned dead = 0;                                    // bit m: frequency m of this lane has turned evanescent
    const int ntile = (P.snum + TT - 1) / TT;
    for (int tile = 0; tile < ntile; ++tile) {
        const int tau0 = tile * TT;
        if (F32 && tile > 0 && tile % ANCHOR_TILES == 0) {
            // anchor: the state from the ORIGINAL spectrum and the float64 phase, the rotation from its own phase
#pragma unroll
            for (int m = 0; m < M; ++m) {
                double ph = fma(wdt[m], ysum[F32 ? m : 0], Phi[F32 ? m : 0]);
                ph -= 6.283185307179586 * rint(ph * 0.15915494309189535);
                Phi[F32 ? m : 0] = ph;
                ysum[F32 ? m : 0] = 0.0;
                T sn, cs;
                sincos_t<T>((T)ph, &sn, &cs);
                const Cp<T> f0 = f0_lds[m * BLOCK + tid];
                sr_[m] = fma(f0.x, cs, -(f0.y * sn));
                si_[m] = fma(f0.x, sn, f0.y * cs);
                sincos_t<T>((T)(wdt[m] * y[m]), &sn, &cs);
                rc[m] = cs;
                rs[m] = sn;
                asm volatile("" : "+v"(sr_[m]), "+v"(si_[m]), "+v"(rc[m]), "+v"(rs[m]));
            }
        }
        // groups all of whose frequencies, in all 64 lanes of this wave, are out: nothing to carry, nothing to add (the
        // evanescent region v kx / 2 > w is a contiguous block of slots: 40 % of the plane at 1 m / 10 ns / 1.69e8 m/s)
        unsigned gdead = 0;
#pragma unroll
        for (int m0 = 0; m0 < M; m0 += G) {
            bool alive = false;
#pragma unroll
            for (int j = 0; j < G; ++j) alive = alive || !((dead >> (m0 + j)) & 1u);
            if (__builtin_amdgcn_ballot_w64(alive) == 0) gdead |= 1u << m0;
        }
        gdead = __builtin_amdgcn_readfirstlane(gdead);
#pragma unroll 1
        for (int t = 0; t < TT; ++t) {
            const int tau = tau0 + t;
            T psr = 0, psi = 0;
            if (tau < P.snum) {                                   // uniform
                const double4 sc = stepc[(tile & 1) * TT + t];
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
                    // (a struct per frequency, not an array per quantity: arrays of two floats become <2 x float> and
                    // the packed instructions that follows -- same rate as two plain ones -- cost moves to pack and unpack)
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
                    // left, 7e-8 on a float64 image of 700 steps; refreshed from y1 it is a true Newton step: rho^4)
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
                    if (!F32) {
                        // float64 data: a third step (rho^4 = 2e-12 per step at the band's edge adds up over a record)
#pragma unroll
                        PSS_EACH q[j].e1 = fma(-q[j].y2, q[j].y2, q[j].cs);
                        PSS_PIN(e1);
#pragma unroll
                        PSS_EACH q[j].y2 = fma(q[j].e1, q[j].gn, q[j].y2);
                        PSS_PIN(y2);
                    }
                    // R *= exp(i d), d = phi - phi_previous = w dt (y2 - y).  float32: 1 - d^2/2 + i d (re-anchored every
                    // 64 steps; d^3/6 < 1e-12 but at the band's edge); float64: cos to d^6, sin to d^5 (with
                    // 1 - d^2/2 + i (d - d^3/6) the modulus is 1 - d^4/24: a systematic loss, 6e-9 over 700 steps; at the
                    // band's edge d reaches 0.03)
#pragma unroll
                    PSS_EACH q[j].dd = q[j].y2 - y[m];
                    PSS_PIN(dd);
#pragma unroll
                    PSS_EACH q[j].d = F32 ? (T)q[j].dd * wdtT[F32 ? m : 0] : (T)(q[j].dd * wdt[m]);
                    PSS_PIN(d);
#pragma unroll
                    PSS_EACH q[j].hd2 = q[j].d * q[j].d * (T)0.5;
                    PSS_PIN(hd2);
#pragma unroll
                    PSS_EACH {
                        // with h = d^2 / 2:  cos d = 1 - h + h^2/6 - h^3/90,  sin d = d (1 - h/3 + h^2/30)
                        const T h = q[j].hd2;
                        q[j].ur = F32 ? (T)1 - h : fma(h, fma(h, fma(h, (T)(-1.0 / 90.0), (T)(1.0 / 6.0)), (T)-1), (T)1);
                        q[j].ui = F32 ? q[j].d : q[j].d * fma(h, fma(h, (T)(1.0 / 30.0), (T)(-1.0 / 3.0)), (T)1);
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
                        // (one or two frequencies of every wavenumber are in the band at any step of a smooth profile: this
                        // branch is taken by one wave of the workgroup at almost every step, and what it costs counts)
                        const double thr = sc.z, vd = sc.w;
                        const bool narrow = csb <= PSS_NARROW;    // uniform
#pragma unroll
                        PSS_EACH {
                            if (q[j].band) {
                                const int slot = tid + m * BLOCK;
                                double cr = q[j].cs;
                                const bool cheap = narrow && fabs(cr - thr) > 1.0e-13;
                                if (!cheap) {
                                    // coss in the reference's own rounding (:456-460): what decides at the threshold
                                    const double wx = slot < P.nf ? P.w[slot] : 1.0;
                                    const double a = ((0.5 * vd) * kxk) / wx;
                                    cr = 1.0 - a * a;
                                }
                                if (cr <= thr || !(slot < P.nf) || ((dead >> m) & 1u)) {
                                    // evanescent: zero from here on (:484-485 zero the spectrum itself: once out, out for
                                    // good -- a velocity that falls again must not revive the carried values of such a
                                    // lane, which are parked where they stay finite)
                                    dead |= 1u << m;
                                    sr_[m] = 0;
                                    si_[m] = 0;
                                    if (F32) {
                                        Cp<T> z;
                                        z.x = z.y = 0;
                                        f0_lds[m * BLOCK + tid] = z;
                                    }
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
                                    const T ph = (T)(wdt[m] * y0), s2 = ph * ph;
                                    T pc, ps;
                                    if (F32) {
                                        pc = fma(s2, fma(s2, fma(s2, fma(s2, (T)(1.0 / 40320), (T)(-1.0 / 720)), (T)(1.0 / 24)), (T)-0.5), (T)1);
                                        ps = fma(s2, fma(s2, fma(s2, fma(s2, (T)(1.0 / 362880), (T)(-1.0 / 5040)), (T)(1.0 / 120)), (T)(-1.0 / 6)), (T)1);
                                    } else {
                                        pc = (T)(1.0 / 87178291200.0);                       // 1/14!
                                        ps = (T)(1.0 / 1307674368000.0);                     // 1/15!
                                        pc = fma(s2, -pc, (T)(1.0 / 479001600.0));           // 1/12!
                                        ps = fma(s2, -ps, (T)(1.0 / 6227020800.0));          // 1/13!
                                        pc = fma(s2, -pc, (T)(1.0 / 3628800.0));
                                        ps = fma(s2, -ps, (T)(1.0 / 39916800.0));
                                        pc = fma(s2, -pc, (T)(1.0 / 40320.0));
                                        ps = fma(s2, -ps, (T)(1.0 / 362880.0));
                                        pc = fma(s2, -pc, (T)(1.0 / 720.0));
                                        ps = fma(s2, -ps, (T)(1.0 / 5040.0));
                                        pc = fma(s2, -pc, (T)(1.0 / 24.0));
                                        ps = fma(s2, -ps, (T)(1.0 / 120.0));
                                        pc = fma(s2, -pc, (T)0.5);
                                        ps = fma(s2, -ps, (T)(1.0 / 6.0));
                                        pc = fma(s2, -pc, (T)1);
                                        ps = fma(s2, -ps, (T)1);
                                    }
                                    q[j].ncr = pc;
                                    q[j].nsr = ps * ph;
                                } else {
                                    // this step from scratch (:456-464)
                                    q[j].y2 = sqrt(cr);
                                    q[j].gn = 0.5 / q[j].y2;
                                    T sn, cn;
                                    sincos_t<T>((T)(wdt[m] * q[j].y2), &sn, &cn);
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
                        if (F32) ysum[F32 ? m : 0] += q[j].y2;
                        rc[m] = q[j].ncr;
                        rs[m] = q[j].nsr;
                        const T nr = fma(sr_[m], q[j].ncr, -(si_[m] * q[j].nsr));     // FK *= exp(i phi), :464
                        const T ni = fma(sr_[m], q[j].nsr, si_[m] * q[j].ncr);
                        sr_[m] = nr;
                        si_[m] = ni;
                        psr += nr;                                                  // :487
                        psi += ni;
                    }
                    asm volatile("" : "+v"(psr), "+v"(psi));
                }
#undef PSS_PIN
#undef PSS_EACH
            }
            Cp<T> pv;
            pv.x = psr;
            pv.y = psi;
            part[t * BLOCK + tid] = pv;                   // (each lane reads back only what it wrote)
        }
        // ---- sum over frequencies: the tile's 16 step sums of this lane, wave butterfly, then across waves via LDS
        T acc[2 * TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const Cp<T> pv = part[t * BLOCK + tid];
            acc[2 * t] = pv.x;
            acc[2 * t + 1] = pv.y;
        }
        wave_reduce_scatter<T, 2 * TT>(acc, lane);
        T(*buf)[2 * TT] = red[tile & 1];
        if ((lane & 1) == 0) buf[wave][lane >> 1] = acc[0];
        step_constants(tile + 1);                         // (the other half of the table: last read a tile ago)
        __syncthreads();
        if (tid < 2 * TT) {
            T s = 0;
#pragma unroll
            for (int q = 0; q < NW; ++q) s += buf[q][tid];
            const int tau = tau0 + (tid >> 1);
            if (tau < P.snum) {
                T *dst = reinterpret_cast<T *>(TK + tau) + (tid & 1);
                *dst = s / (T)P.snum;                                   // TK /= snum, :492
            }
        }
        // red[] is double-buffered: the next tile writes the other buffer and the barrier of that tile orders it
    }
}

template <typename T, int BLOCK, int M> static size_t ps_smooth_lds()
{
    return (size_t)(sizeof(T) == 4 ? M : 0) * BLOCK * sizeof(Cp<T>) + (size_t)16 * BLOCK * sizeof(Cp<T>) +
           2 * (BLOCK / 64) * 32 * sizeof(T) + 2 * 16 * sizeof(double4);
}

template <typename T, int BLOCK, int M> static void ps_smooth_launch_one(const PsParams &P, hipStream_t st)
{
    auto k = ps_smooth_kernel<T, BLOCK, M>;
    const size_t lds = ps_smooth_lds<T, BLOCK, M>();
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(P.nk), dim3(BLOCK), lds, st, P);
}

// true when a kernel was launched (frequency counts it is instantiated for)
template <typename T> static bool ps_smooth_launch(const PsParams &P, hipStream_t st)
{
    const int nf = P.nf;
    if (nf <= 64) ps_smooth_launch_one<T, 64, 1>(P, st);
    else if (nf <= 128) ps_smooth_launch_one<T, 128, 1>(P, st);
    else if (nf <= 256) ps_smooth_launch_one<T, 256, 1>(P, st);
    else if (nf <= 512) ps_smooth_launch_one<T, 512, 1>(P, st);
    else if (nf <= 1024) ps_smooth_launch_one<T, 512, 2>(P, st);
    else if (nf <= 2048) ps_smooth_launch_one<T, 512, 4>(P, st);
    else if (nf <= 4096) ps_smooth_launch_one<T, 512, 8>(P, st);
    else return false;
    return true;
}
