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
    const int k = P.k0 + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Cp<T> *F = reinterpret_cast<const Cp<T> *>(P.F) + (size_t)k * P.fstride;
    Cp<T> *TK = reinterpret_cast<Cp<T> *>(P.TK) + (size_t)(k - P.k0) * P.snum;
    const double kxk = P.kx[k];

    // per owned frequency: x = (kx / 2w)^2, w dt, y = sqrt(coss) and g ~ 1 / (2 y) of the last step; float32 data also
    // the sum of y since the last anchor and the phase at that anchor
    double x[M], wdt[M], y[M], g[M], ysum[F32 ? M : 1], Phi[F32 ? M : 1];
    T sr_[M], si_[M], rc[M], rs[M];                       // state FK and rotation R = exp(i phi)
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
    double c_prev = 0.0;
    double vz_next = P.vz[0], thr_next = P.thr[0];
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
                // this step's velocity and threshold were requested a step ago
                const double vd = vz_next, thr = thr_next;
                const int tn = tau + 1 < P.snum ? tau + 1 : tau;
                vz_next = P.vz[tn];
                thr_next = P.thr[tn];
                const double c = vd * vd;
                // band: coss below this moved too much relative to itself for the carried values (every lane at step 0).
                // (1 / c from the float reciprocal: a float64 division here is ~40 instructions per step and wave)
                const double csb = tau == 0 ? 4.0 : BAND * fabs(c - c_prev) * (double)__builtin_amdgcn_rcpf((float)c) + 1.0e-9;
                c_prev = c;
#define PSS_EACH for (int j = 0, m = m0; j < G; ++j, ++m)
#define PSS_PIN(a)                                                                   \
    do {                                                                             \
        if constexpr (G == 2) asm volatile("" : "+v"(a[0]), "+v"(a[G > 1 ? 1 : 0])); \
        else asm volatile("" : "+v"(a[0]));                                         \
    } while (0)
#pragma unroll
                for (int m0 = 0; m0 < M; m0 += G) {
                    if ((gdead >> m0) & 1u) continue;             // uniform
                    double csg[G], e1g[G], y1g[G], rg[G], e2g[G], y2g[G], gng[G], ddg[G];
                    T dg[G], hd2g[G], urg[G], uig[G], m1g[G], m2g[G], ncrg[G], nsrg[G];
                    bool bandg[G];
                    bool any = false;
                    // stage by stage across the group (the pins keep the scheduler from putting the chains back one after
                    // the other to save registers)
#pragma unroll
                    PSS_EACH csg[j] = fma(-c, x[m], 1.0);
                    PSS_PIN(csg);
                    // Newton steps for y = sqrt(cs) from the previous step's y, g ~ 1 / (2 y).  (g is refreshed BETWEEN
                    // the steps: with the previous step's g in both, the second step only gains a factor rho -- rho^3
                    // left, 7e-8 on a float64 image of 700 steps; refreshed from y1 it is a true Newton step: rho^4)
#pragma unroll
                    PSS_EACH e1g[j] = fma(-y[m], y[m], csg[j]);
                    PSS_PIN(e1g);
#pragma unroll
                    PSS_EACH y1g[j] = fma(e1g[j], g[m], y[m]);
                    PSS_PIN(y1g);
#pragma unroll
                    PSS_EACH {
                        rg[j] = fma(-(y1g[j] + y1g[j]), g[m], 1.0);
                        e2g[j] = fma(-y1g[j], y1g[j], csg[j]);
                    }
                    PSS_PIN(rg);
                    PSS_PIN(e2g);
#pragma unroll
                    PSS_EACH gng[j] = fma(g[m], rg[j], g[m]);
                    PSS_PIN(gng);
#pragma unroll
                    PSS_EACH y2g[j] = fma(e2g[j], gng[j], y1g[j]);
                    PSS_PIN(y2g);
                    if (!F32) {
                        // float64 data: a third step (rho^4 = 2e-12 per step at the band's edge adds up over a record)
#pragma unroll
                        PSS_EACH e1g[j] = fma(-y2g[j], y2g[j], csg[j]);
                        PSS_PIN(e1g);
#pragma unroll
                        PSS_EACH y2g[j] = fma(e1g[j], gng[j], y2g[j]);
                        PSS_PIN(y2g);
                    }
                    // R *= exp(i d), d = phi - phi_previous = w dt (y2 - y).  float32: 1 - d^2/2 + i d (re-anchored every
                    // 64 steps; d^3/6 < 1e-12 but at the band's edge); float64: cos to d^6, sin to d^5 (with
                    // 1 - d^2/2 + i (d - d^3/6) the modulus is 1 - d^4/24: a systematic loss, 6e-9 over 700 steps; at the
                    // band's edge d reaches 0.03)
#pragma unroll
                    PSS_EACH ddg[j] = y2g[j] - y[m];
                    PSS_PIN(ddg);
#pragma unroll
                    PSS_EACH dg[j] = F32 ? (T)ddg[j] * (T)wdt[m] : (T)(ddg[j] * wdt[m]);
                    PSS_PIN(dg);
#pragma unroll
                    PSS_EACH hd2g[j] = dg[j] * dg[j] * (T)0.5;
                    PSS_PIN(hd2g);
#pragma unroll
                    PSS_EACH {
                        // with h = d^2 / 2:  cos d = 1 - h + h^2/6 - h^3/90,  sin d = d (1 - h/3 + h^2/30)
                        const T h = hd2g[j];
                        urg[j] = F32 ? (T)1 - h : fma(h, fma(h, fma(h, (T)(-1.0 / 90.0), (T)(1.0 / 6.0)), (T)-1), (T)1);
                        uig[j] = F32 ? dg[j] : dg[j] * fma(h, fma(h, (T)(1.0 / 30.0), (T)(-1.0 / 3.0)), (T)1);
                    }
                    PSS_PIN(urg);
#pragma unroll
                    PSS_EACH {
                        m1g[j] = rs[m] * uig[j];
                        m2g[j] = rs[m] * urg[j];
                    }
                    PSS_PIN(m1g);
                    PSS_PIN(m2g);
#pragma unroll
                    PSS_EACH {
                        ncrg[j] = fma(rc[m], urg[j], -m1g[j]);
                        nsrg[j] = fma(rc[m], uig[j], m2g[j]);
                        bandg[j] = csg[j] < csb;
                        any = any || bandg[j];
                    }
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(any) != 0, 0)) {
#pragma unroll
                        PSS_EACH {
                            if (bandg[j]) {
                                // this step from scratch, in the reference's own rounding (:456-460, :484-485)
                                const int slot = tid + m * BLOCK;
                                const double wx = slot < P.nf ? P.w[slot] : 1.0;
                                const double a = ((0.5 * vd) * kxk) / wx;
                                const double cr = 1.0 - a * a;
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
                                    y2g[j] = 1.0;
                                    gng[j] = 0.5;
                                    ncrg[j] = 1;
                                    nsrg[j] = 0;
                                } else {
                                    y2g[j] = sqrt(cr);
                                    gng[j] = 0.5 / y2g[j];
                                    T sn, cn;
                                    sincos_t<T>((T)(wdt[m] * y2g[j]), &sn, &cn);
                                    ncrg[j] = cn;
                                    nsrg[j] = sn;
                                }
                            }
                        }
                    }
#pragma unroll
                    PSS_EACH {
                        y[m] = y2g[j];
                        g[m] = gng[j];
                        if (F32) ysum[F32 ? m : 0] += y2g[j];
                        rc[m] = ncrg[j];
                        rs[m] = nsrg[j];
                        const T nr = fma(sr_[m], ncrg[j], -(si_[m] * nsrg[j]));     // FK *= exp(i phi), :464
                        const T ni = fma(sr_[m], nsrg[j], si_[m] * ncrg[j]);
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
           2 * (BLOCK / 64) * 32 * sizeof(T);
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
