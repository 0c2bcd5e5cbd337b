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
// velocity jumps.  Lanes with coss < 250 |d(v^2)| / v^2 (rho > 0.002; all lanes at step 0) take the exact path for
// that step: coss in the reference's own rounding (the evanescence test coss <= thr is decided there and only
// there), a real square root, a real sincos.  As v grows with depth the cut-off frequency v kx / 2 sweeps upwards:
// the frequencies in the band are a handful of NEIGHBOURS, i.e. lanes of one wave -- the exact path runs for about
// one (wave, frequency slot) pair per step.
#pragma once

template <typename T> struct PsSmoothTraits;
template <> struct PsSmoothTraits<float> { static constexpr bool anchors = true; };
template <> struct PsSmoothTraits<double> { static constexpr bool anchors = false; };

template <typename T, int BLOCK, int M>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(2))) void ps_smooth_kernel(PsParams P)
{
    constexpr int TT = 16;
    constexpr int NW = BLOCK / 64;
    constexpr bool F32 = sizeof(T) == 4;
    constexpr int ANCHOR_TILES = 4;                       // float32: anchors every 64 steps
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

    double x[M], wdt[M], y[M], g[M], php[M], Phi[F32 ? M : 1];
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
        php[m] = 0.0;
        if (F32) {
            Phi[F32 ? m : 0] = 0.0;
            f0_lds[m * BLOCK + tid] = f;
        }
        sr_[m] = f.x;
        si_[m] = f.y;
        rc[m] = 1;
        rs[m] = 0;
    }
    double c_prev = 0.0;
    unsigned dead = 0;                                    // bit m: frequency m has turned evanescent
    const int ntile = (P.snum + TT - 1) / TT;
    for (int tile = 0; tile < ntile; ++tile) {
        const int tau0 = tile * TT;
        if (F32 && tile > 0 && tile % ANCHOR_TILES == 0) {
            // anchor: the state from the ORIGINAL spectrum and the float64 phase sum, the rotation from its own phase
#pragma unroll
            for (int m = 0; m < M; ++m) {
                double ph = Phi[F32 ? m : 0];
                ph -= 6.283185307179586 * rint(ph * 0.15915494309189535);
                Phi[F32 ? m : 0] = ph;
                T sn, cs;
                sincos_t<T>((T)ph, &sn, &cs);
                const Cp<T> f0 = f0_lds[m * BLOCK + tid];
                sr_[m] = fma(f0.x, cs, -(f0.y * sn));
                si_[m] = fma(f0.x, sn, f0.y * cs);
                sincos_t<T>((T)php[m], &sn, &cs);
                rc[m] = cs;
                rs[m] = sn;
                asm volatile("" : "+v"(sr_[m]), "+v"(si_[m]), "+v"(rc[m]), "+v"(rs[m]));
            }
        }
#pragma unroll 1
        for (int t = 0; t < TT; ++t) {
            const int tau = tau0 + t;
            T psr = 0, psi = 0;
            if (tau < P.snum) {                                   // uniform
                const double vd = P.vz[tau], thr = P.thr[tau];
                const double c = vd * vd;
                // band: coss below this moved too much relative to itself for the carried values (every lane at step 0)
                const double csb = tau == 0 ? 4.0 : 250.0 * fabs(c - c_prev) / c + 1.0e-9;
                c_prev = c;
                // G frequencies at a time: their chains are independent, and a wave that issues one dependent float64
                // operation after the other waits ~16 cycles for each (one chain at a time measured 160 cycles per
                // frequency and step with two waves per SIMD); ONE branch per group for the band
                constexpr int G = M < 4 ? M : 4;
#pragma unroll
                for (int m0 = 0; m0 < M; m0 += G) {
                    double y2g[G], gng[G], phig[G];
                    T ncrg[G], nsrg[G];
                    bool bandg[G];
                    bool any = false;
#pragma unroll
                    for (int j = 0; j < G; ++j) {
                        const int m = m0 + j;
                        const double cs = fma(-c, x[m], 1.0);
                        // two Newton steps for y = sqrt(cs) from the previous step's y, g ~ 1 / (2 y)
                        // (g is refreshed BETWEEN the two steps: with the previous step's g in both, the second step
                        // only gains a factor rho -- rho^3 left, 7e-8 on a float64 image of 700 steps; with g refreshed
                        // from y1 the second step is a true Newton step: rho^4)
                        const double e1 = fma(-y[m], y[m], cs);
                        const double y1 = fma(e1, g[m], y[m]);
                        const double r = fma(-(y1 + y1), g[m], 1.0);
                        const double g1 = fma(g[m], r, g[m]);
                        const double e2 = fma(-y1, y1, cs);
                        double y2 = fma(e2, g1, y1);
                        if (!F32) y2 = fma(fma(-y2, y2, cs), g1, y2);      // float64 data: a third step (rho^4 = 2e-12 per
                        //                                                     step at the band's edge adds up over a record)
                        y2g[j] = y2;
                        gng[j] = g1;
                        phig[j] = wdt[m] * y2;
                        // R *= exp(i d), d = phi - phi_previous, to second (float64: third) order
                        const T d = (T)(phig[j] - php[m]);
                        const T hd2 = d * d * (T)0.5;
                        // float32: 1 - d^2/2 + i d (re-anchored every 64 steps); float64: cos and sin to d^4 / d^3 -- the
                        // modulus of 1 - d^2/2 + i (d - d^3/6) is 1 - d^4/24, a systematic loss (6e-9 over 700 steps)
                        const T ur = F32 ? (T)1 - hd2 : fma(hd2, fma(hd2, (T)(1.0 / 6.0), (T)-1), (T)1);
                        const T ui = F32 ? d : fma(-d, hd2 * (T)(1.0 / 3.0), d);
                        ncrg[j] = fma(rc[m], ur, -(rs[m] * ui));
                        nsrg[j] = fma(rc[m], ui, rs[m] * ur);
                        bandg[j] = cs < csb;
                        any = any || bandg[j];
                    }
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(any) != 0, 0)) {
#pragma unroll
                        for (int j = 0; j < G; ++j) {
                            const int m = m0 + j;
                            if (bandg[j]) {
                                // this step from scratch, in the reference's own rounding (:456-460, :484-485)
                                const int slot = tid + m * BLOCK;
                                const double wx = slot < P.nf ? P.w[slot] : 1.0;
                                const double a = ((0.5 * vd) * kxk) / wx;
                                const double cr = 1.0 - a * a;
                                if (cr <= thr || !(slot < P.nf) || ((dead >> m) & 1u)) {
                                    // (once out, out for good -- :484-485 zero the spectrum itself; a velocity that falls
                                    // again must not revive the carried values of such a lane)
                                    dead |= 1u << m;
                                    // evanescent: zero from here on; the carried values are parked where they stay finite
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
                                    phig[j] = wdt[m];
                                    ncrg[j] = 1;
                                    nsrg[j] = 0;
                                } else {
                                    y2g[j] = sqrt(cr);
                                    gng[j] = 0.5 / y2g[j];
                                    phig[j] = wdt[m] * y2g[j];
                                    T sn, cn;
                                    sincos_t<T>((T)phig[j], &sn, &cn);
                                    ncrg[j] = cn;
                                    nsrg[j] = sn;
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < G; ++j) {
                        const int m = m0 + j;
                        y[m] = y2g[j];
                        g[m] = gng[j];
                        php[m] = phig[j];
                        if (F32) Phi[F32 ? m : 0] += phig[j];
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
