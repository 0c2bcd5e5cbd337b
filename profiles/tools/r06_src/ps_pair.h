// Phase-shift frequency sum on the matrix cores, TWO wavenumbers per workgroup (float32 data; included by phaseshift.hip
// after ps_mfma.h, whose tiles, splits and set-up pass it shares).
//
// The phase of a frequency depends on the wavenumber through kx^2 only (mig_python.py:411-415, :456-460): the wavenumbers
// kx and -kx -- rows k and tnum - k of the transformed image -- turn by the same angles at every depth step and differ
// in their spectra alone.  ps_mfma_kernel's product  TK = sum_w S_w(a) B_w(b)  carries the spectrum in the state rows S;
// here it moves into the step factors,
//     TK[start + 64 a + b, +-k] = sum_w  [ 2^8 e^{i (Phi_w + 64 a phi_w)} ] * [ sigma F_w(+-k) e^{i (b + 1) phi_w} ]  =  sum_w A_w(a) G_w(b; +-k),
// so that the state tiles A -- pure rotations, 32 rows per 2048 depth steps and the larger part of the vector work -- are
// generated ONCE for the pair, and only the 64 step factors per (frequency, run) are made per wavenumber.
//
// Workgroup = 8 waves = (member of the pair) x (steps 16 p .. 16 p + 15 of every tile), one per CU (144 KB of LDS: the
// group's five state tiles TWICE, one step-factor tile per wave).  The two members' waves share a SIMD pairwise and run
// the round's two halves in OPPOSITE order --
//     member 0:  generate its half of the NEXT round's state tiles;  this round's MFMAs;  next round's step factors
//     member 1:  this round's MFMAs;  generate its half of the next round's state tiles;  next round's step factors
// -- so that on every SIMD one wave's vector work runs beside the other's matrix work (v_mfma_f32_32x32x16_f16 and vector
// instructions of different waves co-execute: SQ_VALU_MFMA_COEXEC_CYCLES, profiles/r05_ps_pair.txt) with ONE barrier per
// round; ps_mfma_kernel's two barriers put all waves of a workgroup into the same phase, and only a second workgroup on
// the CU overlapped anything.  (the review asked for the overlap inside a wave; it is had here
// between the two waves of a SIMD, which needs no instruction-level interleaving from the compiler.)
//
// k = 0 and the Nyquist row are their own partners: member 1 of those two workgroups generates its state rows and does
// nothing else.  A kx-sharded rank (a slab of wavenumbers without its mirror image) keeps ps_mfma_kernel.
#pragma once

constexpr int PP_WAVES = 8;
constexpr size_t PP_LDS_BYTES = ((size_t)2 * PM_NRB * 2 + (size_t)PP_WAVES * 2) * PM_TILE * 4 + 64;
static_assert(PM_CH == 32 && PM_NSUB == 2 && PM_NP == 4 && PM_NRB == 5 && PM_NSLOT == 8, "the roles below are written for 32-frequency chunks");
static_assert(PP_LDS_BYTES <= 160 * 1024, "one workgroup per CU");

__global__ __launch_bounds__(PP_WAVES * 64, 1) void ps_pair_kernel(PsMfmaParams Q)
{
    extern __shared__ __attribute__((aligned(16))) unsigned pm_lds[];
    const PsParams &P = Q.P;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef PP_MEM_LOW
    const int mem = wave & 1, part = wave >> 1;
#else
    const int mem = wave >> 2, part = wave & 3;
#endif
    // pairs in ascending |kx|: the long workgroups (few evanescent frequencies) first
    const int g = (int)blockIdx.x % Q.ngroups, pi = (int)blockIdx.x / Q.ngroups;
    const int ka = pi, kb = (P.tnum - pi) % P.tnum;
    const int k = mem ? kb : ka;
    const bool mine = !(mem && kb == ka);                  // (uniform over the wave) this wave has a wavenumber of its own
    const int om = lane & 31, hh = lane >> 5;              // frequency of the chunk; which of its two generating lanes
    const Cp<float> *Frow = reinterpret_cast<const Cp<float> *>(P.F) + (size_t)k * P.fstride;
    float *TKrow = reinterpret_cast<float *>(reinterpret_cast<Cp<float> *>(P.TK) + (size_t)k * P.snum);

    // ---- scale of this member's row: the largest component into [2^11, 2^12)
    float sigma = 1.0f;
    {
        float m = 0.f;
        if (mine)
            for (int slot = part * 64 + lane; slot < P.nf; slot += 256) {
                const Cp<float> f = ps_load_slot<float>(Frow, P, slot);
                m = fmaxf(m, fmaxf(fabsf(f.x), fabsf(f.y)));
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float *mx = reinterpret_cast<float *>(pm_lds);
        if (lane == 0) mx[wave] = m;
        __syncthreads();
#ifdef PP_MEM_LOW
        m = fmaxf(fmaxf(mx[mem], mx[mem + 2]), fmaxf(mx[mem + 4], mx[mem + 6]));
#else
        m = fmaxf(fmaxf(mx[4 * mem], mx[4 * mem + 1]), fmaxf(mx[4 * mem + 2], mx[4 * mem + 3]));
#endif
        __syncthreads();
        int e = 0;
        (void)frexpf(m, &e);
        if (m > 0.f && m < 3.0e38f) sigma = ldexpf(1.0f, 12 - e);
    }

    // ---- tiles as in ps_mfma_kernel (32 rows of 32 dwords, the 16-byte slots of row r at slot ^ ((r / 2) % 8)); state
    // tiles [buffer][row block][hi, lo], then one step-factor tile per wave [hi, lo]
    unsigned *const A0 = pm_lds;
    unsigned *const Bhi = pm_lds + (size_t)2 * PM_NRB * 2 * PM_TILE + (size_t)wave * 2 * PM_TILE, *const Blo = Bhi + PM_TILE;
    int wx[PM_NSLOT];
#pragma unroll
    for (int m = 0; m < PM_NSLOT; ++m) wx[m] = PM_ROW * hh + (om ^ (4 * m));
    const int rd0 = (lane & 31) * PM_ROW + ((4 * (lane >> 5)) ^ (4 * (((lane & 31) / PM_NSUB) % PM_NSLOT)));

    int brun[PM_NRB], ba0[PM_NRB], blong[PM_NRB];
#pragma unroll
    for (int rb = 0; rb < PM_NRB; ++rb) {
        const int2 d = Q.blocks[(size_t)g * PM_NRB + rb];
        brun[rb] = __builtin_amdgcn_readfirstlane(d.x);
        ba0[rb] = __builtin_amdgcn_readfirstlane(d.y);
        blong[rb] = brun[rb] >= 0 ? Q.long_of[brun[rb]] : 0;
    }
    // first tile of the row block whose state rows this wave generates (block `part`)
    const int a0_mine = __builtin_amdgcn_readfirstlane(part == 0 ? ba0[0] : part == 1 ? ba0[1] : part == 2 ? ba0[2] : ba0[3]);
    pm_float16 acc[PM_NRB];
#pragma unroll
    for (int rb = 0; rb < PM_NRB; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[rb][i] = 0.f;

    // ---- this wave's share of a round's state tiles (pure rotations, 2^8 e^{i theta}; zero rows where the frequency takes
    // no part in the tile's run): rows 16 mem + hh + 2 j of tile `part`, rows 4 wave + hh + 2 j of the fifth tile
    auto gen_A = [&](unsigned *Aw, const double (&incs)[PM_NRB], const double (&phis)[PM_NRB], unsigned rb_in) {
        double inc_o = incs[0], phi_o = phis[0];
#pragma unroll
        for (int rb = 1; rb < PM_NP; ++rb) {
            const bool me = rb == part;
            inc_o = me ? incs[rb] : inc_o;
            phi_o = me ? phis[rb] : phi_o;
        }
        {
            const bool in_o = phi_o == phi_o;
            float sx, cx, Es, Ec;
            pm_sincos((in_o ? phi_o : 0.0) + (double)(PM_TT * (a0_mine + 16 * mem + hh)) * inc_o, &sx, &cx);
            pm_sincos((double)(PM_NSUB * PM_TT) * inc_o, &Es, &Ec);
            const float amp = in_o ? 256.f : 0.f;
            float xr = amp * cx, xi = amp * sx;
            unsigned *Ahi = Aw + (size_t)part * 2 * PM_TILE + 512 * mem, *Alo = Ahi + PM_TILE;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float rr, ri;
                const int o = 64 * j + wx[j];
                Ahi[o] = pm_split(xr, xi, &rr, &ri);
                Alo[o] = pm_pack(rr, ri);
                const float nxr = fmaf(xr, Ec, -(xi * Es)), nxi = fmaf(xr, Es, xi * Ec);
                xr = nxr;
                xi = nxi;
            }
        }
        if ((rb_in >> (PM_NRB - 1)) & 1u) {                                   // uniform
            constexpr int rb = PM_NRB - 1;
            const double inc = incs[rb];
            const bool in = phis[rb] == phis[rb];
            float s, c, Es, Ec;
            pm_sincos((in ? phis[rb] : 0.0) + (double)(PM_TT * (ba0[rb] + 4 * wave + hh)) * inc, &s, &c);
            pm_sincos((double)(PM_NSUB * PM_TT) * inc, &Es, &Ec);
            const float amp = in ? 256.f : 0.f;
            float sr = amp * c, si = amp * s;
            unsigned *Ahi = Aw + (size_t)rb * 2 * PM_TILE, *Alo = Ahi + PM_TILE;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * wave + jj;                                  // row 2 j + hh
                const int o = 64 * j + PM_ROW * hh + (om ^ (4 * (j & 7)));
                float rr, ri;
                Ahi[o] = pm_split(sr, si, &rr, &ri);
                Alo[o] = pm_pack(rr, ri);
                const float nr = fmaf(sr, Ec, -(si * Es)), ni = fmaf(sr, Es, si * Ec);
                sr = nr;
                si = ni;
            }
        }
    };
    // ---- this wave's step factors of one run: G = sigma F e^{i (b + 1) phi}, steps b = 16 part + hh + 2 j.  Against a state
    // (A_re, A_im) the column (b, re) holds (G_re, -G_im), the column (b, im) holds (G_im, G_re)
    auto gen_B = [&](float fr, float fi, double inc) {
        float e2s, e2c, bs, bc;
        pm_sincos((double)PM_NSUB * inc, &e2s, &e2c);
        pm_sincos((double)(16 * part + hh + 1) * inc, &bs, &bc);
        float gr = fmaf(fr, bc, -(fi * bs)), gi = fmaf(fr, bs, fi * bc);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float rc, rs;
            const unsigned h0 = pm_split(gr, gi, &rc, &rs), l0 = pm_pack(rc, rs);
            const int o = 64 * j + wx[j];
            Bhi[o] = pm_conj(h0);
            Bhi[o + 16 * PM_ROW] = pm_swap(h0);
            Blo[o] = pm_conj(l0);
            Blo[o + 16 * PM_ROW] = pm_swap(l0);
            const float nr = fmaf(gr, e2c, -(gi * e2s)), ni = fmaf(gr, e2s, gi * e2c);
            gr = nr;
            gi = ni;
        }
    };

    unsigned nmfma = 0;
    // ---- a round's products: every live row block of the group against this wave's 16 steps.  The step factors of the
    // group's FIRST run are in the wave's tile (written at the end of the slot before); another run's are made here
    auto products = [&](const unsigned *Ar, unsigned rb_in, float fr, float fi, const double (&incs)[PM_NRB]) {
        // (tried and dropped, profiles/r05_ps_pair.txt: state operands requested two K-steps ahead across row blocks -- no
        // change; the MFMAs of two row blocks of one run alternating, so that none waits for its predecessor's accumulator --
        // 23 spilled registers, slower)
        uint4 bh[4], bl[4];
#pragma unroll
        for (int rb = 0; rb < PM_NRB; ++rb) {
            const int run = brun[rb];
            if (run < 0 || !((rb_in >> rb) & 1u)) continue;                   // uniform
            if (rb > 0 && brun[rb - 1] != run) {
                __builtin_amdgcn_wave_barrier();
                gen_B(fr, fi, incs[rb]);
            }
            __builtin_amdgcn_wave_barrier();
            if (rb == 0 || brun[rb - 1] != run) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int o = rd0 ^ (8 * s);
                    bh[s] = *reinterpret_cast<const uint4 *>(Bhi + o);
                    bl[s] = *reinterpret_cast<const uint4 *>(Blo + o);
                }
            }
            const unsigned *Ahi = Ar + (size_t)rb * 2 * PM_TILE, *Alo = Ahi + PM_TILE;
            nmfma += 12;
            uint4 ra_hi = *reinterpret_cast<const uint4 *>(Ahi + rd0), ra_lo = *reinterpret_cast<const uint4 *>(Alo + rd0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const pm_half8 a_hi = __builtin_bit_cast(pm_half8, ra_hi), a_lo = __builtin_bit_cast(pm_half8, ra_lo);
                const pm_half8 b_hi = __builtin_bit_cast(pm_half8, bh[s]), b_lo = __builtin_bit_cast(pm_half8, bl[s]);
                if (s + 1 < 4) {
                    const int o = rd0 ^ (8 * (s + 1));
                    ra_hi = *reinterpret_cast<const uint4 *>(Ahi + o);
                    ra_lo = *reinterpret_cast<const uint4 *>(Alo + o);
                }
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, acc[rb], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    const int nchunk = P.nf / PM_CH;
    const double2 *tab = Q.runtab + (size_t)ka * P.nf * Q.nlong;      // kx^2 decides: one table for the pair
    Cp<float> f_next = mine ? ps_load_slot<float>(Frow, P, om) : Cp<float>{0.f, 0.f};
    double2 t_next[PM_NRB];
#pragma unroll
    for (int rb = 0; rb < PM_NRB; ++rb) t_next[rb] = tab[(size_t)om * Q.nlong + blong[rb]];

    // the round whose products are due: its live row blocks, spectrum, phases per step
    bool have_cur = false;                                  // uniform over the workgroup
    unsigned rb_cur = 0;
    float fr_cur = 0.f, fi_cur = 0.f;
    double inc_cur[PM_NRB];
#pragma unroll
    for (int rb = 0; rb < PM_NRB; ++rb) inc_cur[rb] = 0.0;
    int buf = 0;
    for (int c = 0; c <= nchunk; ++c) {
        const bool flush = c == nchunk;                     // one more slot for the last round's products
        double incs[PM_NRB], phis[PM_NRB];
        float fr = 0.f, fi = 0.f;
        unsigned rb_in = 0;
        if (!flush) {
            fr = f_next.x * sigma;
            fi = f_next.y * sigma;
#pragma unroll
            for (int rb = 0; rb < PM_NRB; ++rb) {
                incs[rb] = t_next[rb].x;
                phis[rb] = t_next[rb].y;
            }
#ifdef PP_ABL_NOLOAD
            const int sn = (nchunk - 1) * PM_CH + om;      // timing only: the same (live) entries every round
#else
            const int sn = min(c + 1, nchunk - 1) * PM_CH + om;
#endif
            if (mine) f_next = ps_load_slot<float>(Frow, P, sn);
#pragma unroll
            for (int rb = 0; rb < PM_NRB; ++rb) t_next[rb] = tab[(size_t)sn * Q.nlong + blong[rb]];
            // a chunk none of whose frequencies takes part in any of the group's runs (the evanescent band: out for good, NaN
            // start phase) is skipped; every wave sees the same 32 x PM_NRB entries and decides alike
#pragma unroll
            for (int rb = 0; rb < PM_NRB; ++rb)
                if (brun[rb] >= 0 && __builtin_amdgcn_ballot_w64(phis[rb] == phis[rb]) != 0) rb_in |= 1u << rb;
            if (rb_in == 0) continue;
        } else {
            if (!have_cur) break;
#pragma unroll
            for (int rb = 0; rb < PM_NRB; ++rb) incs[rb] = phis[rb] = 0.0;
        }
        unsigned *Aw = A0 + (size_t)(buf ^ 1) * PM_NRB * 2 * PM_TILE;
        const unsigned *Ar = A0 + (size_t)buf * PM_NRB * 2 * PM_TILE;
#ifndef PP_ABL_NOGEN
        if (mem == 0 && !flush) gen_A(Aw, incs, phis, rb_in);
#endif
#ifndef PP_ABL_NOPROD
        if (have_cur && mine) products(Ar, rb_cur, fr_cur, fi_cur, inc_cur);
#endif
#ifndef PP_ABL_NOGEN
        if (mem == 1 && !flush) gen_A(Aw, incs, phis, rb_in);
        if (!flush && mine) gen_B(fr, fi, incs[0]);
#endif
#ifndef PP_ABL_NOBAR
        __syncthreads();                                    // the next round's tiles are complete, this round's are read
#endif
        buf ^= 1;
        have_cur = !flush;
        rb_cur = rb_in;
        fr_cur = fr;
        fi_cur = fi;
#pragma unroll
        for (int rb = 0; rb < PM_NRB; ++rb) inc_cur[rb] = incs[rb];
    }

    if (lane == 0 && Q.mfma_count && mine) atomicAdd(Q.mfma_count, (unsigned long long)nmfma);
    if (!mine) return;
    // accumulator register i of lane l: row (i & 3) + 8 (i >> 2) + 4 (l >> 5), column l & 31 = 16 (im ? 1 : 0) + b
    const float scale = 1.0f / (sigma * 256.0f * (float)P.snum);
#pragma unroll
    for (int rb = 0; rb < PM_NRB; ++rb) {
        const int run = brun[rb];
        if (run < 0) continue;
        const int start = Q.runs[run].start, end = start + Q.runs[run].len;
        const int col = lane & 31;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
            const int tau = start + PM_TT * (ba0[rb] + row) + 16 * part + (col & 15);
            if (tau < end) TKrow[2 * (size_t)tau + (col >> 4)] = acc[rb][i] * scale;
        }
    }
}
