// Phase-shift frequency sum on the matrix cores (float32 data; included by phaseshift.hip).
//
// Inside a run of constant velocity every frequency turns by a fixed angle per depth step, so with the steps of the
// run cut into tiles of 64 (step = start + 64 a + b) the sum the reference accumulates (mig_python.py:418-420, :464, :487)
// factorises,
//     TK[start + 64 a + b, k] = sum_w  [ F_w e^{i Phi_w} e^{i 64 a phi_w} ] * [ e^{i (b+1) phi_w} ]  =  sum_w S_w(a) B_w(b),
// a complex matrix product (tiles x frequencies) . (frequencies x 64 steps) per wavenumber: 1/64 of the rotations
// remain vector work for the states S_w(a) (a recurrence over a from an anchor that carries the phase in float64)
// plus 64 step factors per (frequency, run); the contraction over the frequencies -- all of the rotate-accumulate
// work of ps_kernel / ps_vz32_kernel -- goes to v_mfma_f32_32x32x16_f16.
// Real form: [C_re C_im] = [S_re S_im] . [[B_re B_im] [-B_im B_re]].
//
// float32 accuracy from float16 operands: every operand is split into two halves (x = hi + lo, 11 + 11 bits) and
// three products are accumulated in float32, hi.hi + hi.lo + lo.hi (the dropped lo.lo term is 2^-22 of the product).
// The spectrum row is scaled by a power of two so that its largest component sits in [2^11, 2^12), the step factors
// by 2^8 (keeps their low halves out of the float16 subnormals); both scales are exact and divided out at the end.
// Phases are float64 on the host side of every anchor (Phi at the start of a run, the increment of the run), and
// enter float32 as a two-float argument (sincos of the high part, first-order correction by the low part).
//
// Work split: one workgroup of 8 waves per (wavenumber, group of up to 5 row blocks).  A row block is 32 tiles of one
// run (2048 depth steps: the 32 rows of an MFMA accumulator).  Wave (q, p) takes the frequency chunks c = q mod 2
// (32 frequency slots each) and the STEPS 16 p .. 16 p + 15 of every tile (16 steps x (re, im) = the 32 columns of
// its accumulators), for all row blocks of the group: up to 5 accumulators.  Per round (one chunk per frequency
// half): wave (q, p) GENERATES the state tile of row block p -- lane (frequency, hh) its rows hh, hh + 2, ..., hh + 30
// from one float64-phase anchor by S *= e^{i 128 phi} -- into the half's LDS, and its own step-factor tile B;
// barrier; 5 row blocks x 4 K-steps x 3 MFMAs against the half's state tiles; barrier.  Every state element is
// generated once per workgroup and used by four waves, every step factor once.  At the end the two frequency halves
// of every accumulator are added in a fixed order through LDS (no atomics).
//
// How it got here (config 5, constant velocity, kernel ms; profiles/r03_ps_mfma_*.txt): 16-step tiles, 8 waves, every
// wave its own tiles for 8-9 row blocks: 10.8; 16 waves with one row part each, swizzled unpadded tiles, recurrence
// carried across row blocks: 10.9 -- ablation: 5.1 ms of it generating state rows, 1.0 ms step factors, 1.6 ms MFMA
// exposed: a row costs ~24 vector-pipe cycles (two float16 pack conversions and two mixed-precision fmas at half
// rate beside the four rotation ops); 64-step tiles with state tiles shared by the waves of a frequency quarter
// (16 waves, 16-frequency chunks) generate 1/16 of the rows: 9.35 -- now the per-frequency set-up (two float64
// divisions, a square root, four phase reductions + sincos) dominated, executed by 16 lanes per frequency; 32-frequency
// chunks and 8 waves halve that redundancy: 8.7; the float64 set-up moved to its own pass (ps_setup_kernel): 8.55 for
// constant velocity, 17.1 -> 12.2 for the config-5 table; workgroups of 4 waves (one frequency part, no final
// reduction), two per CU: 7.8.  In-kernel stamps of that version (IMPDAR_PS_STAMPS, profiles/r03_ps_mfma_stamps.txt): per
// wave and round ~3000 cycles for the state tile, ~1350 for the step factors, ~2750 for the 48 MFMAs (1536 of pipe
// time), ~500 at the two barriers; two independent recurrences per lane and operand reads a K-step ahead changed none
// of them.
//
// Runs of up to PM_SHORT steps -- 2 * gradient(z(t)) of a layered table smears every layer boundary over three or
// four steps, each a "run" of its own -- would waste a whole 512-step row block each: they get none (the frequency
// kernel only carries the phase across them) and ps_trans_kernel sums their few steps directly.
//
// Frequencies on the evanescent boundary of some run (|coss| < 1e-8: kept or dropped by the reference at every
// step's own velocity, mig_python.py:456-485) take no part: they are listed per wavenumber and ps_edge_kernel walks
// them over the whole depth axis in float64 afterwards, the way the reference does.
#pragma once

typedef _Float16 pm_half8 __attribute__((ext_vector_type(8)));
typedef float pm_float16 __attribute__((ext_vector_type(16)));

constexpr int PM_TT = 64;                   // depth steps per tile
#ifndef PM_CHUNK
#define PM_CHUNK 32      // 16 (three waves per SIMD, 168 VGPRs) measured 12.1 / 20.2 ms at config 5 against 11.3 / 19.1 ms for 32 (two waves): the per-frequency sincos work doubles
#endif
constexpr int PM_CH = PM_CHUNK;             // frequency slots per chunk (16 or 32): lane % PM_CH
constexpr int PM_NSUB = 64 / PM_CH;         // lanes per frequency: each generates every PM_NSUB-th row / step
constexpr int PM_ROW = PM_CH;               // dwords per tile row (one complex float16 pair per frequency; slots XOR-swizzled, no padding)
constexpr int PM_NSLOT = PM_ROW / 4;        // 16-byte slots per row
constexpr int PM_TILE = 32 * PM_ROW;        // dwords per tile (hi or lo halves of 32 rows): 2 or 4 KB
constexpr int PM_NQ = 1, PM_NP = 4;         // waves of a workgroup: (one frequency part) x step blocks of 16
constexpr int PM_NRB = 5;                   // row blocks per group (state tiles per frequency half, accumulators per wave)
constexpr int PM_WAVES = PM_NQ * PM_NP;
constexpr int PM_RED_LD = 20;               // dwords per lane in the final reduction image (16 + 4: conflict-free b128)
constexpr int PM_EMAX = 16;                 // boundary frequencies listed per wavenumber
constexpr int PM_MAX_RUNS = 96;
constexpr int PM_SHORT = 8;                 // runs of up to this many steps (the few steps a layer boundary is smeared over) get no row blocks: ps_trans_kernel
constexpr size_t PM_LDS_BYTES = ((size_t)PM_NQ * PM_NRB * 2 + (size_t)PM_WAVES * 2) * PM_TILE * 4 + 64;

struct PsMfmaRun {
    double v;               // velocity of the run (v(z)); unused for constant velocity
    int start, len;         // first depth step, number of steps
};

struct PsMfmaParams {
    PsParams P;
    PsMfmaRun runs[PM_MAX_RUNS];
    int nruns;
    const int2 *blocks;     // [ngroups][PM_NRB]: (run, first tile inside the run) of every row block; run = -1: none
    int ngroups;
    int *edge_cnt;          // [tnum] boundary frequencies found for the wavenumber (v(z))
    int *edge_list;         // [tnum][PM_EMAX] their slots
    // ps_setup_kernel -> ps_mfma_kernel: per (wavenumber, frequency slot, LONG run) the phase per depth step and the
    // phase at the start of the run, both float64; the start phase is NaN where the frequency takes no part in the run
    // (evanescent at or before it, or a boundary frequency)
    double2 *runtab;        // [tnum][nf][nlong]
    int nlong;
    int long_of[PM_MAX_RUNS];   // run -> index among the long runs (-1: short)
    int vz;
    int pairs;              // the whole wavenumber axis with kx[tnum - k] = -kx[k]: runtab row min(k, tnum - k) serves both (ps_setup_kernel makes it once)
    unsigned long long *mfma_count;     // MFMA instructions issued, summed over the launch (bench.py: mfma_flop_executed)
};

// coss = 1 - (0.5 v kx / w)^2 (mig_python.py:456) with the division by w as a multiplication by rw = 1/w: one
// float64 division per frequency instead of one per (frequency, run).  The last-bit difference from the reference's
// own rounding moves a phase by ~1e-16 relative; the only place where the last bit of coss decides anything is the
// boundary band |coss| < 1e-8, and those frequencies are classified HERE, by this one function, for both kernels that
// must agree (ps_mfma_kernel, ps_trans_kernel) and then walked by ps_edge_kernel with the reference's own expression.
__device__ __forceinline__ double pm_coss(double v, double kxk, double rw)
{
    const double a = (0.5 * v * kxk) * rw;
    return 1.0 - a * a;
}

// sqrt(x) for x in [1e-8, 1]: float32 reciprocal square root as the seed, two Newton steps in float64 (relative
// error ~1e-16 -- not correctly rounded; the phase per step it feeds tolerates 1e-13).  The IEEE sqrt is ~2.5x the
// instructions, and ps_setup_kernel takes one per (wavenumber, frequency, run).
__device__ __forceinline__ double pm_sqrt01(double x)
{
    double y = (double)__builtin_amdgcn_rsqf((float)x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return x * y;
}

__device__ __forceinline__ double pm_wrap(double x) { return x - 6.283185307179586 * rint(x * 0.15915494309189535); }

// sin / cos of a float64 angle (any size the phases of a record reach) as float32.  The reduction is done in float64 --
// t = x 2/pi, n = rint(t), r = (t - n) pi/2 in [-pi/4, pi/4], exact to ~1e-12 -- so the float32 argument of the
// polynomials carries no more than its own rounding (3e-8), and the quadrant is n mod 4: 6 float64-rate and 17 float32
// instructions.  (Rounds 2-3: wrap to [-pi, pi] in float64, split into a high and a low float, the library's sincosf
// on the high one -- its own reduction and both polynomials, ~50 instructions -- and a first-order correction by the
// low one; five of these per lane and round were a third of the kernel's vector instructions.)  Polynomials: the
// single-precision kernels of Cephes on [-pi/4, pi/4], below 1 ulp.
__device__ __forceinline__ void pm_sincos(double x, float *s, float *c)
{
    const double t = x * 0.6366197723675814;              // 2 / pi
    const double n = rint(t);
    const float r = (float)((t - n) * 1.5707963267948966);
    const int q = (int)n;
    const float z = r * r;
    const float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
    const float cp = fmaf(z * z, fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f),
                          fmaf(-0.5f, z, 1.0f));
    const bool odd = q & 1;
    const float sv = odd ? cp : sp, cv = odd ? sp : cp;
    // quadrant 0: (s, c); 1: (c, -s); 2: (-s, -c); 3: (-c, s)
    *s = (q & 2) ? -sv : sv;
    *c = ((q + 1) & 2) ? -cv : cv;
}

// (x, y) -> the float16 pair nearest towards zero as one dword, and what is left of x and y
__device__ __forceinline__ unsigned pm_split(float x, float y, float *rx, float *ry)
{
    const auto h = __builtin_amdgcn_cvt_pkrtz(x, y);
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    // x - (float)h.lo and y - (float)h.hi, each one mixed-precision fma: the float16 half is source 0 (op_sel_hi bit 0),
    // its low or high word picked by op_sel bit 0
    const float m1 = -1.0f;
    float a, b;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(a) : "v"(hb), "v"(m1), "v"(x));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(b) : "v"(hb), "v"(m1), "v"(y));
    *rx = a;
    *ry = b;
    return hb;
}
__device__ __forceinline__ unsigned pm_pack(float x, float y) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x, y)); }
// from the packed pair (x, y): (x, -y) and (y, x) -- exact: rounding toward zero is symmetric in the sign, so these are the
// words pm_pack(x, -y) and pm_pack(y, x) would give, for one conversion instead of three (the conversion is a half-rate
// instruction; the kernel is bound by vector issue)
__device__ __forceinline__ unsigned pm_conj(unsigned h) { return h ^ 0x80000000u; }
__device__ __forceinline__ unsigned pm_swap(unsigned h) { return __builtin_amdgcn_alignbit(h, h, 16); }

// Workgroup barrier for data exchanged through LDS only: every wave's LDS operations are complete (lgkmcnt 0) before it
// arrives, but its global loads stay in flight (__syncthreads() also waits for those, vmcnt 0).  Tried because the
// loads of a round always hitting the same cached entries (PM_ABL_HOTLOAD) made the kernel 14 % faster; this barrier
// changed nothing, so that gain is the data's (the same low, mostly evanescent frequencies every round), not the loads'.
__device__ __forceinline__ void pm_lds_barrier() { __syncthreads(); }

__global__ __launch_bounds__(PM_WAVES * 64, PM_CH == 16 ? 3 : 2) void ps_mfma_kernel(PsMfmaParams Q)
{
    extern __shared__ __attribute__((aligned(16))) unsigned pm_lds[];
    const PsParams &P = Q.P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = wave & (PM_NQ - 1), part = wave / PM_NQ;
    // wavenumbers from both ends of the axis in turn: small |kx| -- few evanescent frequencies, the long workgroups -- first
    // (natural order: 0 .. tnum/2 ascending |kx|, then descending: the launch ended on its longest workgroups)
    const int g = (int)blockIdx.x % Q.ngroups, bq = (int)blockIdx.x / Q.ngroups;
    const int kb = (bq & 1) ? P.nk - 1 - (bq >> 1) : (bq >> 1), k = P.k0 + kb;
    const int om = lane % PM_CH, hh = lane / PM_CH;      // frequency of the chunk; which of its PM_NSUB generating lanes
    const Cp<float> *Frow = reinterpret_cast<const Cp<float> *>(P.F) + (size_t)k * P.fstride;
    float *TKrow = reinterpret_cast<float *>(reinterpret_cast<Cp<float> *>(P.TK) + (size_t)kb * P.snum);

    // ---- scale of the row: the largest component into [2^11, 2^12)
    float sigma;
    {
        float m = 0.f;
        for (int slot = tid; slot < P.nf; slot += PM_WAVES * 64) {
            const Cp<float> f = ps_load_slot<float>(Frow, P, slot);
            m = fmaxf(m, fmaxf(fabsf(f.x), fabsf(f.y)));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float *mx = reinterpret_cast<float *>(pm_lds);
        if (lane == 0) mx[wave] = m;
        __syncthreads();
        m = mx[0];
#pragma unroll
        for (int i = 1; i < PM_WAVES; ++i) m = fmaxf(m, mx[i]);
        __syncthreads();
        int e = 0;
        (void)frexpf(m, &e);
        sigma = (m > 0.f && m < 3.0e38f) ? ldexpf(1.0f, 12 - e) : 1.0f;
    }

    // ---- tiles (32 rows of PM_ROW dwords = one complex float16 pair per frequency of the chunk, no padding).  A 256-byte
    // bank line holds PM_NSUB rows; the 16-byte slots of row r sit at slot ^ ((r / PM_NSUB) % PM_NSLOT), so that the 16
    // lanes a ds_read_b128 serves together (rows r .. r + 15, one slot) and the 64 lanes of a generating ds_write_b32
    // (PM_CH frequencies x the rows PM_NSUB j + hh) each touch all 64 banks once.  The state tiles of the group's row
    // blocks are shared by the four waves; one step-factor tile per wave; hi halves, then lo halves.
    unsigned *Aq = pm_lds + (size_t)q * PM_NRB * 2 * PM_TILE;                          // [rb][hi, lo]
    unsigned *Bhi = pm_lds + (size_t)PM_NQ * PM_NRB * 2 * PM_TILE + (size_t)wave * 2 * PM_TILE, *Blo = Bhi + PM_TILE;
    // generation: this lane writes logical dword `om` of rows PM_NSUB j + hh; (row / PM_NSUB) % PM_NSLOT = j % PM_NSLOT
    int wx[PM_NSLOT];
#pragma unroll
    for (int m = 0; m < PM_NSLOT; ++m) wx[m] = PM_ROW * hh + (om ^ (4 * m));
    // MFMA operands: row / column lane & 31, K-step s, k-half h = lane >> 5: logical slot 2 s + h; K-step s at rd0 ^ 8 s
    const int rd0 = (lane & 31) * PM_ROW + ((4 * (lane >> 5)) ^ (4 * (((lane & 31) / PM_NSUB) % PM_NSLOT)));

    // the group's row blocks: (run, first tile inside the run); block i is GENERATED by the waves of part i (i < 4;
    // block 4 by part 0 as well)
    int brun[PM_NRB], ba0[PM_NRB];
#pragma unroll
    for (int rb = 0; rb < PM_NRB; ++rb) {
        const int2 d = Q.blocks[(size_t)g * PM_NRB + rb];
        brun[rb] = __builtin_amdgcn_readfirstlane(d.x);
        ba0[rb] = __builtin_amdgcn_readfirstlane(d.y);
    }
    pm_float16 acc[PM_NRB];
#pragma unroll
    for (int rb = 0; rb < PM_NRB; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[rb][i] = 0.f;

    // this lane's step factors: steps b = 16 part + hh + PM_NSUB j of every 64-step tile
    auto gen_B = [&](double inc) {
        float e2s, e2c, bs, bc;
        pm_sincos((double)PM_NSUB * inc, &e2s, &e2c);
        pm_sincos((double)(16 * part + hh + 1) * inc, &bs, &bc);
        bs *= 256.f;
        bc *= 256.f;
#pragma unroll
        for (int j = 0; j < 16 / PM_NSUB; ++j) {
            // column b (re) holds (B_re, -B_im) against (S_re, S_im); column 16 + b (im) holds (B_im, B_re): four pack
            // conversions (the negation is a source modifier) instead of shifting and masking two packed pairs apart
            float rc, rs;
            const unsigned h0 = pm_split(bc, bs, &rc, &rs), l0 = pm_pack(rc, rs);      // (cos, sin): high halves, residuals
            const int o = 64 * j + wx[j % PM_NSLOT];
            Bhi[o] = pm_conj(h0);                                      // (cos, -sin)
            Bhi[o + 16 * PM_ROW] = pm_swap(h0);                        // (sin, cos)
            Blo[o] = pm_conj(l0);
            Blo[o + 16 * PM_ROW] = pm_swap(l0);
            const float nc = fmaf(bc, e2c, -(bs * e2s)), ns = fmaf(bc, e2s, bs * e2c);
            bc = nc;
            bs = ns;
        }
    };

    unsigned nmfma = 0;                     // (uniform) MFMA instructions this wave has issued
    const int nchunk = P.nf / PM_CH;        // a multiple of PM_NQ (host): every wave makes the same number of rounds
    // long-run index of every row block (ps_setup_kernel's table)
    int blong[PM_NRB];
#pragma unroll
    for (int rb = 0; rb < PM_NRB; ++rb) blong[rb] = brun[rb] >= 0 ? Q.long_of[brun[rb]] : 0;
    const double2 *tab = Q.runtab + (size_t)(Q.pairs ? min(kb, P.tnum - kb) : kb) * P.nf * Q.nlong;       // (phases depend on kx^2 only)
    // the next round's spectrum and run entries are requested a round ahead
    Cp<float> f_next = ps_load_slot<float>(Frow, P, q * PM_CH + om);
    double2 t_next[PM_NRB];
#pragma unroll
    for (int rb = 0; rb < PM_NRB; ++rb) t_next[rb] = tab[(size_t)(q * PM_CH + om) * Q.nlong + blong[rb]];
    for (int c = q; c < nchunk; c += PM_NQ) {
        const float f0r = f_next.x * sigma, f0i = f_next.y * sigma;
        double incs[PM_NRB], phis[PM_NRB];
#pragma unroll
        for (int rb = 0; rb < PM_NRB; ++rb) {
            incs[rb] = t_next[rb].x;
            phis[rb] = t_next[rb].y;
        }
        {
            const int sn = min(c + PM_NQ, nchunk - 1) * PM_CH + om;
            f_next = ps_load_slot<float>(Frow, P, sn);
#pragma unroll
            for (int rb = 0; rb < PM_NRB; ++rb) t_next[rb] = tab[(size_t)sn * Q.nlong + blong[rb]];
        }
        // A chunk none of whose 16 frequencies takes part in any of the group's runs adds nothing: the round is skipped
        // (round 4).  The frequencies below v kx / 2 are evanescent -- out for good, NaN start phase in the run table --
        // and they are a contiguous band of slots: 42 % of the (kx, w) plane at config 5, i.e. 42 % of the rounds.  Every
        // wave sees the same 16 frequencies x PM_NRB entries (lane % PM_CH), so all four decide alike and none meets a barrier.
        // ... and a row block whose RUN none of them takes part in (the band widens from run to run as the velocity
        // rises) is left out of phase 2, with its step factors.
        unsigned rb_in = 0;                                                  // uniform over the workgroup
        {
            const bool any_f = f0r != 0.f || f0i != 0.f;
#pragma unroll
            for (int rb = 0; rb < PM_NRB; ++rb)
                if (brun[rb] >= 0 && __builtin_amdgcn_ballot_w64(any_f && phis[rb] == phis[rb]) != 0) rb_in |= 1u << rb;
            static_assert(PM_NQ == 1, "the skip below is uniform over the WORKGROUP only while every wave walks the same chunks: "
                                      "with more frequency parts decide it through LDS / __syncthreads_or");
            if (rb_in == 0) continue;
        }
        // ---- phase 1, fused: the state tile of this wave's own block as TWO recurrences (rows 0..15 and 16..31 of the tile,
        // each from its own float64-phase anchor) and the step-factor tile, advanced together in one straight-line loop.
        // A wave's vector instructions issue ~7 cycles apart when each depends on the one before and 4 apart when they
        // do not (profiles/tools/mfma_valu_probe.hip); the separate loops (below) were three dependent chains one after
        // the other, with the stamps' branches between them.  The block's parameters are picked by the (uniform) part
        // index without branching; a wave whose block does not exist fills a tile nobody reads.
        {
            double inc_o = incs[0], phi_o = phis[0];
            int a0_o = ba0[0];
#pragma unroll
            for (int rb = 1; rb < PM_NP; ++rb) {
                const bool me = rb == part;
                inc_o = me ? incs[rb] : inc_o;
                phi_o = me ? phis[rb] : phi_o;
                a0_o = me ? ba0[rb] : a0_o;
            }
            const bool in_o = phi_o == phi_o;
            const double ph0 = in_o ? phi_o : 0.0;
            constexpr int NJ = 32 / PM_NSUB, NJH = NJ / 2;
            float sx, cx, sy, cy, Es, Ec, e2s, e2c, bs, bc;
            pm_sincos(ph0 + (double)(PM_TT * (a0_o + hh)) * inc_o, &sx, &cx);
            pm_sincos(ph0 + (double)(PM_TT * (a0_o + PM_NSUB * NJH + hh)) * inc_o, &sy, &cy);
            pm_sincos((double)(PM_NSUB * PM_TT) * inc_o, &Es, &Ec);
            pm_sincos((double)PM_NSUB * incs[0], &e2s, &e2c);
            pm_sincos((double)(16 * part + hh + 1) * incs[0], &bs, &bc);
            bs *= 256.f;
            bc *= 256.f;
            const float gr = in_o ? f0r : 0.f, gi = in_o ? f0i : 0.f;
            float xr = fmaf(gr, cx, -(gi * sx)), xi = fmaf(gr, sx, gi * cx);
            float yr = fmaf(gr, cy, -(gi * sy)), yi = fmaf(gr, sy, gi * cy);
            unsigned *Ahi = Aq + (size_t)part * 2 * PM_TILE, *Alo = Ahi + PM_TILE;
#pragma unroll
            for (int j = 0; j < NJH; ++j) {
                float rr, ri, qr, qi, rc, rs;
                const int ox = 64 * j + wx[j % PM_NSLOT], oy = 64 * (j + NJH) + wx[(j + NJH) % PM_NSLOT];
                const unsigned hx = pm_split(xr, xi, &rr, &ri), hy = pm_split(yr, yi, &qr, &qi);
                const unsigned hb = pm_split(bc, bs, &rc, &rs), lb = pm_pack(rc, rs);
                Ahi[ox] = hx;
                Ahi[oy] = hy;
                Alo[ox] = pm_pack(rr, ri);
                Alo[oy] = pm_pack(qr, qi);
                const int ob = 64 * j + wx[j % PM_NSLOT];
                Bhi[ob] = pm_conj(hb);
                Bhi[ob + 16 * PM_ROW] = pm_swap(hb);
                Blo[ob] = pm_conj(lb);
                Blo[ob + 16 * PM_ROW] = pm_swap(lb);
                const float nxr = fmaf(xr, Ec, -(xi * Es)), nxi = fmaf(xr, Es, xi * Ec);
                const float nyr = fmaf(yr, Ec, -(yi * Es)), nyi = fmaf(yr, Es, yi * Ec);
                const float nc = fmaf(bc, e2c, -(bs * e2s)), ns = fmaf(bc, e2s, bs * e2c);
                xr = nxr;
                xi = nxi;
                yr = nyr;
                yi = nyi;
                bc = nc;
                bs = ns;
            }
            static_assert(16 / PM_NSUB == 32 / PM_NSUB / 2, "the step-factor tile has as many rows per lane as half a state tile");
        }
        // the fifth block of a full group: a quarter of its rows by every wave
        if ((rb_in >> (PM_NRB - 1)) & 1u) {                                   // uniform
            constexpr int rb = PM_NRB - 1, NJQ = 32 / PM_NSUB / PM_NP;
            const double inc = incs[rb];
            const bool in = phis[rb] == phis[rb];
            float s, cph, Es, Ec;
            pm_sincos((in ? phis[rb] : 0.0) + (double)(PM_TT * (ba0[rb] + PM_NSUB * NJQ * part + hh)) * inc, &s, &cph);
            pm_sincos((double)(PM_NSUB * PM_TT) * inc, &Es, &Ec);
            const float gr = in ? f0r : 0.f, gi = in ? f0i : 0.f;
            float sr = fmaf(gr, cph, -(gi * s)), si = fmaf(gr, s, gi * cph);
            unsigned *Ahi = Aq + (size_t)rb * 2 * PM_TILE, *Alo = Ahi + PM_TILE;
#pragma unroll
            for (int p4 = 0; p4 < PM_NP; ++p4)
                if (p4 == part) {                                             // uniform
#pragma unroll
                    for (int jj = 0; jj < NJQ; ++jj) {
                        const int j = NJQ * p4 + jj;
                        float rr, ri;
                        const int o = 64 * j + wx[j % PM_NSLOT];
                        Ahi[o] = pm_split(sr, si, &rr, &ri);
                        Alo[o] = pm_pack(rr, ri);
                        const float nr = fmaf(sr, Ec, -(si * Es)), ni = fmaf(sr, Es, si * Ec);
                        sr = nr;
                        si = ni;
                    }
                }
        }
        pm_lds_barrier();           // the half's state tiles are complete
        // ---- phase 2: every row block of the group against this wave's 16 steps: 32 frequencies = 4 K-steps of 8.  The
        // operands of step s + 1 are read while the three MFMAs of step s run (a read-wait-compute sequence per step
        // left the matrix pipe idle for an LDS round trip sixteen times a round).
        // The step factors of a run stay in registers for all of its row blocks (4 K-steps x (hi, lo) = 8 x 16 bytes per
        // lane): with them re-read from LDS for every block the phase moved 16 KB per wave and block through a 256 B/clk
        // port for 384 cycles of matrix pipe -- 341 B/clk with eight waves on a CU: the LDS read port, not the pipe, set
        // its length (~2750 cycles a round against 1536-1920 of MFMA).
        uint4 bh[PM_CH / 8], bl[PM_CH / 8];
#pragma unroll
        for (int rb = 0; rb < PM_NRB; ++rb) {
            const int run = brun[rb];
            if (run < 0 || !((rb_in >> rb) & 1u)) continue;                   // uniform
            if (rb > 0 && brun[rb - 1] != run) {
                __builtin_amdgcn_wave_barrier();
                gen_B(incs[rb]);                                              // another run: other step factors
            }
            __builtin_amdgcn_wave_barrier();
            if (rb == 0 || brun[rb - 1] != run) {
#pragma unroll
                for (int s = 0; s < PM_CH / 8; ++s) {
                    const int o = rd0 ^ (8 * s);
                    bh[s] = *reinterpret_cast<const uint4 *>(Bhi + o);
                    bl[s] = *reinterpret_cast<const uint4 *>(Blo + o);
                }
            }
            const unsigned *Ahi = Aq + (size_t)rb * 2 * PM_TILE, *Alo = Ahi + PM_TILE;
            nmfma += 3 * (PM_CH / 8);
            uint4 ra_hi = *reinterpret_cast<const uint4 *>(Ahi + rd0), ra_lo = *reinterpret_cast<const uint4 *>(Alo + rd0);
#pragma unroll
            for (int s = 0; s < PM_CH / 8; ++s) {
                const pm_half8 a_hi = __builtin_bit_cast(pm_half8, ra_hi), a_lo = __builtin_bit_cast(pm_half8, ra_lo);
                const pm_half8 b_hi = __builtin_bit_cast(pm_half8, bh[s]), b_lo = __builtin_bit_cast(pm_half8, bl[s]);
                if (s + 1 < PM_CH / 8) {
                    const int o = rd0 ^ (8 * (s + 1));
                    ra_hi = *reinterpret_cast<const uint4 *>(Ahi + o);
                    ra_lo = *reinterpret_cast<const uint4 *>(Alo + o);
                }
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, acc[rb], 0, 0, 0);
            }
        }
        pm_lds_barrier();           // ... and read by everybody before the next round overwrites them
    }

    if (lane == 0 && Q.mfma_count) atomicAdd(Q.mfma_count, (unsigned long long)nmfma);
    // ---- every wave owns its sums outright (all frequencies of its 16 steps): TK /= snum (:492) and store.
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


// ---------------------------------------------------------------------------
// Set-up pass, one workgroup per wavenumber -- per PAIR of wavenumbers (k, tnum - k: the same kx^2, the same phases, one
// table row) when the call has the whole symmetric axis --, one thread per frequency at a time: everything that is float64 and per
// (wavenumber, frequency, run) is done HERE, once, at full lane efficiency, instead of by the 8 lanes per frequency
// of ps_mfma_kernel in every round (two float64 divisions, a square root and the catch-up over the runs per
// frequency were the larger part of that kernel's vector work, profiles/r03_ps_mfma_*.txt):
//   * the phase per step of every run (constant velocity: :411-415; v(z): :456-460) and the phase at the start of every
//     LONG run -> runtab; a frequency that is evanescent at a run's velocity is out from there on (:484-485: off the
//     boundary band the reference's threshold (tau/tt_end/1e6)^2 is the sign of coss);
//   * the boundary frequencies (|coss| < 1e-8 in ANY run: kept or dropped by the reference at every step's own
//     velocity) take no part in any run and are listed for ps_edge_kernel;
//   * the depth steps of the SHORT runs (len <= PM_SHORT: a layer boundary smeared over a few steps) are summed over
//     the frequencies directly -- FK0 e^{i Phi} per (frequency, step) -- and stored to TK (the row blocks of the long
//     runs do not cover them).
// ---------------------------------------------------------------------------
template <int PM_SM>      // frequencies per thread: nf <= 512 PM_SM
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(PM_SM <= 4 ? 4 : (PM_SM <= 8 ? 3 : 2)))) void ps_setup_kernel(PsMfmaParams Q)
{
    extern __shared__ __attribute__((aligned(16))) unsigned pm_lds[];
    const PsParams &P = Q.P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Q.pairs: one workgroup per PAIR of wavenumbers (rows k and tnum - k: the same kx^2, the same phases) -- one walk over
    // the runs, one table row, the single steps summed for both spectra
    const int k = P.k0 + (int)blockIdx.x, k2 = Q.pairs ? (P.tnum - (int)blockIdx.x) % P.tnum : k;
    const bool has2 = k2 != k;
    float *red = reinterpret_cast<float *>(pm_lds);                         // [2][8][2 * PM_SHORT]
    const Cp<float> *Frow = reinterpret_cast<const Cp<float> *>(P.F) + (size_t)k * P.fstride;
    const Cp<float> *Frow2 = reinterpret_cast<const Cp<float> *>(P.F) + (size_t)k2 * P.fstride;
    float *TKrow = reinterpret_cast<float *>(reinterpret_cast<Cp<float> *>(P.TK) + (size_t)(k - P.k0) * P.snum);
    float *TKrow2 = reinterpret_cast<float *>(reinterpret_cast<Cp<float> *>(P.TK) + (size_t)(k2 - P.k0) * P.snum);
    double2 *tab = Q.runtab + (size_t)(k - P.k0) * P.nf * Q.nlong;
    const double kxk = P.kx[k];
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    if (!Q.vz) {
        // constant velocity: one run, FK e^{i (tau + 1) phi} for the propagating frequencies (:411-420)
        for (int slot = tid; slot < P.nf; slot += 512) {
            const double w = P.w[slot];
            const double vk = P.vconst * kxk / 2.0;
            const double vkx2 = vk * vk;                                      // :411
            const bool prop = vkx2 < w * w;                                   // :412
            tab[slot] = make_double2(prop ? w * P.dt * sqrt(1.0 - vkx2 / (w * w)) : 0.0, prop ? 0.0 : nan);   // :415
        }
        return;
    }
    // Every thread keeps its frequencies (slot = tid + 512 m) in registers and walks the runs with the m loop unrolled:
    // no LDS round trips (the first form kept phase / 1/w / spectrum in 24 nf bytes of LDS: 2.08 ms at config 5, this one
    // 1.83; a branch-free body -- selects instead of the two ifs -- measured no better: the pass is bound by its ~450
    // instructions per (wavenumber, frequency) of the nine runs of that table, six of them one-step runs with a sincos each).
    double ph[PM_SM], rw[PM_SM], wv[PM_SM];
    float2 fv[PM_SM], fw[PM_SM];
#pragma unroll
    for (int m = 0; m < PM_SM; ++m) {
        const int slot = tid + 512 * m;
        const bool in = slot < P.nf;
        const Cp<float> f = in ? ps_load_slot<float>(Frow, P, slot) : Cp<float>{0.f, 0.f};
        const Cp<float> f2 = (in && has2) ? ps_load_slot<float>(Frow2, P, slot) : Cp<float>{0.f, 0.f};
        wv[m] = in ? P.w[slot] : 1.0;
        rw[m] = 1.0 / wv[m];
        bool edge = false;
        for (int r = 0; r < Q.nruns; ++r) edge = edge || fabs(pm_coss(Q.runs[r].v, kxk, rw[m])) < 1e-8;
        if (edge && in) {
            const int at = atomicAdd(Q.edge_cnt + k, 1);
            if (at < PM_EMAX) Q.edge_list[(size_t)k * PM_EMAX + at] = slot;
            if (has2) {
                const int at2 = atomicAdd(Q.edge_cnt + k2, 1);
                if (at2 < PM_EMAX) Q.edge_list[(size_t)k2 * PM_EMAX + at2] = slot;
            }
        }
        // (a frequency whose spectrum value IS zero stays in: only the NaN phase says "out")
        fv[m] = make_float2(f.x, f.y);
        fw[m] = make_float2(f2.x, f2.y);
        ph[m] = (edge || !in) ? nan : 0.0;                                    // NaN phase = out of every run from here on
    }
    for (int r = 0; r < Q.nruns; ++r) {
        const double v = Q.runs[r].v;
        const int len = Q.runs[r].len, start = Q.runs[r].start;
        const bool is_short = len <= PM_SHORT;                                // uniform
        const int L = Q.long_of[r];
        float acc[2 * PM_SHORT], acc2[2 * PM_SHORT];
#pragma unroll
        for (int j = 0; j < 2 * PM_SHORT; ++j) acc[j] = acc2[j] = 0.f;
#pragma unroll
        for (int m = 0; m < PM_SM; ++m) {
            const int slot = tid + 512 * m;
            const double cs = pm_coss(v, kxk, rw[m]);                         // :456
            double inc = 0.0;
            if (cs <= 0.0) ph[m] = nan;                                       // :484-485, for good
            else inc = wv[m] * P.dt * pm_sqrt01(cs);                          // :458-460 (off the boundary band: cs >= 1e-8)
            if (L >= 0 && slot < P.nf) tab[(size_t)slot * Q.nlong + L] = make_double2(inc, ph[m]);
            if (is_short && ph[m] == ph[m]) {
#pragma unroll
                for (int j = 0; j < PM_SHORT; ++j)
                    if (j < len) {                                            // uniform
                        float sn, c2;
                        pm_sincos(ph[m] + (double)(j + 1) * inc, &sn, &c2);
                        acc[2 * j] += fmaf(fv[m].x, c2, -(fv[m].y * sn));    // :464, :487
                        acc[2 * j + 1] += fmaf(fv[m].x, sn, fv[m].y * c2);
                        acc2[2 * j] += fmaf(fw[m].x, c2, -(fw[m].y * sn));
                        acc2[2 * j + 1] += fmaf(fw[m].x, sn, fw[m].y * c2);
                    }
            }
            ph[m] = pm_wrap(ph[m] + (double)len * inc);                       // NaN stays NaN
        }
        if (is_short) {
#pragma unroll
            for (int j = 0; j < 2 * PM_SHORT; ++j) {
                float x = acc[j], y = acc2[j];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    x += __shfl_xor(x, o, 64);
                    y += __shfl_xor(y, o, 64);
                }
                if (lane == 0) {
                    red[wave * 2 * PM_SHORT + j] = x;
                    red[(8 + wave) * 2 * PM_SHORT + j] = y;
                }
            }
            __syncthreads();
            if (tid < 2 * len) {
                float sum = 0.f, sum2 = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    sum += red[i * 2 * PM_SHORT + tid];
                    sum2 += red[(8 + i) * 2 * PM_SHORT + tid];
                }
                TKrow[2 * (size_t)(start + (tid >> 1)) + (tid & 1)] = sum / (float)P.snum;      // :492
                if (has2) TKrow2[2 * (size_t)(start + (tid >> 1)) + (tid & 1)] = sum2 / (float)P.snum;
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------
// The boundary frequencies ps_mfma_kernel listed, one workgroup per wavenumber: every listed frequency is walked over
// the whole depth axis the reference's way, in float64 -- coss from every step's own velocity, FK *= e^{i w dt
// Re sqrt(coss)}, zero for good once coss <= (tau/tt_end/1e6)^2 (mig_python.py:456-487) -- with the phase as a prefix
// sum over the steps (each thread owns a contiguous stretch of steps), and added to TK.  Slots in ascending order:
// the result does not depend on the order the frequency kernel found them in.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ps_edge_kernel_t(PsMfmaParams Q)
{
    const PsParams &P = Q.P;
    const int k = P.k0 + (int)blockIdx.x, tid = threadIdx.x;
    int n = Q.edge_cnt[k];
    if (n <= 0) return;
    n = min(n, PM_EMAX);
    __shared__ int slots[PM_EMAX];
    __shared__ double part[256];
    __shared__ int dead_at[256];
    if (tid == 0) {
        for (int i = 0; i < n; ++i) slots[i] = Q.edge_list[(size_t)k * PM_EMAX + i];
        for (int i = 1; i < n; ++i)
            for (int j = i; j > 0 && slots[j - 1] > slots[j]; --j) {
                const int t = slots[j];
                slots[j] = slots[j - 1];
                slots[j - 1] = t;
            }
    }
    __syncthreads();
    Cp<T> *TKrow = reinterpret_cast<Cp<T> *>(P.TK) + (size_t)(k - P.k0) * P.snum;
    const double kxk = P.kx[k];
    const int seg = (P.snum + 255) / 256;
    const int lo = min(tid * seg, P.snum), hi = min(lo + seg, P.snum);
    for (int e = 0; e < n; ++e) {
        const int slot = slots[e];
        const Cp<T> f = ps_load_slot_k<T>(P, k, slot);
        const double w = P.w[slot];
        // pass 1: this stretch's phase and its first dead step
        double sum = 0.0;
        int dead = P.snum;
        for (int t = lo; t < hi; ++t) {
            const double a = 0.5 * P.vz[t] * kxk / w;                          // :456
            const double cs = 1.0 - a * a;
            sum += w * P.dt * (cs > 0.0 ? sqrt(cs) : 0.0);                     // :458-460
            if (cs <= P.thr[t] && dead == P.snum) dead = t;                    // :484-485
        }
        part[tid] = sum;
        dead_at[tid] = dead;
        __syncthreads();
        double before = 0.0;
        int first_dead = P.snum;
        for (int i = 0; i < 256; ++i) {
            if (i < tid) before += part[i];
            first_dead = min(first_dead, dead_at[i]);
        }
        // pass 2: the steps themselves
        double ph = before;
        for (int t = lo; t < hi; ++t) {
            const double a = 0.5 * P.vz[t] * kxk / w;
            const double cs = 1.0 - a * a;
            ph += w * P.dt * (cs > 0.0 ? sqrt(cs) : 0.0);
            if (t < first_dead) {
                double sn, c2;
                sincos(pm_wrap(ph), &sn, &c2);
                const double re = (double)f.x * c2 - (double)f.y * sn, im = (double)f.x * sn + (double)f.y * c2;   // :464
                TKrow[t].x += (T)(re / (double)P.snum);                     // :487, :492
                TKrow[t].y += (T)(im / (double)P.snum);
            }
        }
        __syncthreads();
    }
}
