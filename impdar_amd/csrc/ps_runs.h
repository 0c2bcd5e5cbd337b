// Phase-shift frequency sum for MANY runs of constant velocity (float32 data; included by phaseshift.hip).
//
// Reference: mig_python.py:438-487 with the vmig that getVelocityProfile (:582-604) makes of an N-row (v, z) table: N - 1
// layers of constant velocity, each boundary smeared over three or four single steps by 2 * gradient(z(t)).  ps_mfma.h
// cuts a run into 64-step tiles and row blocks of 2048 steps -- right for 2-4 thick layers, mostly padding for layers of
// 100-400 steps: tables of more than 16 runs fell back to the vector kernels (41 / 81 rows at 8192^2: 46.6 / 62.0 ms
// against 9.4 for four rows).  This kernel is built for the short layers:
//
//   * tiles of 8 steps, blocks of 16 tiles: inside a run  step = 8 a + b,
//         TK[start + 8 a + b] = sum_w [2^8 e^{i Phi_w} e^{i 8 a phi_w}] [sigma F_w e^{i (b + 1) phi_w}] = sum_w A_w(a) G_w(b),
//     one 16 x 16 accumulator = 16 tiles x (8 steps x (re, im)) = 128 steps per block, v_mfma_f32_16x16x32_f16 with the
//     float16 hi / lo operands of ps_mfma.h (x = hi + lo, three products hi.hi + hi.lo + lo.hi; the spectrum scaled by a
//     power of two into [2^11, 2^12), the rotations by 2^8).  The first form of this kernel used float32 operands
//     (v_mfma_f32_16x16x4_f32: no split, 4 vector instructions per state row instead of 8) -- but a float32 MFMA runs at
//     the VECTOR rate and keeps the SIMD's vector unit for its 32 cycles (SQ_VALU_MFMA_COEXEC_CYCLES = 0): its 10.9 ms of
//     pipe time at 41 rows came on top of the vector work instead of under it (profiles/r05_ps_runs.txt);
//   * the spectrum sits in the STEP FACTORS G, the state rows A are pure rotations -- and rotations depend on the wavenumber
//     through kx^2 only (:456-460): a workgroup takes the rows k and tnum - k (kx and -kx) TOGETHER, one set-up walk and
//     one set of state rows (16-64 per run piece) for both, 8 step factors per frequency and run for each.  The kernel is
//     bound by vector issue (63 % busy, 302 vector instructions for 24 MFMAs per chunk before): 41 rows 22.7 -> 15.3 ms;
//   * phases generated in the kernel: a thread owns ONE frequency at a time and walks the stage's runs in order with
//     the phase in a float64 register -- coss, the square root (v_rsq_f64 seed + one Newton step), the phase at the
//     run's start, and for the run three sincos: the anchor rotation, the tile rotation e^{i 8 phi}, the step
//     rotation e^{i phi}.  No table in HBM (ps_mfma.h: 16 bytes per wavenumber, frequency and run), no limit on runs;
//   * single steps between the layers are ROWS of their own: the rotation at that step against a 4-column matrix of the
//     two spectra (re, im), up to 12 of them in one more accumulator per stage -- the sum over the frequencies still runs
//     on the matrix pipe.
//
// Work split.  A workgroup = (pair of wavenumbers, PART of the spectrum: 1024 frequency slots, 4 per thread); its sums over
// the part go to a partial image [part][k][tau], added in order by ps_smooth_sum_kernel (deterministic; one part: straight
// to TK).  The depth axis is cut into STAGES of up to four long runs (<= 512 steps each: longer runs are cut) and twelve
// single steps; per stage and super-chunk of 256 frequencies: (1) every thread sets up its frequency for the stage's
// runs -> LDS; (2) wave p multiplies ITS long run: per chunk of 32 frequencies it generates the step-factor tiles of both
// wavenumbers and block after block the state tile (two lanes per frequency, rows by A *= e^{i 16 phi}) in its own LDS
// tile, 12 MFMAs per block and pair; every wave takes a quarter of the single steps' frequencies.  Accumulators: 2 x 4
// blocks x 2 + 1 per wave.  Frequencies on the evanescent boundary of some run take no part and are listed for
// ps_edge_kernel, as in ps_mfma.h; a frequency that has turned evanescent is out for good (NaN phase), chunks of 32 dead
// frequencies are skipped.
#pragma once

typedef float pr_float4 __attribute__((ext_vector_type(4)));

constexpr int PR_TT = 8;            // depth steps per tile
constexpr int PR_ROWS = 16;         // tiles per block
constexpr int PR_NM = 4;            // frequencies per thread (super-chunks of 256 slots per part)
constexpr int PR_PART = PR_NM * 256;
constexpr int PR_LONGS = 4;         // long runs per stage = waves
constexpr int PR_NBLK = 4;          // blocks per long run: up to 512 steps
constexpr int PR_LONG_MAX = PR_NBLK * PR_ROWS * PR_TT;
constexpr int PR_SHORT_LEN = 2;     // runs of up to this many steps: every step a row of its own
constexpr int PR_SROWS = 12;        // single-step rows per stage
constexpr int PR_STAGE_RUNS = 16;   // runs per stage
constexpr int PR_LD = 32;            // dwords per tile row: 32 frequencies, one float16 pair each; the eight 16-byte slots of row r sit at slot ^ (r & 7):
                                     // both the generating ds_write_b32 and the operands' ds_read_b128 touch every bank once (a padded row of 36
                                     // dwords: SQ_LDS_BANK_CONFLICT 23 % of the LDS cycles, profiles/r05_ps_runs.txt)
constexpr int PR_TILE = PR_ROWS * PR_LD;    // dwords per tile (hi or lo halves)
constexpr size_t pr_lds_bytes(int nmem)      // see the kernel's layout
{
    return 4 * (size_t)(3 * 2 * PR_LONGS * 8 * 32 + nmem * 2 * 8 * 32 + 2 * 8 * 4 * 32 + 2 * 8 * PR_SROWS * PR_LD + PR_LONGS * 2 * PR_TILE + 16 + PR_STAGE_RUNS * 6);
}
static_assert(pr_lds_bytes(2) <= 80 * 1024, "two workgroups per CU");

struct PrRun {
    double v;               // velocity
    int start, len;         // first depth step, steps
    int kind;               // 0: long (tiles of 8 steps), 1: single steps
    int slot;               // long: the wave that multiplies it; short: its first row among the stage's single-step rows
};
static_assert(sizeof(PrRun) == 24, "the kernel copies a stage's runs to LDS as 6 words each");
struct PrStage {
    int run0, nruns;        // the stage's runs, in depth order
    int nshort, short_wave; // single-step rows and the wave that multiplies them
    int long_run[PR_LONGS]; // run of wave p (-1: none)
    int long_nblk[PR_LONGS];
    int short_tau[PR_SROWS];
};
struct PrParams {
    PsParams P;
    const PrRun *runs;
    const PrStage *stages;
    const double *rw;       // [nf] 1 / w
    int nstages, nruns, nparts;
    void *part;             // [nparts][nk][snum] complex float32 partial images (nparts > 1)
    int *edge_cnt, *edge_list;
    unsigned long long *mfma_count;     // MFMA instructions issued, summed over the launch (bench.py: mfma_flop_executed)
};

// sqrt(x), x in [1e-8, 1]: v_rsq_f64 as the seed and ONE Newton step that carries h ~ 1 / (2 y) (relative error of the
// seed squared: < 1e-15).  pm_sqrt01 goes through float32 (two conversions and v_rsq_f32: three quarter-rate
// instructions) and takes two steps; in this kernel's set-up, where every instruction of a float64 chain waits for the one
// before, that was a third of a run's time.
__device__ __forceinline__ double pr_sqrt01(double x)
{
    const double r = __builtin_amdgcn_rsq(x);
    const double y = x * r, h = 0.5 * r;
    return fma(fma(-y, y, x), h, y);
}

// pm_sincos with the quadrant found by adding and subtracting 1.5 * 2^52 (the integer is then the sum's low word: no
// v_rndne_f64, no v_cvt_i32_f64 -- quarter-rate instructions both); |x| < 1e9
__device__ __forceinline__ void pr_sincos(double x, float *s, float *c)
{
    const double t = x * 0.6366197723675814;              // 2 / pi
    const double tm = t + 6755399441055744.0;
    const int q = __double2loint(tm);
    const double n = tm - 6755399441055744.0;
    const float r = (float)((t - n) * 1.5707963267948966);
    const float z = r * r;
    const float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
    const float cp = fmaf(z * z, fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f),
                          fmaf(-0.5f, z, 1.0f));
    const bool odd = q & 1;
    const float sv = odd ? cp : sp, cv = odd ? sp : cp;
    *s = (q & 2) ? -sv : sv;
    *c = ((q + 1) & 2) ? -cv : cv;
}

// dword of frequency f (0..31) in row r of a tile; first dword of the 16-byte slot of frequencies 4 q .. 4 q + 3
__device__ __forceinline__ int pr_at(int r, int f) { return r * PR_LD + ((((f >> 2) ^ r) & 7) << 2) + (f & 3); }
__device__ __forceinline__ int pr_slot(int r, int q) { return r * PR_LD + (((q ^ r) & 7) << 2); }

__device__ __forceinline__ float2 pr_cmul(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}

// NMEM = 2: the workgroup takes the wavenumbers kx and -kx (rows k and tnum - k) together.  Every phase depends on kx^2
// only (mig_python.py:456-460), so the set-up walk and the state rows -- pure rotations here, 2^8 e^{i theta}: the spectrum
// sits in the STEP FACTORS, G_w(b) = sigma F_w e^{i (b + 1) phi_w} -- are made once for both; per wavenumber there are only
// the 8 step factors per (frequency, run) and its own MFMAs.  NMEM = 1: a slab of a kx-sharded run (no mirror rows).
template <int NMEM>
__global__ __launch_bounds__(256, 2) void ps_runs_kernel(PrParams Q)
{
    extern __shared__ __attribute__((aligned(16))) float pr_lds[];
    const PsParams &P = Q.P;
    const int tid = threadIdx.x, lane = tid & 63, p = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block -> (wavenumber(s), part): the blocks of ONE part are consecutive -- blocks are dealt round robin to the 8 XCDs, and
    // with the parts of a wavenumber side by side (part = block % 4) every XCD saw one part only: the low parts are mostly
    // evanescent and leave early, so four XCDs did most of the work (first form: 1.4 waves per SIMD on average, 43 ms).
    // The high, long-lived parts first; inside a part small |kx| (the long workgroups) first.
    const int nkb = NMEM == 2 ? P.tnum / 2 + 1 : P.nk;
    const int part = Q.nparts - 1 - (int)blockIdx.x / nkb, bq = (int)blockIdx.x % nkb;
    int kbm[NMEM];                                     // rows of the wavenumber slab
    bool has[NMEM];
    if (NMEM == 2) {
        kbm[0] = bq;
        kbm[NMEM - 1] = (P.tnum - bq) % P.tnum;
        has[0] = true;
        has[NMEM - 1] = kbm[NMEM - 1] != kbm[0];       // k = 0 and the Nyquist row are their own partners
    } else {
        kbm[0] = (bq & 1) ? P.nk - 1 - (bq >> 1) : (bq >> 1);
        has[0] = true;
    }
    const int k = P.k0 + kbm[0];
    // LDS: per long run (wave) and chunk the anchor rotation / tile rotation / step rotation of every frequency (float32); the
    // super-chunk's spectra (float32, and as float16 hi / lo step-factor rows for the single steps); the single-step state
    // tiles of the 8 chunks; one operand tile per wave; the chunks' alive flags; the stage's runs
    float2 *TS = reinterpret_cast<float2 *>(pr_lds);
    float2 *TA = TS + PR_LONGS * 8 * 32, *TB = TA + PR_LONGS * 8 * 32, *TF = TB + PR_LONGS * 8 * 32;    // TF [member][chunk][32]
    unsigned *FHh = reinterpret_cast<unsigned *>(TF + NMEM * 8 * 32), *FHl = FHh + 8 * 4 * 32;          // [chunk][row 2 member + (re, im)][32]
    // single-step state tiles of the 8 chunks, hi halves then lo halves: [chunk][row][PR_LD] dwords (one float16 pair each)
    unsigned *SHh = FHl + 8 * 4 * 32, *SHl = SHh + 8 * PR_SROWS * PR_LD;
    // this wave's tile (hi, lo): the step factors of a chunk first -- [column][PR_LD] -- then, once those are in registers,
    // the state rows of its blocks -- [tile][PR_LD]
    unsigned *Wh = SHl + 8 * PR_SROWS * PR_LD + (size_t)p * 2 * PR_TILE, *Wl = Wh + PR_TILE;
    int *alive = reinterpret_cast<int *>(SHl + 8 * PR_SROWS * PR_LD + PR_LONGS * 2 * PR_TILE);
    // the stage's runs, copied here once per stage (read from global memory inside the set-up walk every iteration waited a
    // scalar-load round trip: 16 runs x ~250 cycles per walk)
    PrRun *sruns = reinterpret_cast<PrRun *>(alive + 16);

    const Cp<float> *Frow[NMEM];
#pragma unroll
    for (int mm = 0; mm < NMEM; ++mm) Frow[mm] = reinterpret_cast<const Cp<float> *>(P.F) + (size_t)(P.k0 + kbm[mm]) * P.fstride;
    const double kxk = P.kx[k];
    const double nan = __longlong_as_double(0x7ff8000000000000LL);

    // ---- this thread's frequencies: slot part * 1024 + 256 m + tid.  The arrays are ROTATED after every super-chunk, so
    // the code below only ever indexes element 0 (a run-time index into registers would be a scratch array)
    double phi[PR_NM], rw[PR_NM], wdt[PR_NM], incq[PR_NM];     // incq / rbq: phase per step and step rotation of the last long run
    float2 F[NMEM][PR_NM], rbq[PR_NM];
    bool edge[PR_NM], inpart[PR_NM];
    float fmx[NMEM];
#pragma unroll
    for (int mm = 0; mm < NMEM; ++mm) fmx[mm] = 0.f;
#pragma unroll
    for (int m = 0; m < PR_NM; ++m) {
        const int slot = part * PR_PART + 256 * m + tid;
        const bool in = slot < P.nf;
#pragma unroll
        for (int mm = 0; mm < NMEM; ++mm) {
            const Cp<float> f = (in && has[mm]) ? ps_load_slot<float>(Frow[mm], P, slot) : Cp<float>{0.f, 0.f};
            F[mm][m] = make_float2(f.x, f.y);
            fmx[mm] = fmaxf(fmx[mm], fmaxf(fabsf(f.x), fabsf(f.y)));
        }
        rw[m] = in ? Q.rw[slot] : 1.0;
        wdt[m] = in ? P.w[slot] * P.dt : 0.0;
        edge[m] = false;
        inpart[m] = in;
        incq[m] = 0.0;
        rbq[m] = make_float2(1.f, 0.f);
    }
    // scale of a row's part of the spectrum: the largest component into [2^11, 2^12) (float16 operands; exact, divided out at the end)
    float sigma[NMEM];
    {
        float *mx = reinterpret_cast<float *>(alive);
#pragma unroll
        for (int mm = 0; mm < NMEM; ++mm) {
            float v = fmx[mm];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
            if (lane == 0) mx[4 * mm + p] = v;
        }
        __syncthreads();
#pragma unroll
        for (int mm = 0; mm < NMEM; ++mm) {
            const float v = fmaxf(fmaxf(mx[4 * mm], mx[4 * mm + 1]), fmaxf(mx[4 * mm + 2], mx[4 * mm + 3]));
            int e = 0;
            (void)frexpf(v, &e);
            sigma[mm] = (v > 0.f && v < 3.0e38f) ? ldexpf(1.0f, 12 - e) : 1.0f;
#pragma unroll
            for (int m = 0; m < PR_NM; ++m) F[mm][m] = make_float2(F[mm][m].x * sigma[mm], F[mm][m].y * sigma[mm]);
        }
        __syncthreads();
    }
    // boundary frequencies (|coss| < 1e-8 at ANY run's velocity: kept or dropped by the reference at every step's own
    // velocity, :456-485) take no part here: listed for ps_edge_kernel (as ps_setup_kernel does).  Runs in the outer loop:
    // one scalar load per distinct velocity for the thread's four frequencies.
    {
        double vprev = 0.0;
        for (int r = 0; r < Q.nruns; ++r) {
            const double v = Q.runs[r].v;
            if (v == vprev) continue;                                         // uniform (pieces of one run, equal layers)
            vprev = v;
#pragma unroll
            for (int m = 0; m < PR_NM; ++m) edge[m] = edge[m] || fabs(pm_coss(v, kxk, rw[m])) < 1e-8;
        }
    }
#pragma unroll
    for (int m = 0; m < PR_NM; ++m) {
        if (edge[m] && inpart[m]) {
#pragma unroll
            for (int mm = 0; mm < NMEM; ++mm)
                if (has[mm]) {
                    const int kk_ = P.k0 + kbm[mm];
                    const int at = atomicAdd(Q.edge_cnt + kk_, 1);
                    if (at < PM_EMAX) Q.edge_list[(size_t)kk_ * PM_EMAX + at] = part * PR_PART + 256 * m + tid;
                }
        }
        phi[m] = (edge[m] || !inpart[m]) ? nan : 0.0;                         // NaN phase = out of every run from here on
    }
    Cp<float> *out[NMEM];
#pragma unroll
    for (int mm = 0; mm < NMEM; ++mm)
        out[mm] = reinterpret_cast<Cp<float> *>(Q.nparts > 1 ? Q.part : P.TK) + ((size_t)(Q.nparts > 1 ? part : 0) * P.nk + kbm[mm]) * P.snum;
    {
        // a part all of whose frequencies are evanescent at the FIRST run's velocity already (the low band) adds nothing
        bool any = false;
#pragma unroll
        for (int m = 0; m < PR_NM; ++m) any = any || (phi[m] == phi[m] && pm_coss(Q.runs[0].v, kxk, rw[m]) > 0.0);
        if (!__syncthreads_or(any)) {
#pragma unroll
            for (int mm = 0; mm < NMEM; ++mm)
                if (has[mm])
                    for (int t = tid; t < P.snum; t += 256) out[mm][t] = Cp<float>{0.f, 0.f};
            return;
        }
    }
    // (:492; with parts ps_smooth_sum_kernel divides by snum); the operands' scales divided out
    float scale[NMEM];
#pragma unroll
    for (int mm = 0; mm < NMEM; ++mm) scale[mm] = (Q.nparts > 1 ? 1.0f : 1.0f / (float)P.snum) / (sigma[mm] * 256.0f);

    const int cc = tid >> 5, f = tid & 31;           // set-up roles: chunk of the super-chunk, frequency of the chunk
    const int gf = lane & 31, hh = lane >> 5;        // generation roles: frequency, which of its two lanes
    const int orow = lane & 15, kk = lane >> 4;      // MFMA operand roles: row (A) / column (B), K slot
    const int ob = orow >> 1, onc = orow & 1;        // ... the column's step and component

    unsigned nmfma = 0;                              // (uniform) MFMA instructions this wave has issued
    for (int j = 0; j < Q.nstages; ++j) {
        const PrStage *st = Q.stages + j;
        const int run0 = st->run0, nruns = st->nruns, nshort = st->nshort;
        const int my_run = st->long_run[p], my_nblk = st->long_nblk[p];
        if (tid < nruns * 6) reinterpret_cast<unsigned *>(sruns)[tid] = reinterpret_cast<const unsigned *>(Q.runs + run0)[tid];
        __syncthreads();
        pr_float4 acc[NMEM][PR_NBLK][2], accs;
#pragma unroll
        for (int mm = 0; mm < NMEM; ++mm)
#pragma unroll
            for (int b = 0; b < PR_NBLK; ++b) acc[mm][b][0] = acc[mm][b][1] = pr_float4{0.f, 0.f, 0.f, 0.f};
        accs = pr_float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int m = 0; m < PR_NM; ++m) {
            // ---- (1) set-up: this thread's frequency through the stage's runs.  A long run costs the square root and two or
            // three sincos (anchor rotation, tile rotation, step rotation).  The single steps between two layers are CHAINED: the
            // first of a group is anchored with a sincos of the float64 phase, every step then turns the state by
            // e^{i phi*} = (step rotation of the last long run) e^{i (phi* - phi_run)} -- the difference is small (the
            // velocity moves by a per cent from layer to layer), its sine and cosine come from their series -- and the state
            // after the group's last step IS the anchor of the next long run (three rotations in float32; the next group
            // starts from the float64 phase again).  First form: a sincos per single step -- the set-up was 63 % of the
            // kernel's vector instructions at 41 rows.
            {
                double ph = phi[0], incp = incq[0];
                float2 rBp = rbq[0];
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(ph == ph);
                if (f == 0) alive[cc] = ((hh ? (bal >> 32) : bal) & 0xffffffffull) != 0ull;
                // the super-chunk's spectra: float32 for the long runs' step factors, float16 hi / lo rows (F_re, -F_im) and
                // (F_im, F_re) for the single steps' (a 4-column step-factor matrix: both wavenumbers, re and im)
#pragma unroll
                for (int mm = 0; mm < NMEM; ++mm) {
                    TF[(mm * 8 + cc) * 32 + f] = F[mm][0];
                    float rr, ri;
                    const unsigned h0 = pm_split(F[mm][0].x, F[mm][0].y, &rr, &ri), l0 = pm_pack(rr, ri);
                    FHh[(cc * 4 + 2 * mm) * 32 + f] = pm_conj(h0);
                    FHh[(cc * 4 + 2 * mm + 1) * 32 + f] = pm_swap(h0);
                    FHl[(cc * 4 + 2 * mm) * 32 + f] = pm_conj(l0);
                    FHl[(cc * 4 + 2 * mm + 1) * 32 + f] = pm_swap(l0);
                }
                float2 chain = make_float2(0.f, 0.f);                           // 2^8 e^{i phase}
                bool chain_valid = false;                                       // uniform
                // (a wave all of whose 64 frequencies are out skips the walk: their chunks are skipped below too)
                for (int rl = 0; rl < (bal != 0ull ? nruns : 0); ++rl) {       // uniform
                    const PrRun *R = sruns + rl;
                    const double v = R->v;
                    const int len = R->len, kind = R->kind, slot = R->slot;
                    const double cs = pm_coss(v, kxk, rw[0]);                  // :456
                    double inc = 0.0;
                    if (cs <= 0.0) ph = nan;                                   // :484-485, for good
                    else inc = wdt[0] * pr_sqrt01(cs);                         // :458-460 (off the boundary band: cs >= 1e-8)
                    const bool in = ph == ph;
                    if (kind == 0) {                                           // uniform
                        float2 S = chain;
                        if (!chain_valid) {                                     // uniform
                            float s, c;
                            pr_sincos(in ? ph : 0.0, &s, &c);
                            S = make_float2(256.f * c, 256.f * s);
                        }
                        if (!in) S = make_float2(0.f, 0.f);
                        float as, ac, bs, bc;
                        pr_sincos(8.0 * inc, &as, &ac);
                        pr_sincos(inc, &bs, &bc);
                        const int idx = (slot * 8 + cc) * 32 + f;
                        TS[idx] = S;
                        TA[idx] = make_float2(ac, as);
                        TB[idx] = make_float2(bc, bs);
                        incp = inc;
                        rBp = make_float2(bc, bs);
                        chain_valid = false;
                        ph += (double)len * inc;                               // NaN stays NaN
                    } else {
                        for (int s_ = 0; s_ < len; ++s_) {                     // uniform, len <= PR_SHORT_LEN
                            if (!chain_valid) {                                 // uniform: the first single step of a group
                                float s, c;
                                pr_sincos(in ? ph : 0.0, &s, &c);
                                chain = make_float2(256.f * c, 256.f * s);
                                chain_valid = true;
                            }
                            const float dl = (float)(inc - incp);
                            const bool exact = in && !(incp != 0.0 && fabsf(dl) <= 0.25f);
                            const float h = dl * dl;
                            const float dc = fmaf(h, fmaf(h, fmaf(h, -1.0f / 720.0f, 1.0f / 24.0f), -0.5f), 1.0f);
                            const float ds = dl * fmaf(h, fmaf(h, 1.0f / 120.0f, -1.0f / 6.0f), 1.0f);
                            float2 rot = pr_cmul(rBp, make_float2(dc, ds));
                            if (__builtin_amdgcn_ballot_w64(exact) != 0ull) {   // no long run before it, or a large step in phi
                                float s, c;
                                pr_sincos(inc, &s, &c);
                                if (exact) rot = make_float2(c, s);
                            }
                            chain = in ? pr_cmul(chain, rot) : make_float2(0.f, 0.f);      // :464
                            const int row = slot + s_;
                            float rr, ri;
                            SHh[cc * PR_SROWS * PR_LD + pr_at(row, f)] = pm_split(chain.x, chain.y, &rr, &ri);
                            SHl[cc * PR_SROWS * PR_LD + pr_at(row, f)] = pm_pack(rr, ri);
                            ph += inc;
                        }
                    }
                }
                phi[0] = pm_wrap(ph);                                           // NaN stays NaN
                incq[0] = incp;
                rbq[0] = rBp;
            }
            __syncthreads();
            // ---- (2) the products, chunk by chunk: per chunk the step factors of every wavenumber go through the wave's tile
            // into registers, then block after block the state rows (8 per lane: rotate, split into float16 hi / lo) and
            // 6 MFMAs per wavenumber
            if (my_run >= 0) {                                                  // uniform over the wave
#pragma unroll 1
                for (int c2 = 0; c2 < 8; ++c2) {
                    if (!alive[c2]) continue;                                   // uniform over the workgroup
                    nmfma += 6 * NMEM * my_nblk;
                    const int idx = (p * 8 + c2) * 32 + gf;
                    const float2 S = TS[idx], rA = TA[idx], rB = TB[idx];
                    const float2 rA2 = pr_cmul(rA, rA), rB2 = pr_cmul(rB, rB);
                    float2 cur = hh ? pr_cmul(S, rA) : S;                      // this lane's rows: hh, hh + 2, ...
                    // ... and steps b = hh, hh + 2, ...: G = sigma F e^{i (b + 1) phi}.  Against (A_re, A_im) the column (b, re)
                    // holds (G_re, -G_im), the column (b, im) holds (G_im, G_re): both written, so that a lane reads its column's form
                    const float2 cb0 = hh ? rB2 : rB;
                    uint4 bh[NMEM][2], bl[NMEM][2];                              // K-steps of 16 frequencies: this lane's 4 of each
#pragma unroll
                    for (int mm = 0; mm < NMEM; ++mm) {
                        float2 cb = pr_cmul(TF[(mm * 8 + c2) * 32 + gf], cb0);
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int jb = 0; jb < PR_TT / 2; ++jb) {
                            const int b = 2 * jb + hh;
                            float rc, rs;
                            const unsigned h0 = pm_split(cb.x, cb.y, &rc, &rs), l0 = pm_pack(rc, rs);
                            Wh[pr_at(2 * b, gf)] = pm_conj(h0);
                            Wh[pr_at(2 * b + 1, gf)] = pm_swap(h0);
                            Wl[pr_at(2 * b, gf)] = pm_conj(l0);
                            Wl[pr_at(2 * b + 1, gf)] = pm_swap(l0);
                            cb = pr_cmul(cb, rB2);
                        }
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            bh[mm][ks] = *reinterpret_cast<const uint4 *>(Wh + pr_slot(orow, 4 * ks + kk));
                            bl[mm][ks] = *reinterpret_cast<const uint4 *>(Wl + pr_slot(orow, 4 * ks + kk));
                        }
                    }
#pragma unroll
                    for (int blk = 0; blk < PR_NBLK; ++blk) {
                        if (blk >= my_nblk) break;                              // uniform
                        __builtin_amdgcn_wave_barrier();                        // (the tile's last readers are done: in-order LDS)
#pragma unroll
                        for (int jr = 0; jr < PR_ROWS / 2; ++jr) {
                            // (leaving out the rows past a run's last tile -- uniform breaks in the unrolled loop -- was slower:
                            // 45 spilled SGPRs, 183 VGPRs, 23.0 -> 24.0 ms at 41 rows)
                            const int row = 2 * jr + hh;
                            float rr, ri;
                            Wh[pr_at(row, gf)] = pm_split(cur.x, cur.y, &rr, &ri);
                            Wl[pr_at(row, gf)] = pm_pack(rr, ri);
                            cur = pr_cmul(cur, rA2);
                        }
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            const pm_half8 a_hi = __builtin_bit_cast(pm_half8, *reinterpret_cast<const uint4 *>(Wh + pr_slot(orow, 4 * ks + kk)));
                            const pm_half8 a_lo = __builtin_bit_cast(pm_half8, *reinterpret_cast<const uint4 *>(Wl + pr_slot(orow, 4 * ks + kk)));
#pragma unroll
                            for (int mm = 0; mm < NMEM; ++mm) {
                                const pm_half8 b_hi = __builtin_bit_cast(pm_half8, bh[mm][ks]), b_lo = __builtin_bit_cast(pm_half8, bl[mm][ks]);
                                acc[mm][blk][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b_hi, acc[mm][blk][0], 0, 0, 0);
                                acc[mm][blk][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b_lo, acc[mm][blk][1], 0, 0, 0);
                                acc[mm][blk][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, b_hi, acc[mm][blk][0], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            // the stage's single steps: a quarter of the frequencies of every chunk per wave (their sums are added at the
            // end of the stage).  Columns 2 member + (0, 1) = the sums' real and imaginary parts for that wavenumber
            if (nshort > 0) {
#pragma unroll 1
                for (int c2 = 0; c2 < 8; ++c2) {
                    if (!alive[c2]) continue;                                   // uniform over the workgroup
                    nmfma += 3;
                    // frequencies 8 p .. 8 p + 7 of the chunk: this wave's share (a K-step holds 16: the other 8 slots are zero)
                    const int fo = c2 * PR_SROWS * PR_LD + pr_slot(orow, 2 * p + (kk & 1));
                    uint4 ah = *reinterpret_cast<const uint4 *>(SHh + fo), al = *reinterpret_cast<const uint4 *>(SHl + fo);
                    const int go = (c2 * 4 + (orow & 3)) * 32 + 4 * (2 * p + (kk & 1));
                    uint4 gh = *reinterpret_cast<const uint4 *>(FHh + go), gl = *reinterpret_cast<const uint4 *>(FHl + go);
                    if (kk >= 2) ah = al = make_uint4(0u, 0u, 0u, 0u);
                    if (kk >= 2 || orow >= 2 * NMEM) gh = gl = make_uint4(0u, 0u, 0u, 0u);
                    accs = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pm_half8, ah), __builtin_bit_cast(pm_half8, gh), accs, 0, 0, 0);
                    accs = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pm_half8, ah), __builtin_bit_cast(pm_half8, gl), accs, 0, 0, 0);
                    accs = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pm_half8, al), __builtin_bit_cast(pm_half8, gh), accs, 0, 0, 0);
                }
            }
            __syncthreads();                    // the stage's tables are re-written by the next super-chunk
            // rotate this thread's frequencies: the next super-chunk's comes to the front
            {
                const double p0 = phi[0], r0 = rw[0], w0 = wdt[0], i0 = incq[0];
                const float2 b0 = rbq[0];
                float2 f0[NMEM];
#pragma unroll
                for (int mm = 0; mm < NMEM; ++mm) f0[mm] = F[mm][0];
#pragma unroll
                for (int i = 0; i + 1 < PR_NM; ++i) {
                    phi[i] = phi[i + 1];
                    rw[i] = rw[i + 1];
                    wdt[i] = wdt[i + 1];
                    incq[i] = incq[i + 1];
#pragma unroll
                    for (int mm = 0; mm < NMEM; ++mm) F[mm][i] = F[mm][i + 1];
                    rbq[i] = rbq[i + 1];
                }
                phi[PR_NM - 1] = p0;
                rw[PR_NM - 1] = r0;
                wdt[PR_NM - 1] = w0;
                incq[PR_NM - 1] = i0;
#pragma unroll
                for (int mm = 0; mm < NMEM; ++mm) F[mm][PR_NM - 1] = f0[mm];
                rbq[PR_NM - 1] = b0;
            }
        }
        // ---- the stage's sums over this part's frequencies.  Accumulator register e of lane l: row 4 (l >> 4) + e, column l & 15
        if (my_run >= 0) {
            const PrRun *R = Q.runs + my_run;
            const int start = R->start, len = R->len;
#pragma unroll
            for (int mm = 0; mm < NMEM; ++mm) {
                if (!has[mm]) continue;
#pragma unroll
                for (int blk = 0; blk < PR_NBLK; ++blk) {
                    if (blk >= my_nblk) break;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int rel = PR_TT * (PR_ROWS * blk + 4 * kk + e) + ob;
                        if (rel < len) reinterpret_cast<float *>(out[mm] + start + rel)[onc] = (acc[mm][blk][0][e] + acc[mm][blk][1][e]) * scale[mm];
                    }
                }
            }
        }
        if (nshort > 0) {
            // the four waves' shares of the single-step sums, added in a fixed order through LDS (the waves' state tiles are
            // free: every wave is past its last block)
            float *red = reinterpret_cast<float *>(SHl + 8 * PR_SROWS * PR_LD);     // = the first wave's tile: 4 x 64 x 4 floats
            __syncthreads();
            *reinterpret_cast<pr_float4 *>(red + (p * 64 + lane) * 4) = accs;
            __syncthreads();
            if (p == 0 && orow < 2 * NMEM) {
                pr_float4 t = *reinterpret_cast<const pr_float4 *>(red + lane * 4);
#pragma unroll
                for (int q = 1; q < PR_LONGS; ++q) {
                    const pr_float4 u = *reinterpret_cast<const pr_float4 *>(red + (q * 64 + lane) * 4);
                    t = t + u;
                }
                const int mm = NMEM == 2 ? orow >> 1 : 0;
                const bool hm = NMEM == 2 ? (mm ? has[NMEM - 1] : true) : true;
                Cp<float> *om_ = NMEM == 2 ? (mm ? out[NMEM - 1] : out[0]) : out[0];
                const float sc = NMEM == 2 ? (mm ? scale[NMEM - 1] : scale[0]) : scale[0];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = 4 * kk + e;
                    if (row < nshort && hm) reinterpret_cast<float *>(om_ + st->short_tau[row])[orow & 1] = t[e] * sc;
                }
            }
            __syncthreads();                    // (the tile is written again by the next stage's first block)
        }
    }
    if (lane == 0 && Q.mfma_count) atomicAdd(Q.mfma_count, (unsigned long long)nmfma);
}
