// The Kirchhoff plan object shared by kirchhoff.hip (plan creation, prep, the ring kernels and their launches)
// and kirch_gen.hip (the non-uniform-spacing kernel).
#pragma once
#include "common.h"
#include <vector>

#define KF_THREADS 256
#define KF_W 512            // LDS floats per ring slot (circular window)
#define KF_PAD_ROWS 128      // zero traces kept on both sides of the image (loops are clipped to the profile)

struct impdar_kirch_plan {
    impdar_ctx *ctx = nullptr;
    int dtype = IMPDAR_F32, snum = 0, tnum = 0, tnum_pad = 0, nranks = 1;
    int nearfield = 0, mode = IMPDAR_KIRCH_EXACT;
    int grad_uniform = 0;
    double grad_h = 1.0, vel = 0, tmax = 0, dt = 1, dx = 1, tt0 = 0, alpha = 1;
    bool uniform = false;
    bool dist_sorted = false;
    // exact path on uniform grids: fp64 pick / weight tables (built at the first migrate)
    DevBuf d_XK, d_XW, d_XW2, d_xhmax;
    int xntab = 0;
    bool xtab_ready = false, xtab_off = false;
    int diag_tables_built = 0;
    bool table_built[2] = {false, false};   // ring kernels: the geometry-only pick table of buffer set b exists
    // device tables
    DevBuf d_dist, d_tt, d_zs, d_zs2, d_ga, d_gb, d_gc;
    // Everything a prep produces is double-buffered: prep / table / all-gather of radargram
    // s+1 run on the context's aux stream while the diffraction sum of radargram s runs on
    // the compute stream.  `buf` flips at the first prep after a migrate.
    DevBuf GT[2], DT[2];           // images with KF_PAD_ROWS all-zero rows before row 0 and after row tnum_pad-1
    int buf = 0;
    bool migrated_since_prep = false;
    const void *last_out = nullptr;   // output of the last migrate (a prep that reads it must wait for that kernel)
    int last_out_buf = 0;
    hipEvent_t ev_ready[2] = {nullptr, nullptr};   // image + table of buffer b complete (aux stream)
    hipEvent_t ev_free[2] = {nullptr, nullptr};    // last migrate reading buffer b done (compute stream)
    bool free_recorded[2] = {false, false};
    DevBuf d_hmax, d_klo, d_khi;
    DevBuf d_TK[2], d_TW[2], d_TW2[2], d_c1, d_c2, d_fin, d_WIN;
    int nrows = 0, mrow0 = 0;   // quad kernel: step-block table rows (see FastParams)
    int quadSH = 0;             // table entries are LDS byte offsets >> quadSH
    int nb = 0, ntab = 0;
    bool quad = false;          // sample-major LDS ring (kirch_quad_kernel)
    std::vector<int> h_hmax;    // host copy of the per-chunk aperture half widths (tile cost model)
    DevBuf d_queue;             // quad kernel, persistent workgroups: per-XCD item counters
    int slots = 0;              // ... and how many of them are resident at once (occupancy query, cached)
    int walk_parts_log2 = 0;    // every tile's aperture walk as 1, 2 or 4 queue items (plans of 4+ / 8+ ranks)
    DevBuf d_partial;           // ... and the partial images of the pieces
    DevBuf d_tilemap;           // ring kernels: (chunk, slot, XCD) -> output tile, balanced over the XCDs
    std::vector<short> h_tilemap;
    int tm_key[5] = {-1, -1, -1, -1, -1};   // (xlo, xhi, tile width, G, tiles_per_xcd) the cached map was built for
    int nh = 1;                 // quad kernel: output tiles per workgroup sharing one ring (256 nh threads)
    int lk = 0;                 // ... and extra ring groups = blocks of additional staging lookahead (nh >= 2 only)
    bool dquad = false;         // the same ring in float64 (kirch_dquad_kernel): exact mode, float64 data, uniform grids
    double xnoise = 0;          // position noise of dist[j] - dist[xi] in units of dx (see plan creation)
    // float32 data on a non-uniform (sorted) dist[]: kirch_gen_kernel (kirch_gen.h)
    bool gen = false;
    int genW = 0;               // samples per LDS slot
    DevBuf d_ga32, d_ga2_32, d_alo2, d_jr;
    std::vector<double> h_dist, h_zs2min;   // host copies for the per-(chunk, tile) input trace ranges
    std::vector<int2> h_jr;
    int jr_key[2] = {-1, -1};   // (xlo, xhi) the cached ranges were built for
    bool tie_ambiguous = false; // more rounding-noise ties than the list holds: per-pair kernel only
    int ntie_groups = 0;        // samples with flagged offsets (kirch_tiefix_kernel after every table-driven diffraction sum)
    DevBuf d_tie_ti, d_tie_off, d_tie_n;
    DevBuf d_c1d, d_c2d, d_find;
    int quadW = 0;              // samples per ring slot in that layout
    // host copies for pair counting
    std::vector<int> h_half;       // exact aperture half-width per sample (uniform grids)
    int nchunks = 0;
    // ring of HIP-event sets so a timed loop can read per-step kernel
    // durations afterwards without synchronising inside the loop
    static constexpr int NSLOT = 64;
    hipEvent_t evs[NSLOT][6] = {};
    bool haves[NSLOT][3] = {};
    int slot = 0;
    int xb = 24;                   // fast-kernel trace tile (24: quad ring, 16: tab ring)

    ~impdar_kirch_plan()
    {
        for (hipEvent_t e : {ev_ready[0], ev_ready[1], ev_free[0], ev_free[1]})
            if (e) (void)hipEventDestroy(e);
        for (int s = 0; s < NSLOT; ++s)
            for (int i = 0; i < 6; ++i)
                if (evs[s][i]) (void)hipEventDestroy(evs[s][i]);
    }
};

static inline char *img_row0(const impdar_kirch_plan *p, const DevBuf &b)  // b = GT[buf] / DT[buf]
{
    return reinterpret_cast<char *>(b.p) + (size_t)KF_PAD_ROWS * p->snum * impdar_dtype_size(p->dtype);
}

static int upload(DevBuf &b, const void *src, size_t bytes)
{
    IMPDAR_HIP_CHECK(b.ensure(bytes ? bytes : 8));
    if (bytes) IMPDAR_HIP_CHECK(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return IMPDAR_OK;
}

// kirch_gen.hip: kirch_gen_kernel + kirch_gen_shell_kernel on output traces [xlo, xhi) of the prepared image
int kirch_launch_gen(impdar_kirch_plan *p, void *d_out, int xlo, int xhi, hipStream_t st);
