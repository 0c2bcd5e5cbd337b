// can Stolt's own passes ride inside rocFFT as load/store callbacks?  (standalone probe, GPU box)
//   hipcc -fgpu-rdc --offload-arch=gfx950 -O3 profiles/tools/stolt_cb_probe.hip -o gpurun_out/stolt_cb_probe -lrocfft
//   gpurun_out/stolt_cb_probe [snum tnum]
// Forward half of Stolt at snum x tnum float32 (mig_python.py:152-159): taper, then rfft2 over (time, traces).
//   A  the product's way: stolt_taper_transpose-like pass ((snum, tnum) -> (tnum, snum), taper) + 2-D real plan on it
//   B  the same plan with a LOAD CALLBACK that multiplies by a weight at the plan's own offset (what a callback costs)
//   C  the same plan with a load callback that reads the caller's (snum, tnum) array transposed and tapers: pass A gone
//   D  a plan whose input strides ARE the caller's layout (no callback, no pass): what rocFFT picks for it
// Prints ms per variant (HIP events, 200 repetitions after 30 warm-ups) and the largest difference from A's spectrum.
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                   \
        }                                                                              \
    } while (0)
#define FK(x)                                                        \
    do {                                                             \
        rocfft_status s_ = (x);                                      \
        if (s_ != rocfft_status_success) {                           \
            printf("rocFFT status %d at %s:%d: %s\n", (int)s_, __FILE__, __LINE__, #x); \
            exit(3);                                                 \
        }                                                            \
    } while (0)

struct CbData {
    const float *src;       // (snum, tnum) as the caller holds it
    int snum, tnum;
    float htaper, vtaper;
};

__device__ __forceinline__ float taper_w(int i, int n, float taper)
{
    const int m = i < n - 1 - i ? i : n - 1 - i;
    const float w = (float)m / taper;
    return w > 1.f ? 1.f : w;
}

__global__ __launch_bounds__(256) void taper_transpose(const float *__restrict__ in, float *__restrict__ X, int snum, int tnum,
                                                        float htaper, float vtaper)
{
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int j0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    for (int r = ty; r < 64; r += 4) {
        const int k = k0 + r, j = j0 + tx;
        if (k < snum && j < tnum) tile[r][tx] = in[(size_t)k * tnum + j] * taper_w(j, tnum, htaper) * taper_w(k, snum, vtaper);
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int j = j0 + r, k = k0 + tx;
        if (k < snum && j < tnum) X[(size_t)j * snum + k] = tile[tx][r];
    }
}

// B: weight at the plan's own offset (X layout: offset = j * snum + k)
__device__ float load_weight(float *data, size_t offset, void *cbdata, void *)
{
    const CbData *c = (const CbData *)cbdata;
    const unsigned o = (unsigned)offset, j = o / (unsigned)c->snum, k = o - j * (unsigned)c->snum;
    return data[offset] * taper_w((int)j, c->tnum, c->htaper) * taper_w((int)k, c->snum, c->vtaper);
}
// C: the caller's array read transposed
__device__ float load_transposed(float *, size_t offset, void *cbdata, void *)
{
    const CbData *c = (const CbData *)cbdata;
    const unsigned o = (unsigned)offset, j = o / (unsigned)c->snum, k = o - j * (unsigned)c->snum;
    return c->src[(size_t)k * c->tnum + j] * taper_w((int)j, c->tnum, c->htaper) * taper_w((int)k, c->snum, c->vtaper);
}
__device__ auto load_weight_ptr = load_weight;
__device__ auto load_transposed_ptr = load_transposed;

struct Plan {
    rocfft_plan plan = nullptr;
    rocfft_execution_info info = nullptr;
    void *work = nullptr;
};

static Plan make_plan(size_t len0, size_t len1, const size_t is[2], size_t idist, const size_t os[2], size_t odist, hipStream_t st)
{
    Plan p;
    rocfft_plan_description desc = nullptr;
    FK(rocfft_plan_description_create(&desc));
    FK(rocfft_plan_description_set_data_layout(desc, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, nullptr, nullptr, 2,
                                               is, idist, 2, os, odist));
    size_t lengths[2] = {len0, len1};
    FK(rocfft_plan_create(&p.plan, rocfft_placement_notinplace, rocfft_transform_type_real_forward, rocfft_precision_single, 2, lengths, 1,
                          desc));
    rocfft_plan_description_destroy(desc);
    FK(rocfft_execution_info_create(&p.info));
    FK(rocfft_execution_info_set_stream(p.info, st));
    size_t wb = 0;
    FK(rocfft_plan_get_work_buffer_size(p.plan, &wb));
    if (wb) {
        CK(hipMalloc(&p.work, wb));
        FK(rocfft_execution_info_set_work_buffer(p.info, p.work, wb));
    }
    return p;
}

template <typename F> static float time_ms(F f, hipStream_t st)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 30; ++i) f();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < 200; ++i) f();
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / 200;
}

static double maxdiff(const std::vector<float> &a, const std::vector<float> &b)
{
    double d = 0, n = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        d = fmax(d, fabs((double)a[i] - b[i]));
        n = fmax(n, fabs((double)a[i]));
    }
    return d / n;
}

int main(int argc, char **argv)
{
    const int snum = argc > 2 ? atoi(argv[1]) : 4096, tnum = argc > 2 ? atoi(argv[2]) : 4096, m = snum / 2 + 1;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    FK(rocfft_setup());
    std::vector<float> h((size_t)snum * tnum);
    srand(1);
    for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    float *d_in, *d_X, *d_F;
    CK(hipMalloc(&d_in, h.size() * 4));
    CK(hipMalloc(&d_X, h.size() * 4));
    CK(hipMalloc(&d_F, (size_t)tnum * m * 8));
    CK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CbData cb{d_in, snum, tnum, 10.f, 20.f}, *d_cb;
    CK(hipMalloc(&d_cb, sizeof cb));
    CK(hipMemcpy(d_cb, &cb, sizeof cb, hipMemcpyHostToDevice));
    void *fw = nullptr, *ft = nullptr;
    CK(hipMemcpyFromSymbol(&fw, HIP_SYMBOL(load_weight_ptr), sizeof fw));
    CK(hipMemcpyFromSymbol(&ft, HIP_SYMBOL(load_transposed_ptr), sizeof ft));

    const size_t is[2] = {1, (size_t)snum}, os[2] = {1, (size_t)m};
    Plan A = make_plan(snum, tnum, is, (size_t)snum * tnum, os, (size_t)m * tnum, st);
    Plan B = make_plan(snum, tnum, is, (size_t)snum * tnum, os, (size_t)m * tnum, st);
    Plan C = make_plan(snum, tnum, is, (size_t)snum * tnum, os, (size_t)m * tnum, st);
    const size_t isD[2] = {(size_t)tnum, 1};
    Plan D = make_plan(snum, tnum, isD, (size_t)snum * tnum, os, (size_t)m * tnum, st);
    void *cbd[1] = {d_cb};
    void *fwa[1] = {fw}, *fta[1] = {ft};
    FK(rocfft_execution_info_set_load_callback(B.info, fwa, cbd, 0));
    FK(rocfft_execution_info_set_load_callback(C.info, fta, cbd, 0));

    dim3 tgrid((tnum + 63) / 64, (snum + 63) / 64);
    void *ib[1], *ob[1] = {d_F};
    std::vector<float> ra((size_t)tnum * m * 2), rb(ra.size());
    auto runA = [&] {
        hipLaunchKernelGGL(taper_transpose, tgrid, dim3(256), 0, st, d_in, d_X, snum, tnum, 10.f, 20.f);
        ib[0] = d_X;
        FK(rocfft_execute(A.plan, ib, ob, A.info));
    };
    const float msA = time_ms(runA, st);
    CK(hipMemcpy(ra.data(), d_F, ra.size() * 4, hipMemcpyDeviceToHost));
    printf("{\"snum\": %d, \"tnum\": %d, \"A_pass_plus_plan_ms\": %.4f", snum, tnum, msA);
    fflush(stdout);

    // the plan alone (X already tapered)
    auto runA0 = [&] {
        ib[0] = d_X;
        FK(rocfft_execute(A.plan, ib, ob, A.info));
    };
    printf(", \"A_plan_alone_ms\": %.4f", time_ms(runA0, st));
    fflush(stdout);

    // B: X untapered copy first
    {
        hipLaunchKernelGGL(taper_transpose, tgrid, dim3(256), 0, st, d_in, d_X, snum, tnum, 1e-30f, 1e-30f);   // weights clip to 1
        CK(hipStreamSynchronize(st));
    }
    auto runB = [&] {
        ib[0] = d_X;
        FK(rocfft_execute(B.plan, ib, ob, B.info));
    };
    const float msB = time_ms(runB, st);
    CK(hipMemcpy(rb.data(), d_F, rb.size() * 4, hipMemcpyDeviceToHost));
    printf(", \"B_weight_callback_ms\": %.4f, \"B_diff\": %.3g", msB, maxdiff(ra, rb));
    fflush(stdout);

    auto runC = [&] {
        ib[0] = d_X;        // (ignored by the callback)
        FK(rocfft_execute(C.plan, ib, ob, C.info));
    };
    const float msC = time_ms(runC, st);
    CK(hipMemcpy(rb.data(), d_F, rb.size() * 4, hipMemcpyDeviceToHost));
    printf(", \"C_transposing_callback_ms\": %.4f, \"C_diff\": %.3g", msC, maxdiff(ra, rb));
    fflush(stdout);

    auto runD = [&] {
        ib[0] = d_in;
        FK(rocfft_execute(D.plan, ib, ob, D.info));
    };
    const float msD = time_ms(runD, st);
    printf(", \"D_strided_plan_no_taper_ms\": %.4f", msD);
    auto runP = [&] { hipLaunchKernelGGL(taper_transpose, tgrid, dim3(256), 0, st, d_in, d_X, snum, tnum, 10.f, 20.f); };
    printf(", \"pass_alone_ms\": %.4f", time_ms(runP, st));
    printf(", \"A_again_ms\": %.4f}\n", time_ms(runA, st));
    return 0;
}
