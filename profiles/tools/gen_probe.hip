// Where does a pair step of kirch_gen_kernel (impdar_amd/csrc/kirch_gen.hip) spend its time?  The step's instruction
// sequence in isolation, pieces added one by one; ns and cycles per wave pair step per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/gen_probe profiles/tools/gen_probe.hip && /tmp/gen_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define MAGIC 12582912.0f

// V: 0 index chain + sum only (no LDS, no branch)   1 + the ds_read_b32 gather   2 + a uniform branch per pair (never taken)
//    3 + D^2 quads read from LDS (broadcast)         4 = 3 with four pairs per branch   5 = 3 without the branch
//    6 = 3 with the normalised-position form (v_fma ... clamp instead of v_min)
template <int V, int XB>
__global__ __launch_bounds__(256, 4) void probe(float *out, const float *d2g, int iters, float thr, float a2in)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef const __attribute__((address_space(3))) float *lds_fp;
    typedef const __attribute__((address_space(3))) f4 *lds_f4p;
    const int tid = threadIdx.x;
    for (int e = tid; e < 9216; e += 256) lds[e] = 1.0f + e * 1e-6f;
    __syncthreads();
    float a2 = a2in + tid * 2.0f * 30.f + 900.f;            // (30 + tid)^2-ish: positions of consecutive lanes one sample apart
    const float nu0 = -0.0f, clampf = 511.f;
    const unsigned bias = 0u - (0x4B400000u << 2);
    float acc[XB];
#pragma unroll
    for (int i = 0; i < XB; ++i) acc[i] = 0.f;
    f4 dreg[XB / 4];
#pragma unroll
    for (int i = 0; i < XB / 4; ++i) dreg[i] = *reinterpret_cast<const f4 *>(d2g + 4 * i);
    const unsigned d2row = 8192u * 4u;
    if (V >= 3)
        for (int e = tid; e < XB; e += 256) lds[8192 + e] = d2g[e];
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        float g_prev = 0.f, r_prev = 0.f;
        a2 += 1.0e-3f;                  // (nothing of the index chain is loop invariant)
#pragma unroll
        for (int iq = 0; iq < XB / 4; ++iq) {
            f4 d4 = dreg[iq];
            if (V >= 3) d4 = *(lds_f4p)(uintptr_t)(d2row + iq * 16u);
            if (V == 4) {
                float rq[4], dfq[4];
                unsigned adq[4];
                bool fl[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float q = a2 + d4[c];
                    rq[c] = __builtin_amdgcn_rsqf(q);
                    const float s2c = fminf(fmaf(q, rq[c], nu0), clampf);
                    const float f = s2c + MAGIC;
                    const float kf = f - MAGIC;
                    dfq[c] = s2c - kf;
                    adq[c] = (__float_as_uint(f) << 2) + bias;
                    fl[c] = fabsf(dfq[c]) > thr;
                }
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(fl[0] || fl[1] || fl[2] || fl[3]) != 0, 0)) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (fl[c]) adq[c] = (unsigned)(dfq[c] * 3.f) & 1020u;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int i = iq * 4 + c;
                    const float g = *(lds_fp)(uintptr_t)adq[c];
                    if (i > 0) acc[i > 0 ? i - 1 : 0] = fmaf(r_prev, g_prev, acc[i > 0 ? i - 1 : 0]);
                    g_prev = g;
                    r_prev = rq[c];
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int i = iq * 4 + c;
                    const float q = a2 + d4[c];
                    const float r = __builtin_amdgcn_rsqf(q);
                    float f, df;
                    if (V == 6) {
                        float p;        // normalised position in the slot, saturated to [0, 1] by the clamp modifier
                        asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(p) : "v"(q), "v"(r), "v"(nu0));
                        f = fmaf(p, clampf, MAGIC);
                        const float kf = f - MAGIC;
                        df = fmaf(p, clampf, -kf);
                    } else {
                        const float s2c = fminf(fmaf(q, r, nu0), clampf);
                        f = s2c + MAGIC;
                        const float kf = f - MAGIC;
                        df = s2c - kf;
                    }
                    unsigned addr = (__float_as_uint(f) << 2) + bias;
                    if (V >= 2 && V != 5) {
                        const bool flag = fabsf(df) > thr;
                        if (__builtin_expect(__builtin_amdgcn_ballot_w64(flag) != 0, 0))
                            if (flag) addr = (unsigned)(df * 3.f) & 1020u;
                    } else {
                        asm volatile("" ::"v"(df));
                    }
                    float g;
                    if (V >= 1) g = *(lds_fp)(uintptr_t)addr;
                    else g = __uint_as_float(addr);
                    if (i > 0) acc[i > 0 ? i - 1 : 0] = fmaf(r_prev, g_prev, acc[i > 0 ? i - 1 : 0]);
                    g_prev = g;
                    r_prev = r;
                }
            }
        }
        acc[XB - 1] = fmaf(r_prev, g_prev, acc[XB - 1]);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < XB; ++i) s += acc[i];
    out[blockIdx.x * 256 + tid] = s;
}

template <int V, int XB> static void run(const char *name, float *d_out, float *d_d2, int wgs_per_cu)
{
    const int iters = 4000;
    const size_t shmem = wgs_per_cu >= 4 ? 36864 : (wgs_per_cu == 2 ? 73728 : 147456);     // forces the occupancy
    auto k = probe<V, XB>;
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), shmem, 0, d_out, d_d2, 10, 10.f, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), shmem, 0, d_out, d_d2, iters, 10.f, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // wave pair steps per SIMD: grid x 4 waves x iters x XB / 1024 SIMDs
    const double steps = (double)grid * 4 * iters * XB / 1024.0;
    printf("%-52s %d waves/SIMD: %.3f ms, %.2f ns per wave pair step per SIMD\n", name, wgs_per_cu, ms, ms * 1e6 / steps);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main()
{
    float *d_out, *d_d2;
    hipMalloc(&d_out, 256 * 8 * 256 * 4);
    hipMalloc(&d_d2, 64 * 4);
    float h[64];
    for (int i = 0; i < 64; ++i) h[i] = 100.f + 37.f * i;
    hipMemcpy(d_d2, h, sizeof h, hipMemcpyHostToDevice);
    for (int w : {4, 2, 1}) {
        run<0, 32>("0 index chain + sum (10 VALU)", d_out, d_d2, w);
        run<1, 32>("1 + ds_read_b32 gather", d_out, d_d2, w);
        run<5, 32>("5 + D^2 quads from LDS, no branch", d_out, d_d2, w);
        run<2, 32>("2 = 1 + branch per pair", d_out, d_d2, w);
        run<3, 32>("3 = 2 + D^2 quads from LDS (the kernel's step)", d_out, d_d2, w);
        run<4, 32>("4 = 3 with one branch per four pairs", d_out, d_d2, w);
        run<6, 32>("6 = 3 with fma-clamp instead of v_min", d_out, d_d2, w);
    }
    return 0;
}
