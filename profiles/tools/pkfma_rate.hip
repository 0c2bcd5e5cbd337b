// Micro-benchmark: issue rate of v_pk_fma_f32 vs v_fma_f32 on gfx950 (one workgroup of 256 threads per CU x 8 waves/SIMD).
//   hipcc --offload-arch=gfx950 -O3 profiles/tools/pkfma_rate.hip -o /tmp/pkfma_rate && /tmp/pkfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float w0)
{
    f2 acc[16];
    f2 d[8];
    f2 w = {w0 + threadIdx.x, w0};
    for (int i = 0; i < 16; ++i) acc[i] = f2{(float)i, (float)threadIdx.x};
    for (int i = 0; i < 8; ++i) d[i] = f2{1.0f + i, 0.5f * threadIdx.x};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(w), "v"(d[i & 7]));
                } else if (MODE == 1) {
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i].x) : "v"(w.x), "v"(d[i & 7].x));
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i].y) : "v"(w.x), "v"(d[i & 7].y));
                } else {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[i]) : "v"(w), "v"(d[i & 7]));
                }
            }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    float *o;
    hipMalloc(&o, 256 * 2048 * 4);
    const int iters = 20000;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    // one 256-thread workgroup = one wave per SIMD of a CU; wps workgroups per CU resident at once
    for (int wps : {1, 2, 3, 4, 8})
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            const int grid = 256 * wps;
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, o, iters, 1.0f);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, o, iters, 1.0f);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, o, iters, 1.0f);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            double fma = (double)grid * 256 * iters * 64 * 2;
            if (rep) printf("waves/SIMD %d mode %d (%s): %.3f ms, %.1f TFLOP/s\n", wps, mode, mode == 1 ? "2x v_fmac_f32" : "v_pk_fma_f32", ms, 2 * fma / ms / 1e9);
        }
    }
    return 0;
}
