// Does vector work hide under MFMAs on gfx950?  One workgroup per CU, cycles per iteration of
//   M   : 12 dependent v_mfma_f32_32x32x16_f16 (one accumulator, as a row block of ps_mfma_kernel)
//   V   : 96 vector instructions of the step-factor generation's kind (v_fma_f32 chain, v_cvt_pkrtz_f16_f32)
//   MV  : the same 12 MFMAs with 8 of those vector instructions behind each (one wave, interleaved)
//   M|V : waves 0..3 run M, waves 4..7 run V (two waves per SIMD, one of each kind)
// for 4 waves (one per SIMD) and 8 waves (two per SIMD).
//   hipcc --offload-arch=gfx950 -O3 profiles/tools/mfma_valu_probe.hip -o build/probe/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

template <int MODE>   // 0 M, 1 V, 2 MV, 3 M|V
__global__ __launch_bounds__(512) void probe(float *out, int iters, long long *cyc)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    half8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(lane * 0.001f + i);
        b[i] = (_Float16)(lane * 0.002f - i);
    }
    float16v acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float x = lane * 1e-3f, y = 1.f - x, c = 0.999f, s = 0.04f;
    unsigned pk = 0;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            if (do_m) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            if (do_v) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    // a rotation (4 ops) + two pack conversions, twice: 8 + 4 instructions... kept to 8 per MFMA
                    float nx, ny;
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(nx) : "v"(y), "v"(s));
                    asm volatile("v_fma_f32 %0, %1, %2, -%0" : "+v"(nx) : "v"(x), "v"(c));
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ny) : "v"(y), "v"(c));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ny) : "v"(x), "v"(s));
                    x = nx;
                    y = ny;
                }
            }
        }
        if (do_v) pk ^= __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x, y));
    }
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
    float r = x + y + (float)pk;
    for (int i = 0; i < 16; ++i) r += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main()
{
    float *out;
    long long *cyc, h[8];
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 64);
    const int iters = 2000;
    const char *names[] = {"M   (12 dependent MFMAs)", "V   (96 vector ops)", "MV  (interleaved, one wave)", "M|V (waves 0-3 MFMA, 4-7 vector)"};
    for (int threads : {256, 512})
        for (int mode = 0; mode < 4; ++mode) {
            if (mode == 3 && threads == 256) continue;
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(cyc, 0, 64);
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
                if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
                if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
                if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
                hipDeviceSynchronize();
                hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
            }
            printf("%d waves per CU, %-36s cycles per iteration: wave 0 %.0f", threads / 64, names[mode], (double)h[0] / iters);
            if (threads == 512) printf(", wave 4 %.0f", (double)h[4] / iters);
            printf("\n");
        }
    return 0;
}
