// VALU issue-rate probe for gfx950: cycles per wave64 instruction and per SIMD for v_fma_f32 / v_pk_fma_f32 / v_add_f32
// chains at 1, 2, 4 and 8 waves per SIMD, with 1..8 independent chains per wave.
//   hipcc --offload-arch=gfx950 -O3 -o build/diag/valu_probe profiles/tools/valu_probe.hip && build/diag/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND, int CH>
__global__ void probe(float *out, int iters, long long *cyc)
{
    float a[8];
    v2f b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = threadIdx.x * 1e-9f + i;
        b[i] = v2f{a[i], a[i] + 1.f};
    }
    const float m = 0.999f, c = 1e-3f;
    const v2f pm = {m, m}, pc = {c, c};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(b[i]) : "v"(pm), "v"(pc));
                if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 4) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + b[i].x + b[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND, int CH> static void run(const char *name, float *d, long long *dc)
{
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {
        const int block = 256 * wps;        // 4 SIMDs x wps waves, one workgroup per CU
        if (block > 1024) {
            // two workgroups of 1024 cannot be forced onto one CU; use 2 x 1024 via grid = 512 and hope for co-residency
        }
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        const int grid = block > 1024 ? 512 : 256;
        const int blk = block > 1024 ? 1024 : block;
        probe<KIND, CH><<<grid, blk>>>(d, 10, dc);
        hipEventRecord(e0);
        probe<KIND, CH><<<grid, blk>>>(d, iters, dc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        long long cyc;
        hipMemcpy(&cyc, dc, sizeof cyc, hipMemcpyDeviceToHost);
        const double instr_per_wave = (double)iters * 16 * CH;
        const double waves_per_simd = wps;
        // s_memtime/readcyclecounter ticks at a fixed 100 MHz on gfx9: use the event time and an assumed clock instead
        const double ns_per_instr_simd = ms * 1e6 / (instr_per_wave * waves_per_simd);
        printf("%-12s chains %d waves/SIMD %d: %.3f ms, %.3f ns per wave-instr per SIMD (= %.2f cycles at 2.4 GHz), counter %lld\n", name, CH,
               wps, ms, ns_per_instr_simd, ns_per_instr_simd * 2.4, cyc);
    }
}

int main()
{
    float *d;
    long long *dc;
    hipMalloc(&d, 1 << 24);
    hipMalloc(&dc, 8);
    run<0, 1>("fma", d, dc);
    run<0, 2>("fma", d, dc);
    run<0, 4>("fma", d, dc);
    run<0, 8>("fma", d, dc);
    run<4, 1>("fmac(vop2)", d, dc);
    run<4, 8>("fmac(vop2)", d, dc);
    run<2, 1>("add", d, dc);
    run<2, 8>("add", d, dc);
    run<3, 8>("mul", d, dc);
    run<1, 1>("pk_fma", d, dc);
    run<1, 2>("pk_fma", d, dc);
    run<1, 8>("pk_fma", d, dc);
    return 0;
}
