// Dependent-issue latency of fp64 VALU operations and quad DPP moves for ONE wavefront on a CU (what bounds
// the band-pass recurrence of csrc/preproc.hip): cycles per operation from s_memtime, and the core clock those
// cycles ran at from s_memrealtime (100 MHz).   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off dp_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE> __global__ void chain(double *out, long long *cyc, long long *rt, double a, double b, int n)
{
    double z = a + threadIdx.x, z2 = b;
    const long long r0 = wall_clock64();
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) z = z + b;                          // dependent add
            if (MODE == 1) z = z * b;                          // dependent mul
            if (MODE == 2) z = __builtin_fma(z, b, a);         // dependent fma
            if (MODE == 3) { z = z + b; z2 = z2 + a; }         // two independent chains
            if (MODE == 4) {                                   // the filter's critical path: dpp, add, mul, sub
                int lo = __builtin_amdgcn_mov_dpp(__double2loint(z), 0x00, 0xf, 0xf, true);
                int hi = __builtin_amdgcn_mov_dpp(__double2hiint(z), 0x00, 0xf, 0xf, true);
                const double y = __hiloint2double(hi, lo) + a;
                z = z2 - y * b;
            }
            if (MODE == 5) { const double y = z + a; z = z2 - y * b; }   // the same without the DPP hop
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    const long long r1 = wall_clock64();
    out[threadIdx.x] = z + z2;
    if (threadIdx.x == 0) { *cyc = t1 - t0; *rt = r1 - r0; }
}

int main()
{
    double *out; long long *cyc, *rt;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8); hipMalloc(&rt, 8);
    const int n = 20000;
    const char *names[] = {"add_f64 dependent", "mul_f64 dependent", "fma_f64 dependent", "2 independent add chains (per pair)",
                           "dpp+add+mul+sub (filter step)", "add+mul+sub (no dpp)"};
    for (int m = 0; m < 6; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (m) {
            case 0: hipLaunchKernelGGL(chain<0>, 1, 64, 0, 0, out, cyc, rt, 1.0, 1e-9, n); break;
            case 1: hipLaunchKernelGGL(chain<1>, 1, 64, 0, 0, out, cyc, rt, 1.0, 1.0000001, n); break;
            case 2: hipLaunchKernelGGL(chain<2>, 1, 64, 0, 0, out, cyc, rt, 1e-9, 0.999, n); break;
            case 3: hipLaunchKernelGGL(chain<3>, 1, 64, 0, 0, out, cyc, rt, 1.0, 1e-9, n); break;
            case 4: hipLaunchKernelGGL(chain<4>, 1, 64, 0, 0, out, cyc, rt, 1e-3, 0.5, n); break;
            case 5: hipLaunchKernelGGL(chain<5>, 1, 64, 0, 0, out, cyc, rt, 1e-3, 0.5, n); break;
            }
            hipDeviceSynchronize();
        }
        long long c, r;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&r, rt, 8, hipMemcpyDeviceToHost);
        printf("%-40s %7.2f cycles per unrolled item, core clock %.0f MHz\n", names[m], (double)c / (16.0 * n),
               (double)c / ((double)r / 100.0));
    }
    return 0;
}
