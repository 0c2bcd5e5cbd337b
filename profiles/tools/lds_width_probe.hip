// LDS read cost by access width on gfx950 (would partial reads of edge quads pay?)
//   hipcc --offload-arch=gfx950 -O3 profiles/tools/lds_width_probe.hip -o /tmp/lds_width_probe && /tmp/lds_width_probe
// Every wave issues N independent reads of one width from conflict-free addresses (lane-linear), 8 waves per CU x 2
// workgroups; prints ns per wave instruction per CU and the bytes per clock per CU that follows at the measured time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

template <int W> __global__ __launch_bounds__(256) void probe(float *out, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (float)i;
    __syncthreads();
    // byte address of this lane's element: lane-linear at the access width (the 64 lanes of an instruction cover
    // 64 W consecutive bytes); W = 12: 16-byte pitch, 12 bytes read
    const unsigned pitch = W == 12 ? 16 : W;
    unsigned a = (threadIdx.x & 63) * pitch + (threadIdx.x >> 6) * 4096;
    float acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if constexpr (W == 4) {
                float v;
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(u * 256));
                asm volatile("" ::"v"(v));
            } else if constexpr (W == 8) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 v;
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(u * 512));
                asm volatile("" ::"v"(v));
            } else if constexpr (W == 12) {
                typedef float f3 __attribute__((ext_vector_type(3)));
                f3 v;
                asm volatile("ds_read_b96 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(u * 1024));
                asm volatile("" ::"v"(v));
            } else {
                typedef float f4 __attribute__((ext_vector_type(4)));
                f4 v;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(u * 1024));
                asm volatile("" ::"v"(v));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (acc == 12345.f) out[0] = acc;
}

template <int W> static void run(const char *name, float *d)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const int iters = 20000, blocks = 512;
    hipLaunchKernelGGL(probe<W>, dim3(blocks), dim3(256), 0, 0, d, 100);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(probe<W>, dim3(blocks), dim3(256), 0, 0, d, iters);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    // wave instructions per CU: 2 workgroups x 4 waves x iters x 16
    const double per_cu = 2.0 * 4 * iters * 16;
    const double ns = ms * 1e6 / per_cu;
    printf("%-14s %7.3f ms  %6.3f ns per wave instruction per CU  = %5.2f clk at 2.4 GHz, %6.1f B/clk/CU\n", name, ms, ns, ns * 2.4,
           64.0 * W / (ns * 2.4));
}

int main()
{
    float *d;
    CK(hipMalloc(&d, 4096));
    for (int rep = 0; rep < 2; ++rep) {
        run<4>("ds_read_b32", d);
        run<8>("ds_read_b64", d);
        run<12>("ds_read_b96", d);
        run<16>("ds_read_b128", d);
    }
    return 0;
}
