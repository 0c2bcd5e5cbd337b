// Do hipStreamWaitValue32 / hipStreamWriteValue32 work on this stack, and on which memory?  (one-shot pipeline, round 4)
//   hipcc --offload-arch=gfx950 -O3 profiles/tools/waitvalue_probe.hip -o /tmp/waitvalue_probe && /tmp/waitvalue_probe
// Stream A runs a kernel that bumps a counter to 5 after ~2 ms; stream B waits for counter >= 5, then records an event.
// Stream C writes a value that a spinning kernel on stream A waits for.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s (%d) line %d\n", hipGetErrorString(e_), (int)e_, __LINE__); return 2; } } while (0)

__global__ void bump_later(unsigned *ctr, long long cycles)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(32);
    __threadfence_system();
    atomicAdd(ctr, 5u);
}
__global__ void wait_for(volatile unsigned *flag, unsigned want, unsigned *result, long long limit)
{
    const long long t0 = wall_clock64();
    unsigned v;
    while ((v = __atomic_load_n((unsigned *)flag, __ATOMIC_RELAXED)) < want && wall_clock64() - t0 < limit) __builtin_amdgcn_s_sleep(8);
    *result = v;
}

static int run(const char *name, unsigned *ctr)
{
    hipStream_t A, B, C;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&C, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipMemset(ctr, 0, 64));
    CK(hipDeviceSynchronize());
    // wall_clock64 ticks at 100 MHz: 2 ms = 200000
    CK(hipEventRecord(e0, A));
    hipLaunchKernelGGL(bump_later, dim3(1), dim3(1), 0, A, ctr, 200000LL);
    hipError_t w = hipStreamWaitValue32(B, ctr, 5, hipStreamWaitValueGte, 0xffffffffu);
    if (w != hipSuccess) {
        printf("%s: hipStreamWaitValue32 -> %s\n", name, hipGetErrorString(w));
        (void)hipGetLastError();
        CK(hipDeviceSynchronize());
    } else {
        CK(hipEventRecord(e1, B));
        const auto t0 = std::chrono::steady_clock::now();
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipDeviceSynchronize());
        hipError_t q = hipEventElapsedTime(&ms, e0, e1);
        printf("%s: wait released %.3f ms after the kernel's start (expected ~2.0)%s, host waited %.3f ms\n", name, ms,
               q == hipSuccess ? "" : " [elapsed time unavailable]",
               std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    // write value: a kernel spins on ctr[8] until stream C writes 7
    unsigned *res;
    CK(hipMalloc(&res, 4));
    hipLaunchKernelGGL(wait_for, dim3(1), dim3(1), 0, A, ctr + 8, 7u, res, 100000000LL);   // up to 1 s
    hipError_t ww = hipStreamWriteValue32(C, ctr + 8, 7, 0);
    if (ww != hipSuccess) {
        printf("%s: hipStreamWriteValue32 -> %s\n", name, hipGetErrorString(ww));
        (void)hipGetLastError();
        unsigned seven = 7;
        CK(hipMemcpyAsync(ctr + 8, &seven, 4, hipMemcpyHostToDevice, C));
    }
    CK(hipDeviceSynchronize());
    unsigned got = 0;
    CK(hipMemcpy(&got, res, 4, hipMemcpyDeviceToHost));
    printf("%s: spinning kernel saw %u (7 = the stream write arrived while it ran)\n", name, got);
    return 0;
}

int main()
{
    unsigned *dev = nullptr, *sig = nullptr, *host = nullptr;
    CK(hipMalloc(&dev, 64));
    run("hipMalloc memory", dev);
    hipError_t e = hipExtMallocWithFlags((void **)&sig, 64, hipMallocSignalMemory);
    if (e == hipSuccess) run("signal memory", sig);
    else printf("hipExtMallocWithFlags(hipMallocSignalMemory) -> %s\n", hipGetErrorString(e));
    (void)hipGetLastError();
    e = hipHostMalloc((void **)&host, 64, hipHostMallocMapped);
    if (e == hipSuccess) run("pinned host memory", host);
    return 0;
}
