// Probe for the one-shot Kirchhoff upload: can column blocks of a pageable (snum, tnum) host array cross PCIe as
// fast as the whole array does?  Times hipMemcpyAsync of the whole array against hipMemcpy2DAsync of trace blocks.
//   hipcc -O2 profiles/tools/h2d_2d_probe.cpp -o build/probe/h2d_2d_probe && build/probe/h2d_2d_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
    const size_t snum = 4096, tnum = 10000, esz = 4, bytes = snum * tnum * esz;
    char *h = (char *)malloc(bytes), *h2 = (char *)malloc(bytes);
    memset(h, 1, bytes);
    memset(h2, 0, bytes);
    char *d;
    CK(hipMalloc(&d, bytes));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    for (int rep = 0; rep < 4; ++rep) {
        auto t0 = now();
        CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        printf("whole 1-D pageable H2D: %.2f ms (%.1f GB/s)\n", ms(t0, now()), bytes / ms(t0, now()) / 1e6);
    }
    const size_t cuts[] = {0, 4460, 8000, 10000};
    for (int rep = 0; rep < 3; ++rep)
        for (int b = 0; b < 3; ++b) {
            const size_t j0 = cuts[b], w = (cuts[b + 1] - cuts[b]) * esz;
            auto t0 = now();
            CK(hipMemcpy2DAsync(d + j0 * esz, tnum * esz, h + j0 * esz, tnum * esz, w, snum, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            printf("2-D pageable H2D traces [%zu, %zu): %.2f ms (%.1f GB/s)\n", cuts[b], cuts[b + 1], ms(t0, now()),
                   w * snum / ms(t0, now()) / 1e6);
        }
    // the same from the device to pageable memory (output blocks)
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = now();
        CK(hipMemcpyAsync(h2, d, bytes, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        printf("whole 1-D pageable D2H: %.2f ms (%.1f GB/s)\n", ms(t0, now()), bytes / ms(t0, now()) / 1e6);
    }
    // device -> pageable, trace (column) blocks
    for (int rep = 0; rep < 3; ++rep)
        for (int b = 0; b < 3; ++b) {
            const size_t j0 = cuts[b], w = (cuts[b + 1] - cuts[b]) * esz;
            auto t0 = now();
            CK(hipMemcpy2DAsync(h2 + j0 * esz, tnum * esz, d + j0 * esz, tnum * esz, w, snum, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            printf("2-D pageable D2H traces [%zu, %zu): %.2f ms (%.1f GB/s)\n", cuts[b], cuts[b + 1], ms(t0, now()),
                   w * snum / ms(t0, now()) / 1e6);
        }
    // a compact (snum x width) device block into columns of the host array (what an output block of the split call is)
    for (int rep = 0; rep < 3; ++rep) {
        const size_t j0 = 4000, wcols = 3000, w = wcols * esz;
        auto t0 = now();
        CK(hipMemcpy2DAsync(h2 + j0 * esz, tnum * esz, d, w, w, snum, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        printf("2-D pageable D2H compact block -> columns [4000, 7000): %.2f ms (%.1f GB/s)\n", ms(t0, now()), w * snum / ms(t0, now()) / 1e6);
    }
    // into FRESH pageable memory (first touch by the copy), as a result array just made by numpy.empty is
    for (int rep = 0; rep < 2; ++rep) {
        char *h3 = (char *)malloc(bytes);
        auto t0 = now();
        CK(hipMemcpyAsync(h3, d, bytes, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        printf("whole 1-D D2H into untouched pageable memory: %.2f ms (%.1f GB/s)\n", ms(t0, now()), bytes / ms(t0, now()) / 1e6);
        free(h3);
    }
    // device -> PINNED staging buffers of different sizes (hipHostMalloc), 54.6 MB pieces
    for (size_t mb : {164, 328, 656}) {
        char *pin = nullptr;
        auto ta = now();
        CK(hipHostMalloc((void **)&pin, mb << 20, hipHostMallocDefault));
        const double alloc_ms = ms(ta, now());
        for (int rep = 0; rep < 3; ++rep) {
            auto t0 = now();
            CK(hipMemcpyAsync(pin + (size_t)rep * (54u << 20), d, (size_t)54 << 20, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            printf("D2H 54 MB into a %zu MB hipHostMalloc buffer (allocated in %.1f ms): %.2f ms (%.1f GB/s)\n", mb, alloc_ms, ms(t0, now()),
                   (double)((size_t)54 << 20) / ms(t0, now()) / 1e6);
        }
        CK(hipHostFree(pin));
    }
    // row slabs (contiguous): pieces of 1/4 of the rows
    for (int rep = 0; rep < 2; ++rep)
        for (int b = 3; b >= 0; --b) {
            const size_t off = b * (bytes / 4);
            auto t0 = now();
            CK(hipMemcpyAsync(d + off, h + off, bytes / 4, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            printf("row slab %d 1-D pageable H2D: %.2f ms (%.1f GB/s)\n", b, ms(t0, now()), bytes / 4 / ms(t0, now()) / 1e6);
        }
    // is the host free while a pageable async copy runs?  (time to return from the call)
    {
        auto t0 = now();
        CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st));
        auto t1 = now();
        CK(hipStreamSynchronize(st));
        printf("pageable hipMemcpyAsync returns after %.2f ms of %.2f ms\n", ms(t0, t1), ms(t0, now()));
    }
    return 0;
}
