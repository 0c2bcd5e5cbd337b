// Probe: upload of a FRESH pageable host array (allocated and written, never seen by the runtime) -- what a radargram
// just read from a file is.  One hipMemcpyAsync against slices issued from several host threads on their own streams.
//   hipcc -O2 profiles/tools/h2d_fresh_probe.cpp -o build/probe/h2d_fresh_probe -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main()
{
    const size_t bytes = (size_t)4096 * 10000 * 4;
    char *d;
    CK(hipMalloc(&d, bytes));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    hipStream_t st[16];
    for (auto &s : st) CK(hipStreamCreate(&s));
    auto fresh = [&]() {
        char *h = (char *)malloc(bytes);
        memset(h, 1, bytes);
        return h;
    };
    for (int rep = 0; rep < 3; ++rep) {
        char *h = fresh();
        auto t0 = now();
        CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st[0]));
        CK(hipStreamSynchronize(st[0]));
        printf("fresh array, one copy: %.2f ms (%.1f GB/s)\n", ms(t0, now()), bytes / ms(t0, now()) / 1e6);
        auto t1 = now();
        CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st[0]));
        CK(hipStreamSynchronize(st[0]));
        printf("  same array again:    %.2f ms\n", ms(t1, now()));
        free(h);
    }
    for (int nthr : {2, 4, 8, 16})
        for (int rep = 0; rep < 2; ++rep) {
            char *h = fresh();
            auto t0 = now();
            std::vector<std::thread> pool;
            for (int t = 0; t < nthr; ++t)
                pool.emplace_back([&, t] {
                    (void)hipSetDevice(0);
                    const size_t a = bytes * t / nthr / 4096 * 4096, b = t + 1 == nthr ? bytes : bytes * (t + 1) / nthr / 4096 * 4096;
                    (void)hipMemcpyAsync(d + a, h + a, b - a, hipMemcpyHostToDevice, st[t]);
                    (void)hipStreamSynchronize(st[t]);
                });
            for (auto &th : pool) th.join();
            printf("fresh array, %2d threads x slices: %.2f ms (%.1f GB/s)\n", nthr, ms(t0, now()), bytes / ms(t0, now()) / 1e6);
            free(h);
        }
    // one thread, slices issued one after another on one stream (does the runtime pipeline pin + copy?)
    for (int np : {4, 16})
        for (int rep = 0; rep < 2; ++rep) {
            char *h = fresh();
            auto t0 = now();
            for (int t = 0; t < np; ++t) {
                const size_t a = bytes * t / np / 4096 * 4096, b = t + 1 == np ? bytes : bytes * (t + 1) / np / 4096 * 4096;
                CK(hipMemcpyAsync(d + a, h + a, b - a, hipMemcpyHostToDevice, st[0]));
            }
            CK(hipStreamSynchronize(st[0]));
            printf("fresh array, one thread, %2d slices in a row: %.2f ms\n", np, ms(t0, now()));
            free(h);
        }
    // 2-D column blocks (what the pipelined one-shot call issues) from a fresh array
    for (int rep = 0; rep < 2; ++rep) {
        char *h = fresh();
        const size_t tnum = 10000, snum = 4096, esz = 4, cuts[] = {0, 4460, 7460, 10000};
        for (int b = 0; b < 3; ++b) {
            auto t0 = now();
            CK(hipMemcpy2DAsync(d + cuts[b] * esz, tnum * esz, h + cuts[b] * esz, tnum * esz, (cuts[b + 1] - cuts[b]) * esz, snum,
                                hipMemcpyHostToDevice, st[0]));
            auto t1 = now();
            CK(hipStreamSynchronize(st[0]));
            printf("fresh array, 2-D traces [%zu, %zu): call returns after %.2f ms, done after %.2f ms\n", cuts[b], cuts[b + 1], ms(t0, t1), ms(t0, now()));
        }
        free(h);
    }
    return 0;
}
