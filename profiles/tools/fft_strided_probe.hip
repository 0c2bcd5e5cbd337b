// Probe: the inverse transform over k of the phase shift (TK [k][tau] complex64, 8192 x 8192, transform along k = stride
// snum) as rocFFT runs it in place, against (B) the same plan writing its output transposed ([tau][k], contiguous per
// transform) and (C) an own transpose followed by a contiguous in-place plan.
//   hipcc -O2 --offload-arch=gfx950 profiles/tools/fft_strided_probe.hip -o build/probe/fft_strided_probe -lrocfft
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define FK(x) do { rocfft_status s = (x); if (s != rocfft_status_success) { printf("%s: rocfft %d\n", #x, (int)s); return 1; } } while (0)

__global__ __launch_bounds__(256) void transpose_c(const float2 *__restrict__ in, float2 *__restrict__ out, int rows, int cols)
{
    __shared__ float2 t[64][65];
    const int bx = blockIdx.x * 64, by = blockIdx.y * 64, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4)
        if (by + r < rows && bx + tx < cols) t[r][tx] = in[(size_t)(by + r) * cols + bx + tx];
    __syncthreads();
    for (int r = ty; r < 64; r += 4)
        if (bx + r < cols && by + tx < rows) out[(size_t)(bx + r) * rows + by + tx] = t[tx][r];
}

struct Plan {
    rocfft_plan p = nullptr;
    rocfft_execution_info info = nullptr;
    void *work = nullptr;
};
static int make(Plan &pl, bool inplace, size_t len, size_t batch, size_t is, size_t id, size_t os, size_t od, hipStream_t st,
                rocfft_transform_type type = rocfft_transform_type_complex_inverse)
{
    rocfft_plan_description d = nullptr;
    FK(rocfft_plan_description_create(&d));
    FK(rocfft_plan_description_set_data_layout(d, rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved,
                                               nullptr, nullptr, 1, &is, id, 1, &os, od));
    FK(rocfft_plan_create(&pl.p, inplace ? rocfft_placement_inplace : rocfft_placement_notinplace, type,
                          rocfft_precision_single, 1, &len, batch, d));
    rocfft_plan_description_destroy(d);
    FK(rocfft_execution_info_create(&pl.info));
    FK(rocfft_execution_info_set_stream(pl.info, st));
    size_t wb = 0;
    FK(rocfft_plan_get_work_buffer_size(pl.p, &wb));
    if (wb) {
        CK(hipMalloc(&pl.work, wb));
        FK(rocfft_execution_info_set_work_buffer(pl.info, pl.work, wb));
    }
    printf("  (work buffer %zu MB)\n", wb >> 20);
    return 0;
}

int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? atoi(argv[1]) : 8192, nf = n / 2;
    FK(rocfft_setup());
    hipStream_t st;
    CK(hipStreamCreate(&st));
    float2 *a, *b;
    CK(hipMalloc(&a, n * n * 8));
    CK(hipMalloc(&b, n * n * 8));
    CK(hipMemset(a, 0, n * n * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto fn) {
        for (int i = 0; i < 2; ++i) fn();
        hipEventRecord(e0, st);
        for (int i = 0; i < 5; ++i) fn();
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-70s %.3f ms\n", name, ms / 5);
    };
    Plan A, B, C, D, E, F;
    printf("A: in place, stride n dist 1\n");
    if (make(A, true, n, n, n, 1, n, 1, st)) return 1;
    printf("B: out of place, in stride n dist 1 -> out stride 1 dist n\n");
    if (make(B, false, n, n, n, 1, 1, n, st)) return 1;
    printf("C: in place contiguous, stride 1 dist n\n");
    if (make(C, true, n, n, 1, n, 1, n, st)) return 1;
    printf("D: forward over x of the half spectrum [x][nf]: in place, length n, batch nf, stride nf dist 1\n");
    if (make(D, true, n, nf, nf, 1, nf, 1, st, rocfft_transform_type_complex_forward)) return 1;
    printf("E: the same out of place -> out stride nf dist 1\n");
    if (make(E, false, n, nf, nf, 1, nf, 1, st, rocfft_transform_type_complex_forward)) return 1;
    printf("F: forward contiguous: length n, batch nf, stride 1 dist n\n");
    if (make(F, true, n, nf, 1, n, 1, n, st, rocfft_transform_type_complex_forward)) return 1;
    void *ia[1] = {a}, *ob[1] = {b};
    timeit("A inverse over k in place, strided (today)", [&] { rocfft_execute(A.p, ia, nullptr, A.info); });
    timeit("B inverse over k, strided in -> contiguous transposed out", [&] { rocfft_execute(B.p, ia, ob, B.info); });
    timeit("C1 own transpose [k][tau] -> [tau][k]", [&] { hipLaunchKernelGGL(transpose_c, dim3(n / 64, n / 64), dim3(256), 0, st, a, b, (int)n, (int)n); });
    void *ib[1] = {b};
    timeit("C2 inverse over k contiguous in place", [&] { rocfft_execute(C.p, ib, nullptr, C.info); });
    timeit("D forward over x in place, strided (today)", [&] { rocfft_execute(D.p, ia, nullptr, D.info); });
    timeit("E forward over x out of place, strided", [&] { rocfft_execute(E.p, ia, ob, E.info); });
    timeit("F1 own transpose [x][nf] -> [nf][x]", [&] { hipLaunchKernelGGL(transpose_c, dim3(nf / 64, n / 64), dim3(256), 0, st, a, b, (int)n, (int)nf); });
    timeit("F2 forward over x contiguous in place", [&] { rocfft_execute(F.p, ib, nullptr, F.info); });
    timeit("F3 own transpose back [nf][x] -> [k][nf]", [&] { hipLaunchKernelGGL(transpose_c, dim3(n / 64, nf / 64), dim3(256), 0, st, b, a, (int)nf, (int)n); });
    return 0;
}
