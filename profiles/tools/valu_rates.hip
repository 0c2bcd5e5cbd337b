// Issue cost of single vector instructions on gfx950 (MI355X): ns and cycles per wave64 instruction per SIMD with
// 4 waves per SIMD and 8 independent chains per wave (the pipe is never waiting for a result).
//   hipcc --offload-arch=gfx950 -O3 -w -o build/diag/valu_rates profiles/tools/valu_rates.hip && build/diag/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define KINDS(X)                                                                                  \
    X(0, "v_add_f32 v, v, v", "v_add_f32 %0, %0, %1", "v")                                         \
    X(1, "v_add_f32 v, literal, v", "v_add_f32 %0, 0x4b400000, %0", "v")                           \
    X(2, "v_fma_f32 v, v, v, v", "v_fma_f32 %0, %0, %1, %4", "v")                                  \
    X(3, "v_fma_f32 v, v, v, s", "v_fma_f32 %0, %0, %1, %2", "v")                                  \
    X(4, "v_min_f32", "v_min_f32 %0, %0, %1", "v")                                                 \
    X(5, "v_rsq_f32", "v_rsq_f32 %0, %0", "v")                                                     \
    X(6, "v_sqrt_f32", "v_sqrt_f32 %0, %0", "v")                                                   \
    X(7, "v_rcp_f32", "v_rcp_f32 %0, %0", "v")                                                     \
    X(8, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 2, %1", "v")                                    \
    X(9, "v_lshlrev_b32", "v_lshlrev_b32 %0, 2, %0", "v")                                          \
    X(10, "v_add_u32", "v_add_u32 %0, %0, %1", "v")                                                \
    X(11, "v_and_b32", "v_and_b32 %0, %0, %1", "v")                                                \
    X(12, "v_cmp_gt_f32 (vcc)", "v_cmp_gt_f32 vcc, %0, %1", "v")                                   \
    X(13, "v_cmp_gt_f32_e64 |v| -> sgpr pair", "v_cmp_gt_f32_e64 s[20:21], |%0|, %1", "v")         \
    X(14, "v_fract_f32", "v_fract_f32 %0, %0", "v")                                                \
    X(15, "v_rndne_f32", "v_rndne_f32 %0, %0", "v")                                                \
    X(16, "v_cvt_i32_f32", "v_cvt_i32_f32 %0, %0", "v")                                            \
    X(17, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %4", "v")                                    \
    X(18, "v_bfe_u32", "v_bfe_u32 %0, %0, 2, 9", "v")                                              \
    X(19, "v_max3_f32", "v_max3_f32 %0, %0, %1, %4", "v")                                          \
    X(20, "v_med3_f32", "v_med3_f32 %0, %0, %1, %4", "v")                                          \
    X(21, "v_mul_f32", "v_mul_f32 %0, %0, %1", "v")                                                \
    X(22, "v_fmac_f32", "v_fmac_f32 %0, %1, %4", "v")                                              \
    X(23, "v_sub_f32", "v_sub_f32 %0, %0, %1", "v")                                                \
    X(24, "v_cndmask_b32 (vcc)", "v_cndmask_b32 %0, %0, %1, vcc", "v")                             \
    X(25, "v_fma_f32 clamp", "v_fma_f32 %0, %0, %1, %4 clamp", "v")                                \
    X(26, "v_add_f32 v, s, v", "v_add_f32 %0, %2, %0", "v")                                        \
    X(27, "v_mov_b32", "v_mov_b32 %0, %1", "v")                                                    \
    X(28, "v_lshrrev_b32", "v_lshrrev_b32 %0, 2, %0", "v")                                         \
    X(29, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %4", "v")                                      \
    X(30, "v_fmaak_f32 (literal addend)", "v_fmaak_f32 %0, %0, %1, 0x4b400000", "v")               \
    X(31, "v_add_f32 v, v, v (dpp row_shr:1)", "v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "v") \
    X(32, "v_cvt_f32_u32", "v_cvt_f32_u32 %0, %0", "v")                                            \
    X(33, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1", "v")                                          \
    X(34, "v_add_f64", "v_add_f64 %3, %3, %3", "v")                                                \
    X(35, "v_fma_f64", "v_fma_f64 %3, %3, %3, %3", "v")                                            \
    X(36, "v_exp_f32", "v_exp_f32 %0, %0", "v")                                                    \
    X(37, "v_pk_add_f32", "v_pk_add_f32 %3, %3, %3", "v")

template <int KIND>
__global__ __launch_bounds__(1024) void probe(float *out, int iters)
{
    float a[8];
    double d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = threadIdx.x * 1e-3f + i + 1.f;
        d[i] = a[i];
    }
    float c = 1.0000001f, c2 = 0.9999999f;
    asm volatile("" : "+v"(c), "+v"(c2));
    float sc;
    asm volatile("s_mov_b32 %0, 0x3f800001" : "=s"(sc));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#define X(id, name, text, cons) \
    if (KIND == id) asm volatile(text : "+v"(a[i]) : "v"(c), "s"(sc), "v"(d[i]), "v"(c2) : "vcc", "s20", "s21");
                KINDS(X)
#undef X
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND> static void run(const char *name, float *d)
{
    const int iters = 1000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<KIND><<<256, 1024>>>(d, 10);         // 16 waves per CU = 4 per SIMD, one workgroup per CU
    hipEventRecord(e0);
    probe<KIND><<<256, 1024>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e6 / ((double)iters * 64 * 4);     // instructions per SIMD = iters x 64 x 4 waves
    printf("%-40s %.2f ns per wave instruction per SIMD\n", name, per);
}

int main()
{
    float *d;
    hipMalloc(&d, 256 * 1024 * 4);
    for (int rep = 0; rep < 2; ++rep) {
#define X(id, name, text, cons) run<id>(name, d);
        KINDS(X)
#undef X
    }
    return 0;
}
