// VGPR bank probe for gfx950: does the issue rate of v_fma_f32 / v_fmac_f32 / v_mul_f32 depend on which registers the
// source operands live in (register number mod 4)?  Fixed registers through inline asm, 8 independent chains,
// 2 and 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o build/diag/vgpr_bank_probe profiles/tools/vgpr_bank_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X X X X X X X X
// chains in v[32..39]; sources at the stated registers
#define FMA(d, a, b) "v_fma_f32 v" #d ", v" #d ", v" #a ", v" #b "\n\t"
#define FMAC(d, a, b) "v_fmac_f32 v" #d ", v" #a ", v" #b "\n\t"
#define MUL(d, a) "v_mul_f32 v" #d ", v" #d ", v" #a "\n\t"

template <int KIND> __global__ void probe(float *out, int iters)
{
    float x = threadIdx.x * 1e-9f;
    asm volatile("v_mov_b32 v32, %0\n\tv_mov_b32 v33, %0\n\tv_mov_b32 v34, %0\n\tv_mov_b32 v35, %0\n\t"
                 "v_mov_b32 v36, %0\n\tv_mov_b32 v37, %0\n\tv_mov_b32 v38, %0\n\tv_mov_b32 v39, %0\n\t"
                 "v_mov_b32 v40, 0x3f7fbe77\n\tv_mov_b32 v41, 0x3f7fbe77\n\tv_mov_b32 v42, 0x3f7fbe77\n\tv_mov_b32 v43, 0x3f7fbe77\n\t"
                 "v_mov_b32 v44, 0x3a83126f\n\tv_mov_b32 v45, 0x3a83126f\n\tv_mov_b32 v46, 0x3a83126f\n\tv_mov_b32 v47, 0x3a83126f\n\t"
                 :: "v"(x) : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0)        // fma, all three operands in the destination's bank (d, d+8, d+12 -> same number mod 4)
            asm volatile(REP8(FMA(32, 40, 44) FMA(33, 41, 45) FMA(34, 42, 46) FMA(35, 43, 47) FMA(36, 40, 44) FMA(37, 41, 45) FMA(38, 42, 46) FMA(39, 43, 47))
                         ::: "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39");
        if (KIND == 1)        // fma, three different banks
            asm volatile(REP8(FMA(32, 41, 46) FMA(33, 42, 47) FMA(34, 43, 44) FMA(35, 40, 45) FMA(36, 41, 46) FMA(37, 42, 47) FMA(38, 43, 44) FMA(39, 40, 45))
                         ::: "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39");
        if (KIND == 2)        // fmac (VOP2), same bank
            asm volatile(REP8(FMAC(32, 40, 44) FMAC(33, 41, 45) FMAC(34, 42, 46) FMAC(35, 43, 47) FMAC(36, 40, 44) FMAC(37, 41, 45) FMAC(38, 42, 46) FMAC(39, 43, 47))
                         ::: "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39");
        if (KIND == 3)        // fmac, different banks
            asm volatile(REP8(FMAC(32, 41, 46) FMAC(33, 42, 47) FMAC(34, 43, 44) FMAC(35, 40, 45) FMAC(36, 41, 46) FMAC(37, 42, 47) FMAC(38, 43, 44) FMAC(39, 40, 45))
                         ::: "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39");
        if (KIND == 4)        // mul, same bank
            asm volatile(REP8(MUL(32, 40) MUL(33, 41) MUL(34, 42) MUL(35, 43) MUL(36, 40) MUL(37, 41) MUL(38, 42) MUL(39, 43))
                         ::: "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39");
        if (KIND == 5)        // mul, different banks
            asm volatile(REP8(MUL(32, 41) MUL(33, 42) MUL(34, 43) MUL(35, 40) MUL(36, 41) MUL(37, 42) MUL(38, 43) MUL(39, 40))
                         ::: "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39");
    }
    float s;
    asm volatile("v_add_f32 %0, v32, v33\n\tv_add_f32 %0, %0, v34\n\tv_add_f32 %0, %0, v35\n\tv_add_f32 %0, %0, v36\n\t"
                 "v_add_f32 %0, %0, v37\n\tv_add_f32 %0, %0, v38\n\tv_add_f32 %0, %0, v39" : "=v"(s));
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND> static void run(const char *name, float *d)
{
    const int iters = 500;
    for (int wps : {2, 8}) {
        const int grid = wps == 8 ? 512 : 256, blk = wps == 8 ? 1024 : 512;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        probe<KIND><<<grid, blk>>>(d, 10);
        hipEventRecord(e0);
        probe<KIND><<<grid, blk>>>(d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s waves/SIMD %d: %.3f ms, %.3f ns per wave-instr per SIMD\n", name, wps, ms, ms * 1e6 / ((double)iters * 64 * wps));
    }
}

int main()
{
    float *d;
    hipMalloc(&d, 1 << 24);
    run<0>("fma  same bank", d);
    run<1>("fma  three banks", d);
    run<2>("fmac same bank", d);
    run<3>("fmac three banks", d);
    run<4>("mul  same bank", d);
    run<5>("mul  two banks", d);
    return 0;
}
