// valu_microbench.hip -- issue cost (cycles per wave-instruction per SIMD) of the
// VALU ops the blend loop is made of (profiles/archive/r01_valu_microbench.txt).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float *out, int iters, float s) {
    float a[8]; unsigned u[8]; f2 p[4];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; u[i] = threadIdx.x * 7919u + i; }
    for (int i = 0; i < 4; ++i) p[i] = f2{a[i], a[i + 4]};
    f2 sv = {s, s};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE == 0) {  // v_fma_f32 x8 independent
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], s, 1.0f);
            } else if (MODE == 1) {  // v_pk_fma_f32 x4 (= 8 fma)
#pragma unroll
                for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma(p[i], sv, sv);
            } else if (MODE == 2) {  // v_and_b32 x8
#pragma unroll
                for (int i = 0; i < 8; ++i) { asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[i])); }
            } else if (MODE == 3) {  // v_lshlrev_b32 x8
#pragma unroll
                for (int i = 0; i < 8; ++i) { asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[i])); }
            } else if (MODE == 4) {  // v_mul_f32 x8
#pragma unroll
                for (int i = 0; i < 8; ++i) { asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(s)); }
            } else if (MODE == 5) {  // v_cvt_pk_bf16_f32 x8
#pragma unroll
                for (int i = 0; i < 8; ++i) { asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s)); }
            } else if (MODE == 7) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(65536u)); }
            } else if (MODE == 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(u[i]) : "v"(0u)); }
            } else if (MODE == 9) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { asm volatile("v_lshl_add_u32 %0, %0, 16, %1" : "+v"(u[i]) : "v"(0u)); }
            } else if (MODE == 10) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(u[i]) : "v"(0u)); }
            } else if (MODE == 11) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[i]) : "v"(65536u), "v"(0u)); }
            } else if (MODE == 12) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { asm volatile("v_cvt_f32_bf16 %0, %0" : "+v"(u[i])); }
            } else if (MODE == 13) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "+v"(u[i]) : "v"(16u)); }
            } else if (MODE == 14) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { asm volatile("v_mov_b32_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0" : "+v"(u[i])); }
            } else if (MODE == 6) {  // v_perm_b32 x8
#pragma unroll
                for (int i = 0; i < 8; ++i) { asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "s"(0x07060302)); }
            }
        }
    }
    float acc = 0; for (int i = 0; i < 8; ++i) acc += a[i] + (float)u[i]; for (int i = 0; i < 4; ++i) acc += p[i].x + p[i].y;
    if (acc == 12345.678f) out[0] = acc;
}
template <int MODE> int run(const char *name, float *out) {
    const int iters = 2000;
    for (int waves_per_simd : {1, 2, 4}) {
        dim3 grid(256), block(256 * waves_per_simd);  // one block per CU; block = 4*w waves
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        k<MODE><<<grid, block>>>(out, 10, 1.0001f); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a)); k<MODE><<<grid, block>>>(out, iters, 1.0001f); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        double instr_per_wave = (double)iters * 64 * (MODE == 1 ? 0.5 : 1.0);
        double ns_per_instr_simd = ms * 1e6 / (instr_per_wave * waves_per_simd);
        printf("%-18s waves/SIMD %d : %.3f ms  %.2f ns per wave-instr per SIMD (x clock GHz = cycles)\n", name, waves_per_simd, ms, ns_per_instr_simd);
    }
    return 0;
}
int main() {
    float *out; CK(hipMalloc(&out, 64));
    run<0>("v_fma_f32", out); run<1>("v_pk_fma_f32", out); run<2>("v_and_b32", out); run<3>("v_lshlrev_b32", out);
    run<4>("v_mul_f32", out); run<5>("v_cvt_pk_bf16_f32", out); run<6>("v_perm_b32", out);
    run<7>("v_mul_u32_u24", out); run<8>("v_alignbit_b32", out); run<9>("v_lshl_add_u32", out);
    run<10>("v_lshl_or_b32", out); run<11>("v_mad_u32_u24", out);
#ifdef TRY_EXOTIC
    run<12>("v_cvt_f32_bf16", out); run<13>("v_lshlrev_sdwa", out); run<14>("v_mov_sdwa_w1", out);
#endif
    return 0;
}
