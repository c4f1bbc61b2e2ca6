// VALU issue-rate microbenchmark: v_fma_f32 vs v_pk_fma_f32 (gfx950), one wave per SIMD and two.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    v2f p0 = {a0, 1}, p1 = {a1, 2}, p2 = {a2, 3}, p3 = {a3, 4}, p4 = {1, 2}, p5 = {3, 4}, p6 = {5, 6}, p7 = {7, 8};
    const float m = 0.999f, c = 0.001f;
    const v2f pm = {m, m}, pc = {c, c};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                             "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y +
                                                 p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
}
int main() {
    float* d; hipMalloc(&d, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wpb : {256, 512, 1024}) for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) k<0><<<256, wpb>>>(d, iters); else k<1><<<256, wpb>>>(d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double instr = (double)iters * 64;                       // VALU instructions per wave
        double waves_per_simd = wpb / 64 / 4.0;
        double cyc = ms * 1e-3 * 2.4e9 / (instr * waves_per_simd);
        printf("%s  waves/SIMD %.0f: %.3f ms, %.2f cycles per instruction per SIMD (at 2.4 GHz), %.1f TFLOP/s\n", mode ? "v_pk_fma_f32" : "v_fma_f32   ",
               waves_per_simd, ms, cyc, (mode ? 4.0 : 2.0) * 64 * instr * (wpb / 64) * 256 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
