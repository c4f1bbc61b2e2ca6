// VALU issue-rate microbenchmark (gfx950): v_pk_fma_f32 whose multiplier is a VGPR pair against one whose multiplier is an
// SGPR pair read through op_sel / op_sel_hi (how the front end's row loop takes its taps: frontend.hip, feg_trips), and the
// same with an LDS read per 13 multiply-adds next to them.      hipcc --offload-arch=gfx950 -O3 pkfma_sgpr.hip -o pkfma_sgpr
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
// MODE 0: VGPR multiplier; 1: SGPR pair, low half (op_sel_hi:[0,1,1]); 2: SGPR pair, alternating low / high half;
// 3: as 2 with one ds_read_b64 per 13 multiply-adds feeding the next 13 (the row loop's shape, taps resident)
template <int MODE>
__global__ void k(float* out, int iters, float m0, float m1) {
    __shared__ float2 lds[1024];
    lds[threadIdx.x & 1023] = make_float2(threadIdx.x, 1.f);
    __syncthreads();
    v2f acc[13];
#pragma unroll
    for (int q = 0; q < 13; ++q) acc[q] = v2f{(float)q, (float)threadIdx.x};
    v2f x = {1.0f + threadIdx.x * 1e-6f, 0.5f};
    const v2f tv = {m0, m1};
    v2f ts[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) ts[j] = v2f{__builtin_amdgcn_readfirstlane(__float_as_int(m0 + j)) * 1e-9f, __builtin_amdgcn_readfirstlane(__float_as_int(m1 - j)) * 1e-9f};
    const float2* lp = lds + (threadIdx.x & 63);
    for (int i = 0; i < iters; ++i) {
        if (MODE == 3) {
            const float2 r = lp[(i & 7) * 64];
            x = v2f{r.x, r.y};
        }
#pragma unroll
        for (int q = 0; q < 13; ++q) {
            if (MODE == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(tv), "v"(x));
            else if (MODE == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[q]) : "s"(ts[q >> 1]), "v"(x));
            else if (MODE == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[q]) : "s"(ts[q >> 1]), "v"(x));
            else if (MODE == 7) asm volatile("v_fmac_f32 %0, %2, %3\n v_fmac_f32 %1, %2, %4" : "+v"(acc[q].x), "+v"(acc[q].y) : "s"(ts[q >> 1].x), "v"(x.x), "v"(x.y));
            else if (MODE == 8) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[q]) : "s"(ts[q % 7]), "v"(x));
            else if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[q]) : "v"(tv), "v"(x));
            else if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[q]) : "s"(ts[q >> 1]), "v"(x));
            else if (q & 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[q]) : "s"(ts[q >> 1]), "v"(x));
            else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[q]) : "s"(ts[q >> 1]), "v"(x));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 13; ++q) s += acc[q].x + acc[q].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 40000;
    const char* names[9] = {"VGPR multiplier, low half broadcast (op_sel_hi:[0,1,1])", "SGPR pair, low half broadcast                          ",
                            "SGPR pair, low / high half broadcast in turn           ", "as before + one ds_read_b64 per 13                     ",
                            "VGPR multiplier, no modifiers                          ", "SGPR pair multiplier, no modifiers                     ",
                            "SGPR pair, x low half broadcast (op_sel_hi:[1,0,1])    ", "2 x v_fmac_f32 with an SGPR multiplier                 ",
                            "SGPR pair, low half broadcast, a new pair every time   "};
    for (int wpb : {256, 512, 1024}) for (int mode = 0; mode < 9; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) k<0><<<256, wpb>>>(d, iters, 0.999f, 0.001f);
            else if (mode == 1) k<1><<<256, wpb>>>(d, iters, 0.999f, 0.001f);
            else if (mode == 2) k<2><<<256, wpb>>>(d, iters, 0.999f, 0.001f);
            else if (mode == 3) k<3><<<256, wpb>>>(d, iters, 0.999f, 0.001f);
            else if (mode == 4) k<4><<<256, wpb>>>(d, iters, 0.999f, 0.001f);
            else if (mode == 5) k<5><<<256, wpb>>>(d, iters, 0.999f, 0.001f);
            else if (mode == 6) k<6><<<256, wpb>>>(d, iters, 0.999f, 0.001f);
            else if (mode == 7) k<7><<<256, wpb>>>(d, iters, 0.999f, 0.001f);
            else k<8><<<256, wpb>>>(d, iters, 0.999f, 0.001f);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)iters * 13;                 // packed multiply-adds per wave
        const double waves_per_simd = wpb / 64 / 4.0;
        printf("%s waves/SIMD %.0f: %.3f ms, %.2f cycles per complex multiply-add per SIMD (at 2.4 GHz), %.1f TFLOP/s\n", names[mode], waves_per_simd, ms,
               ms * 1e-3 * 2.4e9 / (instr * waves_per_simd), 4.0 * 64 * instr * (wpb / 64) * 256 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
