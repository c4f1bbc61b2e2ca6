// VALU throughput of the FFT's own arithmetic (dft16 + 15 twiddle multiplies, all in registers, no memory) at 1..5
// wavefronts per SIMD: what the radix-16 butterflies can reach when nothing waits.  hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../passiveradar_amd/csrc/fft_wave.h"
void prc_set_error(const char*, ...) {}
__global__ __launch_bounds__(256) void k(float2* out, int iters, float2 w) {
    extern __shared__ float2 lds[];
    float2 x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = make_float2((float)(threadIdx.x + r), (float)r * 0.5f);
    for (int i = 0; i < iters; ++i) {
        dft16<1>(x);
#pragma unroll
        for (int r = 1; r < 16; ++r) x[r] = mul_tw<1>(x[r], w);
#pragma unroll
        for (int r = 0; r < 16; ++r) { x[r].x *= 0.25f; x[r].y *= 0.25f; }
    }
    float2 s = make_float2(0.f, 0.f);
#pragma unroll
    for (int r = 0; r < 16; ++r) { s.x += x[r].x; s.y += x[r].y; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (iters < 0) lds[threadIdx.x] = s;
}
int main() {
    float2* d; hipMalloc(&d, 1 << 26);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 4000;
    const int valu_per_iter = 0;   // filled from the ISA listing by the reader: see tools/isa_stats.py
    for (int wps = 1; wps <= 5; ++wps) {
        const size_t lds = (160 * 1024) / wps - 512;
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            k<<<256 * wps, 256, lds>>>(d, iters, make_float2(0.6f, 0.8f));
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        // per SIMD: wps waves each running iters iterations
        printf("waves/SIMD %d: %.3f ms -> %.1f ns per iteration per wave-slot, %.1f cycles@2.4GHz per iteration per SIMD\n", wps, ms,
               ms * 1e6 / iters, ms * 1e-3 * 2.4e9 / ((double)iters * wps));
        (void)valu_per_iter;
    }
    return 0;
}
