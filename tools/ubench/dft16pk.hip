// Packed-f32 radix-16 butterflies (fft_pk.h) against the scalar dft16 of fft_wave.h: (1) results bit for bit on random
// data, forward and inverse, zero-tail variants included; (2) VALU time of dft16 + 15 twiddle multiplies in registers at
// 1..5 wavefronts per SIMD, both forms.  hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../passiveradar_amd/csrc/fft_pk.h"
void prc_set_error(const char*, ...) {}

template <int DIR, int NZ>
__global__ void check_k(const float2* in, float2* out_s, float2* out_p, float2 w) {
    float2 x[16];
    v2f y[16];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        x[r] = r < NZ ? in[t * 16 + r] : make_float2(0.f, 0.f);
        y[r] = pk_from(x[r]);
    }
    dft16<DIR, NZ>(x);
    pk_dft16<DIR, NZ>(y);
#pragma unroll
    for (int r = 1; r < 16; ++r) { x[r] = mul_tw<DIR>(x[r], w); y[r] = pk_mul_tw<DIR>(y[r], pk_from(w)); }
    float2 a = make_float2(0.25f, -0.5f);
    v2f b = pk_from(a);
#pragma unroll
    for (int r = 0; r < 16; ++r) { cmac_conj_a(a, x[r], x[(r + 3) & 15]); pk_cmac_conj_a(b, y[r], y[(r + 3) & 15]); }
#pragma unroll
    for (int r = 0; r < 16; ++r) { out_s[t * 16 + r] = x[r]; out_p[t * 16 + r] = pk_to(y[r]); }
    out_s[t * 16].x += a.x; out_s[t * 16].y += a.y;
    const float2 bb = pk_to(b);
    out_p[t * 16].x += bb.x; out_p[t * 16].y += bb.y;
}

template <bool PK>
__global__ __launch_bounds__(256) void time_k(float2* out, int iters, float2 w) {
    extern __shared__ float2 lds[];
    float2 s = make_float2(0.f, 0.f);
    if (!PK) {
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
#pragma unroll
        for (int r = 0; r < 16; ++r) { s.x += x[r].x; s.y += x[r].y; }
    } else {
        v2f x[16];
        const v2f wv = pk_from(w);
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = v2f{(float)(threadIdx.x + r), (float)r * 0.5f};
        for (int i = 0; i < iters; ++i) {
            pk_dft16<1>(x);
            pk_twiddle<1, 1>(x, [&](int) { return wv; });
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = pk_scale(x[r], 0.25f);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { s.x += x[r].x; s.y += x[r].y; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (iters < 0) lds[threadIdx.x] = s;
}

template <int DIR, int NZ>
static int check(const float2* d_in, float2* d_s, float2* d_p, int n) {
    check_k<DIR, NZ><<<n / 256, 256>>>(d_in, d_s, d_p, make_float2(0.6f, 0.8f));
    float2* hs = (float2*)malloc(sizeof(float2) * n * 16);
    float2* hp = (float2*)malloc(sizeof(float2) * n * 16);
    hipMemcpy(hs, d_s, sizeof(float2) * n * 16, hipMemcpyDeviceToHost);
    hipMemcpy(hp, d_p, sizeof(float2) * n * 16, hipMemcpyDeviceToHost);
    const int bad = memcmp(hs, hp, sizeof(float2) * n * 16) != 0;
    double worst = 0;
    for (int i = 0; i < n * 16; ++i) {
        const double e = fabs((double)hs[i].x - hp[i].x) + fabs((double)hs[i].y - hp[i].y);
        if (e > worst) worst = e;
    }
    printf("dft16<%d,%d> + twiddle + cmac: packed vs scalar %s (worst |diff| %.3g)\n", DIR, NZ, bad ? "DIFFER" : "bit-identical", worst);
    free(hs); free(hp);
    return bad;
}

int main() {
    const int n = 4096;
    float2* h = (float2*)malloc(sizeof(float2) * n * 16);
    srand(7);
    for (int i = 0; i < n * 16; ++i) h[i] = make_float2(rand() / (float)RAND_MAX - 0.5f, rand() / (float)RAND_MAX - 0.5f);
    float2 *d_in, *d_s, *d_p;
    hipMalloc(&d_in, sizeof(float2) * n * 16); hipMalloc(&d_s, sizeof(float2) * n * 16); hipMalloc(&d_p, sizeof(float2) * n * 16);
    hipMemcpy(d_in, h, sizeof(float2) * n * 16, hipMemcpyHostToDevice);
    int bad = 0;
    bad |= check<1, 16>(d_in, d_s, d_p, n);
    bad |= check<-1, 16>(d_in, d_s, d_p, n);
    bad |= check<1, 12>(d_in, d_s, d_p, n);
    bad |= check<1, 8>(d_in, d_s, d_p, n);

    float2* d; hipMalloc(&d, 1 << 26);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)time_k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)time_k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 4000;
    for (int pk = 0; pk < 2; ++pk)
        for (int wps = 1; wps <= 5; ++wps) {
            const size_t lds = (160 * 1024) / wps - 512;
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (pk) time_k<true><<<256 * wps, 256, lds>>>(d, iters, make_float2(0.6f, 0.8f));
                else time_k<false><<<256 * wps, 256, lds>>>(d, iters, make_float2(0.6f, 0.8f));
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            printf("%s waves/SIMD %d: %.3f ms -> %.1f cycles@2.4GHz per iteration per SIMD\n", pk ? "packed" : "scalar", wps, ms,
                   ms * 1e-3 * 2.4e9 / ((double)iters * wps));
        }
    return bad;
}
