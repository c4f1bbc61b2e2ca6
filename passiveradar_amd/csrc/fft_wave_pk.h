// fft1024_fwd / fft1024_inv of fft_wave.h on packed-f32 instructions (FT_PK builds): the radix-16 passes and the twiddle
// products through fft_pk.h, the quad stage unchanged (its butterflies are single DPP instructions on the components).
// Same stages, same roundings, same layouts.
#pragma once
#include "fft_pk.h"
#ifdef FT_PK
// w += conj(u) * v and a * s (s real) as packed instructions
__device__ __forceinline__ void cmac_conj_a(float2& w, float2 u, float2 v) {
    v2f ww = pk_from(w);
    pk_cmac_conj_a(ww, pk_from(u), pk_from(v));
    w = pk_to(ww);
}
__device__ __forceinline__ float2 cscale(float2 a, float s) { return pk_to(pk_scale(pk_from(a), s)); }

template <int NZ = 16>
__device__ __forceinline__ void fft1024_fwd(float2 (&xs)[16], float2* tile, const float2* tab, const FftLane& f) {
    v2f x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = pk_from(xs[r]);
    pk_dft16<1, NZ>(x);
    pk_twiddle<1, 1>(x, [&](int k1) { return pk_from(tab[k1 * 64 + f.lane]); });
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) tile[k1 * FFTW_PITCH + f.lane] = pk_to(x[k1]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < 16; ++m) x[m] = pk_from(tile[f.rd_off + 4 * m]);
    __builtin_amdgcn_wave_barrier();
    pk_dft16<1>(x);
    const float2* tw2s = tab + FFTW_TW1 + FFTW_TW2 + (f.lane & 3);   // sigma_j W_64^(j m')
    x[0] = pk_scale(x[0], f.sg);
    pk_twiddle<1, 1>(x, [&](int m) { return pk_from(tw2s[4 * m]); });
#pragma unroll
    for (int r = 0; r < 16; ++r) xs[r] = pk_to(x[r]);
    quad_dft4_fwd<1>(xs, f);
}

template <bool PRESCALED = false>
__device__ __forceinline__ void fft1024_inv(float2 (&xs)[16], float2* tile, const float2* tab, const FftLane& f) {
    if (!PRESCALED) {
#pragma unroll
        for (int m = 0; m < 16; ++m) xs[m] = pk_to(pk_scale(pk_from(xs[m]), f.sg));
    }
    quad_dft4_bwd<-1>(xs, f);
    v2f x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = pk_from(xs[r]);
    const float2* tw2 = tab + FFTW_TW1 + (f.lane & 3);
    pk_twiddle<-1, 1>(x, [&](int m) { return pk_from(tw2[4 * m]); });
    pk_dft16<-1>(x);
#pragma unroll
    for (int m = 0; m < 16; ++m) tile[f.rd_off + 4 * m] = pk_to(x[m]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) x[k1] = pk_from(tile[k1 * FFTW_PITCH + f.lane]);
    __builtin_amdgcn_wave_barrier();
    pk_twiddle<-1, 1>(x, [&](int k1) { return pk_from(tab[k1 * 64 + f.lane]); });
    pk_dft16<-1>(x);
#pragma unroll
    for (int r = 0; r < 16; ++r) xs[r] = pk_to(x[r]);
}
#endif
