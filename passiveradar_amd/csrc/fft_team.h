// 4096-point complex FFT by a team of four wavefronts (256 threads x 16 points) for gfx950.
//
// Why: the one-wavefront 1024-point FFT of fft_wave.h cuts a correlation into pieces of 1025 - lags
// samples.  With 1025 lags (config 3) or 2049 (config 5) that is 512 new samples per two or three
// transforms -- 7-15 FFT points per input sample.  4096 points keep pieces of 2048-3072 samples
// (4-5 points per sample) and carry least-squares filters of up to 3073 taps.
//
// Data flow (index for index in tools/fft4096_model.py, which also checks the bank behaviour of every LDS
// phase); t = thread 0..255, hi = t >> 4, lo = t & 15, r = register 0..15,
// n = 256 n1 + 16 n2 + n3, k = k1 + 16 k2 + 256 k3:
//   time layout      : thread t, register r  <->  sample 256 r + t               (coalesced global access)
//   frequency layout : thread t, register r  <->  bin hi + 16 lo + 256 r         (pointwise products only)
//   forward : S1 DFT16 over n1 (registers)            thread (n2, n3) = (hi, lo)
//             T1 twiddle W_256^(n2 k1)                 LDS table [k1][n2], broadcast reads
//             X1 exchange across the four waves        write (t, k1) | __syncthreads | read (16 m + lo, hi)
//             S2 DFT16 over n2                         thread (k1, n3) = (hi, lo)
//             T2 twiddle W_4096^(n3 (k1 + 16 k2))      16 per-thread constants
//             X2 exchange inside each 16-lane row      write (t, k2) | read (16 hi + j, lo); no workgroup barrier
//             S3 DFT16 over n3                         thread (k1, k2) = (hi, lo), register k3
//   inverse : the same stages backwards with conjugated twiddles, unnormalised (x 4096).
// Both exchange buffers are "thread major": element (thread tau, register rho) sits at 17 tau + rho, so every
// access is one per-thread base plus a compile-time offset (no address arithmetic in the transforms), every
// write and the row-private reads are bank-conflict free, and a wave only ever WRITES its own quarter of a
// buffer.  That last property is what makes ONE workgroup barrier per forward transform enough.  Transforms
// alternate the buffer they use for X1 (template parameter CUR); a forward transform keeps its X2 in the other
// buffer.  When a wave writes X1 of transform i into buffer T, every wave has passed the barrier of transform
// i-1 and therefore finished its cross-wave reads of T (X1 of transform i-2); a slower wave may still be busy
// with the row-private X2 of transform i-1 in T, but inside its own quarter, which this wave never touches.
// An inverse transform opens with a second barrier (its row-private exchange comes first and needs its buffer
// quiet), and a sequence that does not alternate (e.g. restarting at buffer 0 after an inverse that used
// buffer 0) must be separated by ft_team_sync().
#pragma once
#include "fft_wave.h"
#ifdef FT_PK
#include "fft_pk.h"
#endif

#define FT_P 4096
#define FT_THREADS 256
#define FT_PITCH 17                     // float2 per thread row of an exchange buffer
#define FT_BUF (FT_THREADS * FT_PITCH)  // float2 per exchange buffer
#define FT_TW1 256                      // W_256^(n2 k1), [k1][n2]
#ifndef FT_NBUF
#define FT_NBUF 2                       // exchange buffers: 2 = one barrier per forward transform; 1 = two barriers, half the LDS
#endif
#ifdef FT_TW2_LDS
#define FT_TW2_ELEMS FT_P               // the 16 T2 constants of every thread live in LDS ([k2][t], 32 KB) instead of 32 VGPRs:
#else                                   // for kernels held at two wavefronts per SIMD anyway (80 KB of LDS each)
#define FT_TW2_ELEMS 0
#endif
#define FT_LDS_ELEMS (FT_TW1 + FT_NBUF * FT_BUF + FT_TW2_ELEMS)   // float2 elements of LDS per workgroup (71 680 B with two buffers)
#define FT_GTAB (FT_TW1 + FT_P)         // global table: TW1 then W_4096^m, m = 0..4095

struct FtLane {
    int t;              // thread in the workgroup
    const float2* tw1;  // lds + hi                       : TW1[k1][n2 = hi] at tw1[16 k1]
    float2* wr;         // exchange base + 17 t           : write (t, rho) at wr[rho]            (+ FT_BUF for buffer 1)
    const float2* rdA;  // exchange base + 272 hi + lo    : row-private read (16 hi + m, lo) at rdA[17 m]
    const float2* rdB;  // exchange base + 17 lo + hi     : cross-wave read (16 m + lo, hi) at rdB[272 m]
#if defined(FT_TW2_LDS)
    const float2* tw2l; // lds + FT_TW1 + FT_NBUF FT_BUF + t : W_4096^(lo (hi + 16 k2)) at tw2l[256 k2]
#elif defined(FT_TW2_FACTORED)
    float2 tw2b;        // W_4096^(lo hi); the k2 part W_256^(lo k2) comes from the TW1 table (tw2t[16 k2])
    const float2* tw2t; // lds + lo
#else
    float2 tw2[16];     // W_4096^(lo (hi + 16 k2))
#endif
};

// T2 twiddle of register k2 (forward sense; the inverse multiplies by its conjugate)
__device__ __forceinline__ float2 ft_tw2(const FtLane& f, int k2) {
#if defined(FT_TW2_LDS)
    return f.tw2l[FT_THREADS * k2];
#elif defined(FT_TW2_FACTORED)
    return k2 == 0 ? f.tw2b : cmul(f.tw2b, f.tw2t[16 * k2]);
#else
    return f.tw2[k2];
#endif
}

// Fill the TW1 table in LDS and this thread's T2 constants from the host-made global table; ends with
// __syncthreads().  lds: FT_LDS_ELEMS float2.
__device__ __forceinline__ FtLane ft_setup(float2* lds, const float2* __restrict__ gtab) {
    FtLane f;
    f.t = threadIdx.x;
    const int hi = f.t >> 4, lo = f.t & 15;
    f.tw1 = lds + hi;
    float2* x = lds + FT_TW1;
    f.wr = x + FT_PITCH * f.t;
    f.rdA = x + 16 * FT_PITCH * hi + lo;
    f.rdB = x + FT_PITCH * lo + hi;
    lds[f.t] = gtab[f.t];               // FT_TW1 == FT_THREADS
#if defined(FT_TW2_LDS)
    {
        float2* tl = lds + FT_TW1 + FT_NBUF * FT_BUF + f.t;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) tl[FT_THREADS * k2] = gtab[FT_TW1 + ((lo * (hi + 16 * k2)) & (FT_P - 1))];
        f.tw2l = tl;
    }
#elif defined(FT_TW2_FACTORED)
    f.tw2b = gtab[FT_TW1 + lo * hi];
    f.tw2t = lds + lo;
#else
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) f.tw2[k2] = gtab[FT_TW1 + ((lo * (hi + 16 * k2)) & (FT_P - 1))];
#endif
    __syncthreads();
    return f;
}

#ifdef FT_EXP_NOBARRIER          // timing ablation only (wrong results)
#define FT_BARRIER() ((void)0)
#endif
#ifndef FT_BARRIER
#define FT_BARRIER() __syncthreads()
#endif
__device__ __forceinline__ void ft_team_sync() { FT_BARRIER(); }

// Forward FFT: time layout -> frequency layout.  CUR: buffer of the cross-wave exchange (alternate 0, 1, 0, ...).
// NZ: registers x[NZ..15] are zero in every thread (a zero-padded piece of at most 256 NZ samples)
#ifndef FT_PK
template <int CUR, int NZ = 16>
__device__ __forceinline__ void ft4096_fwd(float2 (&x)[16], const FtLane& f) {
    constexpr int T = (FT_NBUF == 2 ? CUR : 0) * FT_BUF, O = (FT_NBUF == 2 ? (CUR ^ 1) : 0) * FT_BUF;
    dft16<1, NZ>(x);
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) x[k1] = mul_tw<1>(x[k1], f.tw1[16 * k1]);
#ifndef FT_EXP_NOX1
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) f.wr[T + k1] = x[k1];
    FT_BARRIER();
#pragma unroll
    for (int m = 0; m < 16; ++m) x[m] = f.rdB[T + 16 * FT_PITCH * m];
    if (FT_NBUF == 1) FT_BARRIER();       // single buffer: every cross-wave read phase is closed by a barrier
#endif
    dft16<1>(x);
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) x[k2] = mul_tw<1>(x[k2], ft_tw2(f, k2));
#ifndef FT_EXP_NOX2
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) f.wr[O + k2] = x[k2];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = f.rdA[O + FT_PITCH * j];
    __builtin_amdgcn_wave_barrier();
#endif
    dft16<1>(x);
}
#else
// FT_PK: the same stages on packed-f32 instructions (fft_pk.h): half the VALU instructions, identical roundings
template <int CUR, int NZ = 16>
__device__ __forceinline__ void ft4096_fwd(float2 (&xs)[16], const FtLane& f) {
    constexpr int T = (FT_NBUF == 2 ? CUR : 0) * FT_BUF, O = (FT_NBUF == 2 ? (CUR ^ 1) : 0) * FT_BUF;
    v2f x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = pk_from(xs[r]);
    pk_dft16<1, NZ>(x);
    pk_twiddle<1, 1>(x, [&](int k1) { return pk_from(f.tw1[16 * k1]); });
#ifndef FT_EXP_NOX1
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) f.wr[T + k1] = pk_to(x[k1]);
    FT_BARRIER();
#pragma unroll
    for (int m = 0; m < 16; ++m) x[m] = pk_from(f.rdB[T + 16 * FT_PITCH * m]);
    if (FT_NBUF == 1) FT_BARRIER();
#endif
    pk_dft16<1>(x);
    pk_twiddle<1, 0>(x, [&](int k2) { return pk_from(ft_tw2(f, k2)); });
#ifndef FT_EXP_NOX2
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) f.wr[O + k2] = pk_to(x[k2]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = pk_from(f.rdA[O + FT_PITCH * j]);
    __builtin_amdgcn_wave_barrier();
#endif
    pk_dft16<1>(x);
#pragma unroll
    for (int r = 0; r < 16; ++r) xs[r] = pk_to(x[r]);
}
#endif

// Inverse FFT (unnormalised): frequency layout -> time layout.  Both exchanges in buffer CUR.
#ifndef FT_PK
template <int CUR>
__device__ __forceinline__ void ft4096_inv(float2 (&x)[16], const FtLane& f) {
    constexpr int T = (FT_NBUF == 2 ? CUR : 0) * FT_BUF;
    if (FT_NBUF == 2) FT_BARRIER();       // the buffer must be quiet before the row-private exchange below
    dft16<-1>(x);
#ifndef FT_EXP_NOX2
#pragma unroll
    for (int j = 0; j < 16; ++j) f.wr[T + j] = x[j];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < 16; ++m) x[m] = f.rdA[T + FT_PITCH * m];
    __builtin_amdgcn_wave_barrier();
#endif
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) x[k2] = mul_tw<-1>(x[k2], ft_tw2(f, k2));
    dft16<-1>(x);
#ifndef FT_EXP_NOX1
#pragma unroll
    for (int m = 0; m < 16; ++m) f.wr[T + m] = x[m];
    FT_BARRIER();
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) x[k1] = f.rdB[T + 16 * FT_PITCH * k1];
    if (FT_NBUF == 1) FT_BARRIER();
#endif
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) x[k1] = mul_tw<-1>(x[k1], f.tw1[16 * k1]);
    dft16<-1>(x);
}
#else
template <int CUR>
__device__ __forceinline__ void ft4096_inv(float2 (&xs)[16], const FtLane& f) {
    constexpr int T = (FT_NBUF == 2 ? CUR : 0) * FT_BUF;
    if (FT_NBUF == 2) FT_BARRIER();
    v2f x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = pk_from(xs[r]);
    pk_dft16<-1>(x);
#ifndef FT_EXP_NOX2
#pragma unroll
    for (int j = 0; j < 16; ++j) f.wr[T + j] = pk_to(x[j]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < 16; ++m) x[m] = pk_from(f.rdA[T + FT_PITCH * m]);
    __builtin_amdgcn_wave_barrier();
#endif
    pk_twiddle<-1, 0>(x, [&](int k2) { return pk_from(ft_tw2(f, k2)); });
    pk_dft16<-1>(x);
#ifndef FT_EXP_NOX1
#pragma unroll
    for (int m = 0; m < 16; ++m) f.wr[T + m] = pk_to(x[m]);
    FT_BARRIER();
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) x[k1] = pk_from(f.rdB[T + 16 * FT_PITCH * k1]);
    if (FT_NBUF == 1) FT_BARRIER();
#endif
    pk_twiddle<-1, 1>(x, [&](int k1) { return pk_from(f.tw1[16 * k1]); });
    pk_dft16<-1>(x);
#pragma unroll
    for (int r = 0; r < 16; ++r) xs[r] = pk_to(x[r]);
}
#endif

// Host side: FT_GTAB float2 (double-precision trig, rounded once); device copy per device.
void ft_make_tables(float2* host_tab);
int ft_device_tables(const float2** out);
