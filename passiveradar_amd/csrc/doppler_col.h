// Doppler stage of fast_xambg as ONE kernel: FFT over slow time + fftshift (range_doppler_processing.py:89)
// straight from the segment kernels' row-major slow-time buffer y[frame][j][k] to out[frame][f'][k].
//
// Round 1/2 ran transpose -> rocFFT (contiguous batch) -> shift+transpose: seven passes over the surface for two
// of algorithmic traffic.  Here a workgroup owns a tile of KT adjacent range columns and all F slow-time rows:
// every global access is a KT-wide row segment (coalesced both ways), the F-point transform of each column runs
// 16 x 16 x F3 (F3 = F/256) on 16 points per thread with two exchanges through one LDS tile, and the fftshift is
// the row index k ^ (F/2) of the store.  Bytes moved: read y once, write out once.
//
// Index algebra (per column; Q = F/16 threads p, registers r):   n = r Q + p,   k = k1 + 16 ka + 256 kb
//   S1  thread p              : x[r] = y[r Q + p];  A[k1] = DFT16_r;  A[k1] *= W_F^(p k1)
//   X1  write idx k1 Q + p    | barrier |  thread p' = k1' F3 + b reads idx k1' Q + a F3 + b, a = 0..15
//   S2  Bv[ka] = DFT16_a;  Bv[ka] *= W_F^(16 b ka)                      (= W_Q^(b ka))
//   X2  (F3 > 1) write idx k1' Q + ka F3 + b -- exactly the slots this thread read in X1, so no barrier between
//       the read and the write -- | barrier |  thread (k1', g = b) reads idx k1' Q + (g E + e) F3 + b2, E = 16/F3
//   S3  X[k1' + 16 (g E + e) + 256 kb] = DFT_F3 over b2, per e
// LDS element (idx, c) sits at (idx + idx/16) KT + c: the pad row per 16 idx spreads the g-strided reads of X2 over
// the banks; every other access is a run of consecutive idx.  The phases below are host + device so that
// tests/csrc/doppler_emul.cpp can run them thread by thread on the CPU against numpy (no GPU in the build container).
#pragma once
#include "fft_wave.h"

template <int F>
struct DopCfg {
    static_assert(F == 256 || F == 512 || F == 1024 || F == 2048 || F == 4096, "column FFT sizes: 256..4096");
    static constexpr int Q = F / 16;                 // threads per column
    static constexpr int F3 = F / 256;               // radix of the last stage: 1, 2, 4, 8, 16
    static constexpr int E = 16 / F3;                // last-stage transforms per thread
    static constexpr int KT = F == 256 ? 32 : (F <= 1024 ? 16 : (F == 2048 ? 8 : 4));   // columns per workgroup
    static constexpr int THREADS = Q * KT;           // 512, 512, 1024, 1024, 1024
    static constexpr int LDS_ELEMS = (F + F / 16) * KT;
};

PRC_HD int dop_slot(int idx, int kt, int c) { return (idx + (idx >> 4)) * kt + c; }

PRC_HD void dop_dft2(float2& a, float2& b) {
    const float2 s = f2add(a, b), d = f2sub(a, b);
    a = s;
    b = d;
}
template <int DIR>
PRC_HD void dop_dft8(float2* v) {               // natural order in and out
    constexpr float RH = 0.70710678118654752f;
    bfly4<DIR>(v[0], v[2], v[4], v[6]);          // E[0..3] in v[0], v[2], v[4], v[6]
    bfly4<DIR>(v[1], v[3], v[5], v[7]);          // O[0..3] in v[1], v[3], v[5], v[7]
    v[3] = mul_cs<DIR>(v[3], RH, RH);
    v[5] = mul_mi<DIR>(v[5]);
    v[7] = mul_cs<DIR>(v[7], -RH, RH);
    const float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    const float2 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    v[0] = f2add(e0, o0); v[4] = f2sub(e0, o0);
    v[1] = f2add(e1, o1); v[5] = f2sub(e1, o1);
    v[2] = f2add(e2, o2); v[6] = f2sub(e2, o2);
    v[3] = f2add(e3, o3); v[7] = f2sub(e3, o3);
}

// S1: x holds the 16 samples r Q + p of one column.  tw: W_F^m, m = 0..F-1
template <int F>
PRC_HD void dop_stage1(float2 (&x)[16], const float2* __restrict__ tw, int p) {
    dft16<1>(x);
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) x[k1] = mul_tw<1>(x[k1], tw[p * k1]);
}
template <int F>
PRC_HD void dop_write1(const float2 (&x)[16], float2* lds, int p, int c) {
    constexpr int Q = DopCfg<F>::Q, KT = DopCfg<F>::KT;
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) lds[dop_slot(k1 * Q + p, KT, c)] = x[k1];
}
template <int F>
PRC_HD void dop_read1(float2 (&x)[16], const float2* lds, int p, int c) {
    constexpr int Q = DopCfg<F>::Q, KT = DopCfg<F>::KT, F3 = DopCfg<F>::F3;
    const int k1 = p / F3, b = p % F3;
#pragma unroll
    for (int a = 0; a < 16; ++a) x[a] = lds[dop_slot(k1 * Q + a * F3 + b, KT, c)];
}
template <int F>
PRC_HD void dop_stage2(float2 (&x)[16], const float2* __restrict__ tw, int p) {
    constexpr int F3 = DopCfg<F>::F3;
    const int b = p % F3;
    dft16<1>(x);
    if (F3 > 1) {
#pragma unroll
        for (int ka = 1; ka < 16; ++ka) x[ka] = mul_tw<1>(x[ka], tw[16 * b * ka]);
    }
}
template <int F>
PRC_HD void dop_write2(const float2 (&x)[16], float2* lds, int p, int c) {
    constexpr int Q = DopCfg<F>::Q, KT = DopCfg<F>::KT, F3 = DopCfg<F>::F3;
    const int k1 = p / F3, b = p % F3;
#pragma unroll
    for (int ka = 0; ka < 16; ++ka) lds[dop_slot(k1 * Q + ka * F3 + b, KT, c)] = x[ka];
}
// register e F3 + b2 <- idx k1 Q + (g E + e) F3 + b2
template <int F>
PRC_HD void dop_read2(float2 (&x)[16], const float2* lds, int p, int c) {
    constexpr int Q = DopCfg<F>::Q, KT = DopCfg<F>::KT, F3 = DopCfg<F>::F3;
    const int k1 = p / F3, g = p % F3;
#pragma unroll
    for (int m = 0; m < 16; ++m) x[m] = lds[dop_slot(k1 * Q + g * 16 + m, KT, c)];
}
template <int F>
PRC_HD void dop_stage3(float2 (&x)[16]) {
    constexpr int F3 = DopCfg<F>::F3;
    if (F3 == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) dop_dft2(x[2 * e], x[2 * e + 1]);
    } else if (F3 == 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bfly4<1>(x[4 * e], x[4 * e + 1], x[4 * e + 2], x[4 * e + 3]);
    } else if (F3 == 8) {
        dop_dft8<1>(&x[0]);
        dop_dft8<1>(&x[8]);
    } else if (F3 == 16) {
        dft16<1>(x);
    }
}
// output row (after np.fft.fftshift along the Doppler axis) of register m of thread p
template <int F>
PRC_HD int dop_out_row(int p, int m) {
    constexpr int F3 = DopCfg<F>::F3, E = DopCfg<F>::E;
    const int k1 = p / F3, g = p % F3;
    const int e = m / F3, kb = m % F3;
    const int k = F3 == 1 ? k1 + 16 * m : k1 + 16 * (g * E + e) + 256 * kb;
    return k ^ (F / 2);
}

// host side (caf_doppler.hip): W_F^m table, does the column kernel take this size, launch
void dop_make_table(float2* host_tab, int F);
bool dop_supported(int freq_bins);
int dop_launch(const float2* y, float2* out, const float2* tw, int freq_bins, int cols, int nframes,
               hipStream_t stream);
