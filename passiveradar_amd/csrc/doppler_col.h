// Doppler stage of fast_xambg as ONE kernel: FFT over slow time + fftshift (range_doppler_processing.py:89)
// straight from the segment kernels' slow-time buffer to out[frame][f'][k].  The buffer is TILE-MAJOR (round 4),
// y[frame][k / KT][j][k % KT]: a workgroup's tile is one contiguous block of F KT samples, so every wavefront load is
// 512 contiguous bytes; with plain rows y[frame][j][k] (rounds 1-3) it was eight 64-byte pieces a row pitch apart, most
// of them straddling two 128-byte lines (cols is odd), and the 2048-bin kernel ran at 3.1 TB/s against 5.1 TB/s when
// both sides are contiguous (tools/caf_bench.py --shape dop2048x8).  The stores keep the (F, R+1) layout of the API.
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

// columns per workgroup at 1024 and 2048 bins (A/B builds override them: a narrower tile halves the LDS footprint and
// lets a CU hold two workgroups, at the price of shorter row segments)
#ifndef DOP_KT_1024
#define DOP_KT_1024 8
#endif
#ifndef DOP_KT_2048
#define DOP_KT_2048 8
#endif
template <int F>
struct DopCfg {
    static_assert(F == 256 || F == 512 || F == 1024 || F == 2048 || F == 4096, "column FFT sizes: 256..4096");
    static constexpr int Q = F / 16;                 // threads per column
    static constexpr int F3 = F / 256;               // radix of the last stage: 1, 2, 4, 8, 16
    static constexpr int E = 16 / F3;                // last-stage transforms per thread
    static constexpr int KT = F == 256 ? 32 : (F == 512 ? 16 : (F == 1024 ? DOP_KT_1024 : (F == 2048 ? DOP_KT_2048 : 4)));   // columns per workgroup
    static constexpr int THREADS = Q * KT;           // 512, 512, 512, 1024, 1024
    // A tile of more than half a CU's LDS would leave ONE workgroup per CU, its load, exchange and store phases
    // serialised; such tiles (2048 and 4096 bins) are exchanged in two rounds instead -- real parts, then imaginary parts,
    // through the same float slots -- so that two workgroups fit a CU and one's memory phase runs under the other's
    // exchanges (DOP_SPLIT_ABOVE: A/B builds)
#ifndef DOP_SPLIT_ABOVE
#define DOP_SPLIT_ABOVE (80 * 1024)
#endif
    static constexpr bool SPLIT = (F + F / 16) * KT * 8 > DOP_SPLIT_ABOVE;
    static constexpr int LDS_ELEMS = (F + F / 16) * KT;                       // float2 (or float, when SPLIT) slots
    static constexpr int LDS_BYTES = LDS_ELEMS * (SPLIT ? 4 : 8);
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

// Twiddles.  Thread p needs W^(p k1), k1 = 1..15, after S1 and W^(16 b ka), ka = 1..15, after S2: fifteen gathers from
// the table each, every one a dependent L1/L2 round trip when issued one by one (measured: that serialisation, not the
// arithmetic, set the kernel's time).  Instead four table entries per stage -- exponents 1, 2, 4, 8 times the base -- are
// loaded up front next to the data, and the other eleven are products of at most three of them (|error| <= 3 ulp).
struct DopTw {
    float2 s1[4];     // W_F^(p), W_F^(2p), W_F^(4p), W_F^(8p)
    float2 s2[4];     // W_F^(16b), W_F^(32b), W_F^(64b), W_F^(128b)
};
PRC_HD float2 dop_cmul(float2 a, float2 b) {
    return make_float2(fmaf(-a.y, b.y, a.x * b.x), fmaf(a.x, b.y, a.y * b.x));
}
template <int F>
PRC_HD DopTw dop_load_twiddles(const float2* __restrict__ tw, int p) {
    constexpr int F3 = DopCfg<F>::F3;
    const int b = p % F3;
    DopTw t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        t.s1[i] = tw[p << i];                    // 8 p < 8 Q = F/2
        t.s2[i] = tw[(16 * b) << i];             // 128 b < 128 F3 = F/2
    }
    return t;
}
// x[k] *= w^k, k = 1..15, from w^1, w^2, w^4, w^8
PRC_HD void dop_apply_powers(float2 (&x)[16], const float2 (&w)[4]) {
    const float2 w1 = w[0], w2 = w[1], w4 = w[2], w8 = w[3];
    const float2 w3 = dop_cmul(w1, w2), w5 = dop_cmul(w4, w1), w6 = dop_cmul(w4, w2);
    const float2 w7 = dop_cmul(w4, w3);
    x[1] = dop_cmul(x[1], w1);
    x[2] = dop_cmul(x[2], w2);
    x[3] = dop_cmul(x[3], w3);
    x[4] = dop_cmul(x[4], w4);
    x[5] = dop_cmul(x[5], w5);
    x[6] = dop_cmul(x[6], w6);
    x[7] = dop_cmul(x[7], w7);
    x[8] = dop_cmul(x[8], w8);
    x[9] = dop_cmul(x[9], dop_cmul(w8, w1));
    x[10] = dop_cmul(x[10], dop_cmul(w8, w2));
    x[11] = dop_cmul(x[11], dop_cmul(w8, w3));
    x[12] = dop_cmul(x[12], dop_cmul(w8, w4));
    x[13] = dop_cmul(x[13], dop_cmul(w8, w5));
    x[14] = dop_cmul(x[14], dop_cmul(w8, w6));
    x[15] = dop_cmul(x[15], dop_cmul(w8, w7));
}

// S1: x holds the 16 samples r Q + p of one column
template <int F>
PRC_HD void dop_stage1(float2 (&x)[16], const DopTw& t) {
    dft16<1>(x);
    dop_apply_powers(x, t.s1);
}
// LDS slots as ONE per-thread base plus a compile-time offset per register (so that a phase costs one address VGPR,
// not sixteen): with Q a multiple of 16 and F3 a divisor of 16,
//   X1 write  slot(k1 Q + p)          = [(p + p/16) KT + c]                    + k1 (17 Q / 16) KT
//   X1 read   slot(k1' Q + a F3 + b)  = [(k1' 17 Q / 16 + b) KT + c]           + (a F3 + (a F3)/16) KT      (= X2 write)
//   X2 read   slot(k1' Q + 16 g + m)  = [(k1' 17 Q / 16 + 17 g) KT + c]        + m KT
template <int F>
PRC_HD void dop_write1(const float2 (&x)[16], float2* lds, int p, int c) {
    constexpr int Q = DopCfg<F>::Q, KT = DopCfg<F>::KT;
    float2* w = lds + (p + (p >> 4)) * KT + c;
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) w[k1 * (Q + Q / 16) * KT] = x[k1];
}
template <int F>
PRC_HD int dop_base1(int p, int c) {
    constexpr int Q = DopCfg<F>::Q, KT = DopCfg<F>::KT, F3 = DopCfg<F>::F3;
    return ((p / F3) * (Q + Q / 16) + (p % F3)) * KT + c;
}
template <int F>
PRC_HD void dop_read1(float2 (&x)[16], const float2* lds, int p, int c) {
    constexpr int KT = DopCfg<F>::KT, F3 = DopCfg<F>::F3;
    const float2* r = lds + dop_base1<F>(p, c);
#pragma unroll
    for (int a = 0; a < 16; ++a) x[a] = r[(a * F3 + ((a * F3) >> 4)) * KT];
}
// the same four accesses on ONE component (C = 0: real, 1: imaginary) through float slots
template <int F, int C>
PRC_HD void dop_write1c(const float2 (&x)[16], float* lds, int p, int c) {
    constexpr int Q = DopCfg<F>::Q, KT = DopCfg<F>::KT;
    float* w = lds + (p + (p >> 4)) * KT + c;
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) w[k1 * (Q + Q / 16) * KT] = C ? x[k1].y : x[k1].x;
}
template <int F, int C>
PRC_HD void dop_read1c(float2 (&x)[16], const float* lds, int p, int c) {
    constexpr int KT = DopCfg<F>::KT, F3 = DopCfg<F>::F3;
    const float* r = lds + dop_base1<F>(p, c);
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        const float v = r[(a * F3 + ((a * F3) >> 4)) * KT];
        if (C) x[a].y = v; else x[a].x = v;
    }
}
template <int F, int C>
PRC_HD void dop_write2c(const float2 (&x)[16], float* lds, int p, int c) {
    constexpr int KT = DopCfg<F>::KT, F3 = DopCfg<F>::F3;
    float* w = lds + dop_base1<F>(p, c);
#pragma unroll
    for (int ka = 0; ka < 16; ++ka) w[(ka * F3 + ((ka * F3) >> 4)) * KT] = C ? x[ka].y : x[ka].x;
}
template <int F, int C>
PRC_HD void dop_read2c(float2 (&x)[16], const float* lds, int p, int c) {
    constexpr int Q = DopCfg<F>::Q, KT = DopCfg<F>::KT, F3 = DopCfg<F>::F3;
    const float* r = lds + ((p / F3) * (Q + Q / 16) + 17 * (p % F3)) * KT + c;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const float v = r[m * KT];
        if (C) x[m].y = v; else x[m].x = v;
    }
}

template <int F>
PRC_HD void dop_stage2(float2 (&x)[16], const DopTw& t) {
    dft16<1>(x);
    if (DopCfg<F>::F3 > 1) dop_apply_powers(x, t.s2);
}
template <int F>
PRC_HD void dop_write2(const float2 (&x)[16], float2* lds, int p, int c) {
    constexpr int KT = DopCfg<F>::KT, F3 = DopCfg<F>::F3;
    float2* w = lds + dop_base1<F>(p, c);
#pragma unroll
    for (int ka = 0; ka < 16; ++ka) w[(ka * F3 + ((ka * F3) >> 4)) * KT] = x[ka];
}
// register e F3 + b2 <- idx k1 Q + (g E + e) F3 + b2 = k1 Q + 16 g + m
template <int F>
PRC_HD void dop_read2(float2 (&x)[16], const float2* lds, int p, int c) {
    constexpr int Q = DopCfg<F>::Q, KT = DopCfg<F>::KT, F3 = DopCfg<F>::F3;
    const float2* r = lds + ((p / F3) * (Q + Q / 16) + 17 * (p % F3)) * KT + c;
#pragma unroll
    for (int m = 0; m < 16; ++m) x[m] = r[m * KT];
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

// dop_out_row(p, m) = [k1 + 16 g E] + dop_out_row_reg(m): the fftshift only flips the top bit of kb (of m when F3 = 1),
// which belongs to the register, so the thread's and the register's parts of the output row add without a carry
template <int F>
PRC_HD int dop_out_row_reg(int m) {
    constexpr int F3 = DopCfg<F>::F3;
    if (F3 == 1) return 16 * (m ^ 8);
    const int e = m / F3, kb = m % F3;
    return 16 * e + 256 * (kb ^ (F3 / 2));
}

// host side (caf_doppler.hip): W_F^m table, does the column kernel take this size, launch
void dop_make_table(float2* host_tab, int F);
bool dop_supported(int freq_bins, int cols);
int dop_tile_cols(int freq_bins);           // KT: columns per tile (the slow-time buffer is laid out in such tiles)
// y: tile-major slow-time buffer, y[frame][tile][j][KT], y_surface elements per frame; out: [frame][f'][cols]
int dop_launch(const float2* y, int64_t y_surface, float2* out, const float2* tw, int freq_bins, int cols, int nframes,
               hipStream_t stream);
// nch channels in ONE launch: channel i's surfaces at y + i * y_ch_stride, its maps to outs[i]
int dop_launch_multi(const float2* y, int64_t y_surface, int64_t y_ch_stride, float2* const* outs, int nch, const float2* tw,
                     int freq_bins, int cols, int nframes, hipStream_t stream);
