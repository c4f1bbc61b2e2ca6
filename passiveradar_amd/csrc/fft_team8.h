// 4096-point complex FFT by a team of EIGHT wavefronts (512 threads x 8 points) for gfx950.
//
// Why (round 6): the 16-point-per-thread transform of fft_team.h needs ~158 VGPRs with a CAF kernel's
// state around it -- three wavefronts per SIMD, i.e. three transforms in flight per CU.  Eight points per thread halve
// every array the kernel holds (u, v, accumulator: 48 VGPRs instead of 96) and keep a transform's butterflies at radix 8:
// the same three teams per CU are then SIX wavefronts per SIMD, twice as many instruction streams to hide the exchanges
// and the loads of one another.  Price: four radix-8 passes instead of three radix-16 ones -- a third exchange through
// LDS and one more twiddle pass (42 against 44 real operations per point: the arithmetic is the same).
//
// Data flow (index for index in tools/fft4096x8_model.py, which also checks every LDS phase for bank conflicts):
// t = thread 0..511, w = t >> 6 (wavefront), l = t & 63 (lane), r = register 0..7,
// n = 512 n1 + 64 n2 + 8 n3 + n4,  k = k1 + 8 k2 + 64 k3 + 512 k4:
//   time layout      : thread t, register r                 <->  sample 512 r + t          (coalesced global access)
//   frequency layout : wave k1, lane 8 k2 + k3, register k4 <->  bin k1 + 8 k2 + 64 k3 + 512 k4   (pointwise products only)
//   forward : S1 DFT8 over n1 (registers)          thread (n2, 8 n3 + n4) = (w, l)
//             T1a twiddle W_4096^(l k1)             LDS table [k1][l]
//             X1 exchange ACROSS the waves          barrier | write E[k1][n2 = w][l] | barrier | wave k1 reads E[k1][n2][l]
//             T1b twiddle W_64^(n2 k1)              k1 = this wave: eight wave-uniform constants (SGPR pairs)
//             S2 DFT8 over n2, T2 W_512^(l k2)      LDS table [k2][l]
//             X2 exchange inside the wave           write tile[72 k2 + l] | lane (k2, n4) reads tile[72 k2 + 8 n3 + n4]
//             S3 DFT8 over n3, T3 W_64^(n4 k3)      LDS table [k3][n4]
//             X3 exchange inside the wave           write tile[72 k2 + 9 k3 + n4] | lane (k2, k3) reads tile[72 k2 + 9 k3 + n4']
//             S4 DFT8 over n4                       register k4
//   inverse : the same stages backwards with conjugated twiddles, unnormalised (x 4096).
// One exchange buffer of eight wave regions of 576 float2 (36 864 B): a wave READS only its own region (X1) and does its
// two private exchanges there; only the X1 WRITE crosses regions, so it stands between two workgroup barriers -- two per
// transform, as the one-buffer schedule of fft_team.h.  Every access is one per-thread base plus a compile-time offset;
// all of them are bank-conflict free (ds_write_b64: 16 lanes over 32 banks, ds_read_b64: 32 lanes over 64 banks).
#pragma once
#include "fft_wave.h"      // first: with FT_PK it pulls fft_pk.h in the order the packed 1024-point forms need
#include "fft_pk.h"

#define F8_P 4096
#define F8_THREADS 512
#define F8_PITCH 72
#define F8_REGION (8 * F8_PITCH)           // float2 per wave region
#define F8_XBUF (8 * F8_REGION)            // float2 of the exchange buffer
#define F8_TW1A (8 * 64)                   // W_4096^(l k1), [k1][l]
#define F8_TW2 (8 * 64)                    // W_512^(l k2),  [k2][l]
#define F8_TW3 (8 * 8)                     // W_64^(n4 k3),  [k3][n4]
#define F8_LDS_ELEMS (F8_XBUF + F8_TW1A + F8_TW2 + F8_TW3)     // 45 568 B: three teams per CU

struct F8Lane {
    int t;
    float2* x1w;         // xbuf + 64 w + l                      : X1 write of register k1 at [F8_REGION k1]
    float2* own;         // xbuf + F8_REGION w + l               : X1 read of n2 at [64 n2]; X2 write of k2 at [72 k2]
    float2* quad;        // xbuf + F8_REGION w + 72 (l >> 3) + (l & 7) : X2 read of n3 at [8 n3]; X3 write of k3 at [9 k3]
    float2* oct;         // xbuf + F8_REGION w + 72 (l >> 3) + 9 (l & 7) : X3 read of n4 at [n4]
    const float2* tw;    // tables + l                           : T1a at [64 k1], T2 at [F8_TW1A + 64 k2]
    const float2* tw3;   // tables + F8_TW1A + F8_TW2 + (l & 7)  : T3 at [8 k3]
    v2f c[8];            // W_64^(n2 w), wave-uniform
};

// Fill the twiddle tables in LDS from the W_4096^m table (gtab4096: 4096 float2, double-precision trig rounded once:
// fft_team.h's host table past its first FT_TW1 entries) and this wave's T1b constants; ends with __syncthreads().
// lds: F8_LDS_ELEMS float2.
__device__ __forceinline__ F8Lane f8_setup(float2* lds, const float2* __restrict__ gtab4096) {
    F8Lane f;
    f.t = threadIdx.x;
    const int w = f.t >> 6, l = f.t & 63;
    float2* tab = lds + F8_XBUF;
    f.x1w = lds + 64 * w + l;
    f.own = lds + F8_REGION * w + l;
    f.quad = lds + F8_REGION * w + F8_PITCH * (l >> 3) + (l & 7);
    f.oct = lds + F8_REGION * w + F8_PITCH * (l >> 3) + 9 * (l & 7);
    f.tw = tab + l;
    f.tw3 = tab + F8_TW1A + F8_TW2 + (l & 7);
    // thread t fills entry t of the two 512-entry tables ([k][l] with k = t >> 6, l = t & 63) and, the first 64 threads, of T3
    tab[f.t] = gtab4096[(l * w) & (F8_P - 1)];
    tab[F8_TW1A + f.t] = gtab4096[(8 * l * w) & (F8_P - 1)];
    if (f.t < F8_TW3) tab[F8_TW1A + F8_TW2 + f.t] = gtab4096[(64 * (f.t & 7) * (f.t >> 3)) & (F8_P - 1)];
    const int wu = __builtin_amdgcn_readfirstlane(w);
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) f.c[n2] = pk_from(gtab4096[(64 * n2 * wu) & (F8_P - 1)]);
    __syncthreads();
    return f;
}

// In-register 8-point DFT on packed instructions, natural order in and out: v[k] = sum_n v[n] W_8^(DIR n k).
// NZ: inputs v[NZ..7] are known to be zero (8, 6 or 4).
template <int DIR, int NZ = 8>
__device__ __forceinline__ void pk_dft8(v2f (&v)[8]) {
    constexpr float RH = 0.70710678118654752f;
    static_assert(NZ == 8 || NZ == 6 || NZ == 4, "pk_dft8: zero tail of 0, 2 or 4 inputs");
    // radix-2 over n = j + 4 a: a_j = v[j] + v[j+4] (even outputs), d_j = (v[j] - v[j+4]) W_8^j (odd outputs)
    v2f a0, a1, a2, a3, d0, d1, d2, d3;
    if (NZ == 4) {
        a0 = d0 = v[0]; a1 = d1 = v[1]; a2 = d2 = v[2]; a3 = d3 = v[3];
    } else {
        a0 = pk_add(v[0], v[4]); d0 = pk_sub(v[0], v[4]);
        a1 = pk_add(v[1], v[5]); d1 = pk_sub(v[1], v[5]);
        if (NZ == 6) { a2 = d2 = v[2]; a3 = d3 = v[3]; }
        else { a2 = pk_add(v[2], v[6]); d2 = pk_sub(v[2], v[6]); a3 = pk_add(v[3], v[7]); d3 = pk_sub(v[3], v[7]); }
    }
    // W_8^1 = (1 -+ i) sqrt(1/2): d (1 -+ i) is one packed add of d with its own rotation; W_8^3 = -(1 +- i) sqrt(1/2);
    // W_8^2 = -+i is folded into the additions of the odd half's radix-4 (pk_bfly4_rot2)
    d1 = pk_scale(pk_add_rot<DIR>(d1, d1), RH);
    d3 = pk_scale(pk_sub_rot<DIR>(d3, d3), -RH);
    pk_bfly4<DIR>(a0, a1, a2, a3);
    pk_bfly4_rot2<DIR>(d0, d1, d2, d3);
    v[0] = a0; v[2] = a1; v[4] = a2; v[6] = a3;
    v[1] = d0; v[3] = d1; v[5] = d2; v[7] = d3;
}

// x[k] *= tw(k) (DIR > 0) or conj(tw(k)), k = 1..7, four at a time: the multiply halves of a group first, its fma halves
// after (no dependent neighbours; eight temporaries, not sixteen -- the kernel around this lives at 80 VGPRs)
template <int DIR, typename TW>
__device__ __forceinline__ void pk_twiddle8(v2f (&x)[8], TW tw) {
#pragma unroll
    for (int g = 0; g < 8; g += 4) {
        v2f t[4], p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (g + i >= 1) { t[i] = tw(g + i); p[i] = pk_cmul_p(x[g + i], t[i]); }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (g + i >= 1) x[g + i] = pk_cmul_q<(DIR < 0)>(x[g + i], t[i], p[i]);
    }
}
// the same with the wave-uniform constants of T1b (SGPR pairs)
template <int DIR>
__device__ __forceinline__ void pk_twiddle8_s(v2f (&x)[8], const v2f (&c)[8]) {
#pragma unroll
    for (int g = 0; g < 8; g += 4) {
        v2f p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (g + i >= 1) p[i] = pk_cmul_p_s(x[g + i], c[g + i]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (g + i >= 1) x[g + i] = pk_cmul_q_s<(DIR < 0)>(x[g + i], c[g + i], p[i]);
    }
}

#ifdef F8_EXP_NOBARRIER          // timing ablation only (wrong results)
#define F8_BARRIER() ((void)0)
#else
#define F8_BARRIER() __syncthreads()
#endif

// Forward FFT: time layout -> frequency layout.  NZ: registers x[NZ..7] are zero in every thread (a zero-padded piece of
// at most 512 NZ samples).
template <int NZ = 8>
__device__ __forceinline__ void f8_fwd(float2 (&xs)[8], const F8Lane& f) {
    v2f x[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) x[r] = pk_from(xs[r]);
    pk_dft8<1, NZ>(x);
    pk_twiddle8<1>(x, [&](int k1) { return pk_from(f.tw[64 * k1]); });
    F8_BARRIER();                         // every wave is done with the previous transform's data in every region
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) f.x1w[F8_REGION * k1] = pk_to(x[k1]);
    F8_BARRIER();
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) x[n2] = pk_from(f.own[64 * n2]);
    pk_twiddle8_s<1>(x, f.c);
    pk_dft8<1>(x);
    pk_twiddle8<1>(x, [&](int k2) { return pk_from(f.tw[F8_TW1A + 64 * k2]); });
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) f.own[F8_PITCH * k2] = pk_to(x[k2]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n3 = 0; n3 < 8; ++n3) x[n3] = pk_from(f.quad[8 * n3]);
    pk_dft8<1>(x);
    pk_twiddle8<1>(x, [&](int k3) { return pk_from(f.tw3[8 * k3]); });
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3) f.quad[9 * k3] = pk_to(x[k3]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n4 = 0; n4 < 8; ++n4) x[n4] = pk_from(f.oct[n4]);
    __builtin_amdgcn_wave_barrier();
    pk_dft8<1>(x);
#pragma unroll
    for (int r = 0; r < 8; ++r) xs[r] = pk_to(x[r]);
}

// Inverse FFT (unnormalised): frequency layout -> time layout.
__device__ __forceinline__ void f8_inv(float2 (&xs)[8], const F8Lane& f) {
    v2f x[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) x[r] = pk_from(xs[r]);
    pk_dft8<-1>(x);                       // over k4 -> n4; lane (k2, k3)
    F8_BARRIER();                         // a slower wave may still read this region (X1 of an inverse just before)
#pragma unroll
    for (int n4 = 0; n4 < 8; ++n4) f.oct[n4] = pk_to(x[n4]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3) x[k3] = pk_from(f.quad[9 * k3]);
    pk_twiddle8<-1>(x, [&](int k3) { return pk_from(f.tw3[8 * k3]); });
    pk_dft8<-1>(x);                       // over k3 -> n3; lane (k2, n4)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n3 = 0; n3 < 8; ++n3) f.quad[8 * n3] = pk_to(x[n3]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) x[k2] = pk_from(f.own[F8_PITCH * k2]);
    pk_twiddle8<-1>(x, [&](int k2) { return pk_from(f.tw[F8_TW1A + 64 * k2]); });
    pk_dft8<-1>(x);                       // over k2 -> n2; thread (k1 = w, l)
    pk_twiddle8_s<-1>(x, f.c);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) f.own[64 * n2] = pk_to(x[n2]);
    F8_BARRIER();
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) x[k1] = pk_from(f.x1w[F8_REGION * k1]);
    pk_twiddle8<-1>(x, [&](int k1) { return pk_from(f.tw[64 * k1]); });
    pk_dft8<-1>(x);                       // over k1 -> n1
#pragma unroll
    for (int r = 0; r < 8; ++r) xs[r] = pk_to(x[r]);
}
