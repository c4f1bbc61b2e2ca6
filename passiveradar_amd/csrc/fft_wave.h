// One-wavefront 1024-point complex FFT held in registers (16 points per lane) for gfx950.
//
// Data flow (validated index-for-index by tools/fft_layout_model.py):
//   forward  : lane n2 holds x[64*n1 + n2] in register n1 (coalesced global loads)
//              S1 radix-16 DFT over n1 in registers            -> A[k1]
//              S2 twiddle W_1024^(n2*k1)                         (LDS table, conflict-free)
//              S3 transpose through this wave's LDS tile        lane l=(k1=l>>2, j=l&3), reg m: A[k1][j+4m]
//              S4 radix-16 DFT over m in registers             -> C[m']
//              S5 twiddle W_64^(j*m')
//              S6 radix-4 DFT over j across the 4 lanes of a quad (DPP quad_perm, no LDS)
//              result: lane 4*k1 + bitrev2(j'), register m'  <->  frequency k1 + 16 m' + 256 j'
//   inverse  : the same stages backwards with conjugated twiddles; natural order out
//              (lane n2, register n1 <-> sample 64*n1 + n2), UNNORMALISED (x P = 1024).
// Pointwise products (correlation, filtering) are taken in the permuted frequency layout, so
// no bit-reversal pass exists anywhere.  No workgroup barrier is involved: every wave owns a
// private 16 x 68 float2 LDS tile (row pitch 68 makes both the row-wise and the quad-strided
// accesses bank-conflict free for ds_read_b64 / ds_write_b64).
#pragma once
#include "common.h"

// pure arithmetic helpers are host + device: tests/csrc/doppler_emul.cpp runs the Doppler kernel's phases on the CPU
#define PRC_HD __host__ __device__ __forceinline__

#define FFTW_P 1024
#define FFTW_R 16                 // points per lane
#define FFTW_PITCH 68             // float2 elements per LDS tile row
#define FFTW_TILE (16 * FFTW_PITCH)   // float2 elements per wave tile
#define FFTW_TW1 (16 * 64)        // W_1024^(n2*k1), [k1][n2]
#define FFTW_TW2 (16 * 4)         // W_64^(j*m'),    [m'][j]
#define FFTW_TW2S (16 * 4)        // the same times the quad sign sigma_j = sA_j sB_j (forward transform)
#define FFTW_TABLE (FFTW_TW1 + FFTW_TW2 + FFTW_TW2S)

PRC_HD float2 f2add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
PRC_HD float2 f2sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// multiply by -i (DIR = +1, forward) or +i (DIR = -1, inverse)
template <int DIR>
PRC_HD float2 mul_mi(float2 a) {
    return DIR > 0 ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);
}
// a * (c - i*DIR*s)
template <int DIR>
PRC_HD float2 mul_cs(float2 a, float c, float s) {
    return DIR > 0 ? make_float2(fmaf(a.y, s, a.x * c), fmaf(-a.x, s, a.y * c))
                   : make_float2(fmaf(-a.y, s, a.x * c), fmaf(a.x, s, a.y * c));
}
// a * t (forward) or a * conj(t) (inverse)
template <int DIR>
PRC_HD float2 mul_tw(float2 a, float2 t) {
    return DIR > 0 ? make_float2(fmaf(-a.y, t.y, a.x * t.x), fmaf(a.x, t.y, a.y * t.x))
                   : make_float2(fmaf(a.y, t.y, a.x * t.x), fmaf(-a.x, t.y, a.y * t.x));
}

template <int DIR>
PRC_HD void bfly4(float2& x0, float2& x1, float2& x2, float2& x3) {
    const float2 s02 = f2add(x0, x2), d02 = f2sub(x0, x2);
    const float2 s13 = f2add(x1, x3), d13 = f2sub(x1, x3);
    const float2 t = mul_mi<DIR>(d13);
    x0 = f2add(s02, s13);
    x2 = f2sub(s02, s13);
    x1 = f2add(d02, t);
    x3 = f2sub(d02, t);
}

// radix-4 butterfly whose third / third and fourth inputs are known to be zero (zero-padded transforms)
template <int DIR>
PRC_HD void bfly4_z3(float2& x0, float2& x1, float2& x2, float2& x3) {     // x3 == 0
    const float2 s02 = f2add(x0, x2), d02 = f2sub(x0, x2);
    const float2 t = mul_mi<DIR>(x1);
    x0 = f2add(s02, x1);
    x2 = f2sub(s02, x1);
    x1 = f2add(d02, t);
    x3 = f2sub(d02, t);
}
template <int DIR>
PRC_HD void bfly4_z23(float2& x0, float2& x1, float2& x2, float2& x3) {    // x2 == x3 == 0
    const float2 a = x0, b = x1;
    const float2 t = mul_mi<DIR>(b);
    x0 = f2add(a, b);
    x2 = f2sub(a, b);
    x1 = f2add(a, t);
    x3 = f2sub(a, t);
}

// In-register 16-point DFT, natural order in and out: v[k] = sum_n v[n] W_16^(DIR*n*k).
// NZ: inputs v[NZ..15] are known to be zero (16, 12 or 8): the first radix-4 pass then skips their additions.
template <int DIR, int NZ = 16>
PRC_HD void dft16(float2 (&v)[16]) {
    constexpr float C1 = 0.92387953251128674f;   // cos(pi/8)
    constexpr float S1 = 0.38268343236508977f;   // sin(pi/8)
    constexpr float RH = 0.70710678118654752f;   // sqrt(1/2)
    static_assert(NZ == 16 || NZ == 12 || NZ == 8, "dft16: zero tail of 0, 4 or 8 inputs");
    // step 1: radix-4 over a (n = 4a + b): v[4c + b] = t_b[c]
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        if (NZ == 16) bfly4<DIR>(v[b], v[4 + b], v[8 + b], v[12 + b]);
        else if (NZ == 12) bfly4_z3<DIR>(v[b], v[4 + b], v[8 + b], v[12 + b]);
        else bfly4_z23<DIR>(v[b], v[4 + b], v[8 + b], v[12 + b]);
    }
    // step 2: t_b[c] *= W_16^(b*c)
    v[5] = mul_cs<DIR>(v[5], C1, S1);      // b=1,c=1: e=1
    v[9] = mul_cs<DIR>(v[9], RH, RH);      // b=1,c=2: e=2
    v[13] = mul_cs<DIR>(v[13], S1, C1);    // b=1,c=3: e=3
    v[6] = mul_cs<DIR>(v[6], RH, RH);      // b=2,c=1: e=2
    v[10] = mul_mi<DIR>(v[10]);            // b=2,c=2: e=4
    v[14] = mul_cs<DIR>(v[14], -RH, RH);   // b=2,c=3: e=6
    v[7] = mul_cs<DIR>(v[7], S1, C1);      // b=3,c=1: e=3
    v[11] = mul_cs<DIR>(v[11], -RH, RH);   // b=3,c=2: e=6
    v[15] = mul_cs<DIR>(v[15], -C1, -S1);  // b=3,c=3: e=9
    // step 3: radix-4 over b for each c: v[4c + d] = X[c + 4d]
#pragma unroll
    for (int c = 0; c < 4; ++c) bfly4<DIR>(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
    // 4x4 transpose (register renaming): out[c + 4d] = v[4c + d]
    float2 t;
    t = v[1]; v[1] = v[4]; v[4] = t;
    t = v[2]; v[2] = v[8]; v[8] = t;
    t = v[3]; v[3] = v[12]; v[12] = t;
    t = v[6]; v[6] = v[9]; v[9] = t;
    t = v[7]; v[7] = v[13]; v[13] = t;
    t = v[11]; v[11] = v[14]; v[14] = t;
}

__device__ __forceinline__ float dpp_quad_xor2(float x) {   // quad_perm [2,3,0,1]
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_quad_xor1(float x) {   // quad_perm [1,0,3,2]
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true));
}

struct FftLane {
    float sA;   // +1 for quad lanes 0,1; -1 for lanes 2,3   (stage over lane^2)
    float sB;   // +1 for even quad lanes; -1 for odd        (stage over lane^1)
    float sg;   // sA * sB
    bool l3;    // quad lane 3 carries the -i twiddle
    int rd_off; // (lane>>2)*PITCH + (lane&3)  : quad-strided tile address
    int lane;
};

__device__ __forceinline__ FftLane fft_lane_setup() {
    FftLane f;
    f.lane = threadIdx.x & 63;
    const int j = f.lane & 3;
    f.sA = (j < 2) ? 1.f : -1.f;
    f.sB = (j & 1) ? -1.f : 1.f;
    f.sg = f.sA * f.sB;
    f.l3 = (j == 3);
    f.rd_off = (f.lane >> 2) * FFTW_PITCH + j;
    return f;
}

// Fill the twiddle tables (FFTW_TABLE float2) from the host-made global copy; call by the whole
// workgroup, then __syncthreads().
__device__ __forceinline__ void fft_load_tables(float2* lds_tab, const float2* __restrict__ gtab) {
    for (int i = threadIdx.x; i < FFTW_TABLE; i += blockDim.x) lds_tab[i] = gtab[i];
}

// Radix-4 across the four lanes of a quad, two butterflies r = s c + c_partner with a per-lane sign s.
// Written on the sign-scaled value U = s' c the butterfly is ONE VOP2-DPP instruction,
//     d += dpp(d) * (-s)        (v_fmac_f32_dpp: the partner's value arrives through the DPP operand),
// because the partner's sign is the opposite of one's own.  The scaling is free: it is folded into the twiddle
// that precedes the stage (forward, table TW2S) or into the caller's pointwise product (inverse, PRESCALED).
//   forward : U = sg c;  R = U - sA dpp_xor2(U) = sB (sA c + c_p);  lane 3: R *= -i;  X = R - sB dpp_xor1(R)
//   inverse : P = sg y;  R = P - sB dpp_xor1(P) = sA (sB y + y_p);  lane 3: R *= +i;  x = R - sA dpp_xor2(R)
// Eight elements (16 VGPRs) per asm block.  A DPP operand read needs two wait states after the VALU write of
// that register (the twiddle multiply / the lane-3 select just before); the compiler cannot see that these
// instructions are DPP, so each block opens with s_nop 1.
#define FFTW_DPP_LINE(n, perm) "v_fmac_f32_dpp %" #n ", %" #n ", -%16 quad_perm:" perm " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define FFTW_DPP_BLOCK(perm)                                                                               \
    "s_nop 1\n" FFTW_DPP_LINE(0, perm) FFTW_DPP_LINE(1, perm) FFTW_DPP_LINE(2, perm) FFTW_DPP_LINE(3, perm)  \
    FFTW_DPP_LINE(4, perm) FFTW_DPP_LINE(5, perm) FFTW_DPP_LINE(6, perm) FFTW_DPP_LINE(7, perm)             \
    FFTW_DPP_LINE(8, perm) FFTW_DPP_LINE(9, perm) FFTW_DPP_LINE(10, perm) FFTW_DPP_LINE(11, perm)           \
    FFTW_DPP_LINE(12, perm) FFTW_DPP_LINE(13, perm) FFTW_DPP_LINE(14, perm) FFTW_DPP_LINE(15, perm)
#define FFTW_DPP_OPERANDS(x, o)                                                                                      \
    "+v"(x[o].x), "+v"(x[o].y), "+v"(x[o + 1].x), "+v"(x[o + 1].y), "+v"(x[o + 2].x), "+v"(x[o + 2].y),             \
    "+v"(x[o + 3].x), "+v"(x[o + 3].y), "+v"(x[o + 4].x), "+v"(x[o + 4].y), "+v"(x[o + 5].x), "+v"(x[o + 5].y),     \
    "+v"(x[o + 6].x), "+v"(x[o + 6].y), "+v"(x[o + 7].x), "+v"(x[o + 7].y)
// x[m] -= s * x[m][lane ^ 2] (XOR2) or [lane ^ 1] for all 16 registers
template <bool XOR2>
__device__ __forceinline__ void quad_bfly(float2 (&x)[16], float s) {
    if (XOR2) {
        asm(FFTW_DPP_BLOCK("[2,3,0,1]") : FFTW_DPP_OPERANDS(x, 0) : "v"(s));
        asm(FFTW_DPP_BLOCK("[2,3,0,1]") : FFTW_DPP_OPERANDS(x, 8) : "v"(s));
    } else {
        asm(FFTW_DPP_BLOCK("[1,0,3,2]") : FFTW_DPP_OPERANDS(x, 0) : "v"(s));
        asm(FFTW_DPP_BLOCK("[1,0,3,2]") : FFTW_DPP_OPERANDS(x, 8) : "v"(s));
    }
}
template <int DIR>
__device__ __forceinline__ void quad_dft4_fwd(float2 (&u)[16], const FftLane& f) {   // u = sg * c
    quad_bfly<true>(u, f.sA);
#pragma unroll
    for (int m = 0; m < 16; ++m)
        if (f.l3) u[m] = mul_mi<DIR>(u[m]);
    quad_bfly<false>(u, f.sB);
}
// exact mirror (unnormalised x4): lane j holds sg * X[bitrev2(j)] in, time-side quad index j out
template <int DIR>   // DIR is the direction of the *transform* (-1 for the inverse FFT)
__device__ __forceinline__ void quad_dft4_bwd(float2 (&p)[16], const FftLane& f) {   // p = sg * y
    quad_bfly<false>(p, f.sB);
#pragma unroll
    for (int m = 0; m < 16; ++m)
        if (f.l3) p[m] = mul_mi<DIR>(p[m]);
    quad_bfly<true>(p, f.sA);
}

// Forward FFT: natural (lane n2, reg n1) -> permuted frequency layout.
// NZ: registers x[NZ..15] are zero in every lane (a zero-padded piece of at most 64 NZ samples)
#ifndef FT_PK
template <int NZ = 16>
__device__ __forceinline__ void fft1024_fwd(float2 (&x)[16], float2* tile, const float2* tab,
                                            const FftLane& f) {
    dft16<1, NZ>(x);
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) x[k1] = mul_tw<1>(x[k1], tab[k1 * 64 + f.lane]);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) tile[k1 * FFTW_PITCH + f.lane] = x[k1];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < 16; ++m) x[m] = tile[f.rd_off + 4 * m];
    __builtin_amdgcn_wave_barrier();
    dft16<1>(x);
    const float2* tw2s = tab + FFTW_TW1 + FFTW_TW2 + (f.lane & 3);   // sigma_j W_64^(j m')
    x[0].x *= f.sg;
    x[0].y *= f.sg;
#pragma unroll
    for (int m = 1; m < 16; ++m) x[m] = mul_tw<1>(x[m], tw2s[4 * m]);
    quad_dft4_fwd<1>(x, f);
}

// Inverse FFT (unnormalised): permuted frequency layout -> natural (lane n2, reg n1).
// PRESCALED: the caller already multiplied the spectrum by the lane's quad sign f.sg (e.g. folded into a filter).
template <bool PRESCALED = false>
__device__ __forceinline__ void fft1024_inv(float2 (&x)[16], float2* tile, const float2* tab,
                                            const FftLane& f) {
    if (!PRESCALED) {
#pragma unroll
        for (int m = 0; m < 16; ++m) { x[m].x *= f.sg; x[m].y *= f.sg; }
    }
    quad_dft4_bwd<-1>(x, f);
    const float2* tw2 = tab + FFTW_TW1 + (f.lane & 3);
#pragma unroll
    for (int m = 1; m < 16; ++m) x[m] = mul_tw<-1>(x[m], tw2[4 * m]);
    dft16<-1>(x);
#pragma unroll
    for (int m = 0; m < 16; ++m) tile[f.rd_off + 4 * m] = x[m];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) x[k1] = tile[k1 * FFTW_PITCH + f.lane];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) x[k1] = mul_tw<-1>(x[k1], tab[k1 * 64 + f.lane]);
    dft16<-1>(x);
}
#endif   // the FT_PK forms of the two transforms: fft_wave_pk.h

// w += conj(u) * v   (per register, same permuted layout on both sides)
#ifndef FT_PK
__device__ __forceinline__ void cmac_conj_a(float2& w, float2 u, float2 v) {
    w.x = fmaf(u.x, v.x, w.x);
    w.x = fmaf(u.y, v.y, w.x);
    w.y = fmaf(u.x, v.y, w.y);
    w.y = fmaf(-u.y, v.x, w.y);
}
// a * s, s real
__device__ __forceinline__ float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }
#endif

// Host side: build the FFTW_TABLE float2 twiddle table (double-precision trig, rounded once).
void fftw_make_tables(float2* host_tab);

#ifdef FT_PK
#include "fft_wave_pk.h"
#endif
