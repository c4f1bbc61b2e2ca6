// Complex arithmetic on VOP3P packed-f32 instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) for gfx950.
//
// A complex64 value lives in an even-aligned VGPR pair (re = low, im = high), which is how dwordx2 loads and
// ds_read_b64 deliver it.  One packed instruction works on both halves, and the op_sel / neg modifiers choose, per
// half, WHICH half of each source it reads and with which sign -- so that
//   a + b, a - b                         one instruction
//   a -+ i b  (the radix-4 rotations)    one instruction (the swap and the sign are operand modifiers)
//   a * t, a * conj(t)                   two (mul + fma), against four scalar ones
//   w += conj(u) v                       two fma, against four
// The radix-16 butterfly below is 80 packed instructions against 160 scalar ones for the same roundings in the
// same order (every result is bit-identical to the scalar dft16 of fft_wave.h; tools/ubench/dft16pk.hip checks that
// on the device).  SLP vectorisation of the scalar code does not get there: it makes v_pk_add/mul only for the
// unswizzled cases and pays v_mov pairs for the rest (round 1 measured that as "no gain, more registers").
#pragma once
#include "fft_wave.h"

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f pk_from(float2 a) { return v2f{a.x, a.y}; }
__device__ __forceinline__ float2 pk_to(v2f a) { return make_float2(a.x, a.y); }

// Plain sums are vector expressions (the compiler makes v_pk_add_f32, with neg modifiers for the difference).  Every
// form with a swizzle is ONE inline instruction with explicit op_sel / neg modifiers: the instruction selector folds
// `v2f{b.y, -b.x}` into modifiers only when b is assembled from scalars, and otherwise emits v_xor + v_mov per use
// (measured on the listing of pk_dft16: 212 instructions instead of 126).  A v_pk_* that reads the result of the VALU
// instruction right before it needs one wait state on gfx950 (the compiler adds the s_nop 0, for its own packed
// instructions as well); the multiply halves of a batch of products are therefore written first, the fma halves after.
__device__ __forceinline__ v2f pk_add(v2f a, v2f b) { return a + b; }
__device__ __forceinline__ v2f pk_sub(v2f a, v2f b) { return a - b; }
#define PK_ADD_FORM(name, mods)                                                      \
    __device__ __forceinline__ v2f name(v2f a, v2f b) {                              \
        v2f d;                                                                       \
        asm("v_pk_add_f32 %0, %1, %2" mods : "=v"(d) : "v"(a), "v"(b));              \
        return d;                                                                    \
    }
PK_ADD_FORM(pk_add_mi, " op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")         // a - i b = (a.x + b.y, a.y - b.x)
PK_ADD_FORM(pk_add_pi, " op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")         // a + i b = (a.x - b.y, a.y + b.x)
#undef PK_ADD_FORM

// a + (-i)^DIR-sense rotation of b: DIR > 0 (forward) a - i b, DIR < 0 (inverse) a + i b; and the opposite
template <int DIR>
__device__ __forceinline__ v2f pk_add_rot(v2f a, v2f b) { return DIR > 0 ? pk_add_mi(a, b) : pk_add_pi(a, b); }
template <int DIR>
__device__ __forceinline__ v2f pk_sub_rot(v2f a, v2f b) { return DIR > 0 ? pk_add_pi(a, b) : pk_add_mi(a, b); }

// a * t in two halves: p = (a.x t.x, a.y t.x), then (p.x - a.y t.y, p.y + a.x t.y) -- the roundings of mul_tw<1>;
// CONJ: a * conj(t) = (p.x + a.y t.y, p.y - a.x t.y) -- mul_tw<-1>.  T is "v" (a VGPR pair) or "s" (an SGPR pair)
#define PK_CMUL_FORMS(sfx, T)                                                                                         \
    __device__ __forceinline__ v2f pk_cmul_p##sfx(v2f a, v2f t) {                                                     \
        v2f p;                                                                                                        \
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(p) : "v"(a), T(t));                         \
        return p;                                                                                                     \
    }                                                                                                                 \
    template <bool CONJ>                                                                                              \
    __device__ __forceinline__ v2f pk_cmul_q##sfx(v2f a, v2f t, v2f p) {                                              \
        v2f d;                                                                                                        \
        if (CONJ)                                                                                                     \
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), T(t), "v"(p)); \
        else                                                                                                          \
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(a), T(t), "v"(p)); \
        return d;                                                                                                     \
    }
PK_CMUL_FORMS(, "v")
PK_CMUL_FORMS(_s, "s")
#undef PK_CMUL_FORMS
__device__ __forceinline__ v2f pk_cmul(v2f a, v2f t) { return pk_cmul_q<false>(a, t, pk_cmul_p(a, t)); }
__device__ __forceinline__ v2f pk_cmul_conj(v2f a, v2f t) { return pk_cmul_q<true>(a, t, pk_cmul_p(a, t)); }
template <int DIR>
__device__ __forceinline__ v2f pk_mul_tw(v2f a, v2f t) { return DIR > 0 ? pk_cmul(a, t) : pk_cmul_conj(a, t); }

// x[k] *= tw(k) (DIR > 0) or conj(tw(k)) for k = K0..15, four at a time: the multiply halves of a group, then its fma
// halves (no dependent neighbours, eight temporaries)
template <int DIR, int K0, typename TW>
__device__ __forceinline__ void pk_twiddle(v2f (&x)[16], TW tw) {
#pragma unroll
    for (int g = K0 & ~3; g < 16; g += 4) {
        v2f t[4], p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (g + i >= K0) { t[i] = tw(g + i); p[i] = pk_cmul_p(x[g + i], t[i]); }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (g + i >= K0) x[g + i] = pk_cmul_q<(DIR < 0)>(x[g + i], t[i], p[i]);
    }
}

// a * s (real scale, both halves)
__device__ __forceinline__ v2f pk_scale(v2f a, float s) { return a * v2f{s, s}; }

// w += conj(u) v: (u.x v.x + u.y v.y, u.x v.y - u.y v.x), the roundings (and their order) of cmac_conj_a
__device__ __forceinline__ void pk_cmac_conj_a(v2f& w, v2f u, v2f v) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(w) : "v"(u), "v"(v));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "+v"(w) : "v"(u), "v"(v));
}
// w += u conj(x): (u.x x.x + u.y x.y, u.y x.x - u.x x.y), the roundings of ltc_cmac_bconj
__device__ __forceinline__ void pk_cmac_bconj(v2f& w, v2f u, v2f x) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(w) : "v"(u), "v"(x));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "+v"(w) : "v"(u), "v"(x));
}

template <int DIR>
__device__ __forceinline__ void pk_bfly4(v2f& x0, v2f& x1, v2f& x2, v2f& x3) {
    const v2f s02 = pk_add(x0, x2), d02 = pk_sub(x0, x2);
    const v2f s13 = pk_add(x1, x3), d13 = pk_sub(x1, x3);
    x0 = pk_add(s02, s13);
    x2 = pk_sub(s02, s13);
    x1 = pk_add_rot<DIR>(d02, d13);
    x3 = pk_sub_rot<DIR>(d02, d13);
}
// the same with x2 standing for (-+i) x2 (the W_16^4 factor of dft16's middle step, folded into the additions)
template <int DIR>
__device__ __forceinline__ void pk_bfly4_rot2(v2f& x0, v2f& x1, v2f& x2, v2f& x3) {
    const v2f s02 = pk_add_rot<DIR>(x0, x2), d02 = pk_sub_rot<DIR>(x0, x2);
    const v2f s13 = pk_add(x1, x3), d13 = pk_sub(x1, x3);
    x0 = pk_add(s02, s13);
    x2 = pk_sub(s02, s13);
    x1 = pk_add_rot<DIR>(d02, d13);
    x3 = pk_sub_rot<DIR>(d02, d13);
}
template <int DIR>
__device__ __forceinline__ void pk_bfly4_z3(v2f& x0, v2f& x1, v2f& x2, v2f& x3) {      // x3 == 0
    const v2f s02 = pk_add(x0, x2), d02 = pk_sub(x0, x2);
    const v2f b = x1;
    x0 = pk_add(s02, b);
    x2 = pk_sub(s02, b);
    x1 = pk_add_rot<DIR>(d02, b);
    x3 = pk_sub_rot<DIR>(d02, b);
}
template <int DIR>
__device__ __forceinline__ void pk_bfly4_z23(v2f& x0, v2f& x1, v2f& x2, v2f& x3) {     // x2 == x3 == 0
    const v2f a = x0, b = x1;
    x0 = pk_add(a, b);
    x2 = pk_sub(a, b);
    x1 = pk_add_rot<DIR>(a, b);
    x3 = pk_sub_rot<DIR>(a, b);
}

// dft16 of fft_wave.h on packed instructions: same decomposition, same roundings, natural order in and out
template <int DIR, int NZ = 16>
__device__ __forceinline__ void pk_dft16(v2f (&v)[16]) {
    constexpr float C1 = 0.92387953251128674f;
    constexpr float S1 = 0.38268343236508977f;
    constexpr float RH = 0.70710678118654752f;
    static_assert(NZ == 16 || NZ == 12 || NZ == 8, "pk_dft16: zero tail of 0, 4 or 8 inputs");
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        if (NZ == 16) pk_bfly4<DIR>(v[b], v[4 + b], v[8 + b], v[12 + b]);
        else if (NZ == 12) pk_bfly4_z3<DIR>(v[b], v[4 + b], v[8 + b], v[12 + b]);
        else pk_bfly4_z23<DIR>(v[b], v[4 + b], v[8 + b], v[12 + b]);
    }
    // step 2: t_b[c] *= W_16^(b c) = (c - i DIR s); constants in SGPR pairs; v[10] *= -+i is folded into the third
    // pass (pk_bfly4_rot2)
    {
        const v2f w1 = {C1, S1}, w2 = {RH, RH}, w3 = {S1, C1}, w6 = {-RH, RH}, w9 = {-C1, -S1};
        constexpr bool CJ = DIR > 0;
        const v2f p5 = pk_cmul_p_s(v[5], w1), p9 = pk_cmul_p_s(v[9], w2), p13 = pk_cmul_p_s(v[13], w3);
        const v2f p6 = pk_cmul_p_s(v[6], w2), p14 = pk_cmul_p_s(v[14], w6), p7 = pk_cmul_p_s(v[7], w3);
        const v2f p11 = pk_cmul_p_s(v[11], w6), p15 = pk_cmul_p_s(v[15], w9);
        v[5] = pk_cmul_q_s<CJ>(v[5], w1, p5);
        v[9] = pk_cmul_q_s<CJ>(v[9], w2, p9);
        v[13] = pk_cmul_q_s<CJ>(v[13], w3, p13);
        v[6] = pk_cmul_q_s<CJ>(v[6], w2, p6);
        v[14] = pk_cmul_q_s<CJ>(v[14], w6, p14);
        v[7] = pk_cmul_q_s<CJ>(v[7], w3, p7);
        v[11] = pk_cmul_q_s<CJ>(v[11], w6, p11);
        v[15] = pk_cmul_q_s<CJ>(v[15], w9, p15);
    }
    pk_bfly4<DIR>(v[0], v[1], v[2], v[3]);
    pk_bfly4<DIR>(v[4], v[5], v[6], v[7]);
    pk_bfly4_rot2<DIR>(v[8], v[9], v[10], v[11]);
    pk_bfly4<DIR>(v[12], v[13], v[14], v[15]);
    v2f t;
    t = v[1]; v[1] = v[4]; v[4] = t;
    t = v[2]; v[2] = v[8]; v[8] = t;
    t = v[3]; v[3] = v[12]; v[12] = t;
    t = v[6]; v[6] = v[9]; v[9] = t;
    t = v[7]; v[7] = v[13]; v[13] = t;
    t = v[11]; v[11] = v[14]; v[14] = t;
}
