// Front end on device (SURVEY 8f "next" #1): the per-block chain of the reference's main.py:105-166
//   deinterleave_IQ (signal_utils.py:19-22)  ->  frequency_shift with a block phase (:24-27, main.py:125-149)
//   ->  resample = scipy.signal.resample_poly(x, up, dn, padtype='line') (:15-17)
// fused into ONE kernel: raw interleaved scalars in, IF complex64 samples out, nothing in between
// touches HBM (the reference materialises a complex64 and two complex128 arrays per block).
//
// Polyphase form of resample_poly (the same closed form the CPU checker restates and pins):
//   y[m] = sum_j h[(t mod up) + up j] xe[t div up - j],   t = (m + n_pre_remove) dn,
// h = firwin(20 max(up,dn)+1, 1/max(up,dn), ('kaiser',5.0)) * up, zero-padded in front (host side,
// scipy), xe = the tuned block extended linearly through its first and last sample (upfirdn 'line').
// Tuning keeps the reference's arithmetic where it is observable: the phase ramp is float32 (sample index
// held as complex64) and the block phase is added in double (array phase_offset promotes to complex128)
// -- at 100 kHz offset the float32 ramp is quantised to 1/16 rad, so this has to be reproduced, not
// improved.  The exponential of that double phase is taken to float32 accuracy (fe_sincos below): the
// tuned samples enter a float32 FIR either way.
//
// Two kernels: frontend_group_kernel (round 4: `up` consecutive outputs per thread; what runs for up <= 16
// when 64 x down input samples + the filter rows fit LDS twice per CU) and frontend_kernel (one output per
// thread: any other ratio).  PRC_OPT_FE_METHOD chooses for A/B runs.
#include "common.h"
#include <math.h>
#include <string.h>
#include <vector>

#define FE_THREADS 256


struct FeArgs {
    const void* raw;
    float2* out;
    const void* raw2;         // second channel of the same blocks (two-channel group kernel), tuned with the SAME phase
    float2* out2;
    const float* taps;        // [J][up] polyphase layout: taps[j*up + p] = h[p + up*j]
    const double* phases;     // per block phase offset (device), or nullptr
    int64_t raw_stride;       // elements of the raw type between blocks (complex elements for C64)
    int64_t out_stride;
    int64_t n_in;             // complex samples per block
    int64_t n_out;
    int32_t up, dn, J, n_pre_remove;
    int32_t mix;              // apply the frequency shift
    int32_t opw;              // outputs per workgroup (<= FE_THREADS; fewer when the decimation ratio is large)
    int32_t src;              // prc_raw_dtype, for the group kernel (it branches on the type once per window)
    PhaseRamp pr;
    // Block phases of a launch of up to FE_PH_INLINE blocks travel INSIDE the kernel arguments (round 6): the caller's host
    // array is read while the call is being made and never again, and no copy command is queued anywhere -- an asynchronous
    // copy out of the caller's (temporary, pageable) array was a use-after-return under eight caller threads, and a copy
    // out of pinned plan memory put a DMA command on the compute stream that queued behind the recordings' 150 MB
    // host-to-device copies (host-to-host step 206 -> 324 ms).  Longer launches fill d_phases with small kernels whose
    // ARGUMENTS carry the values (fe_set_phases_kernel).
    int32_t ph_n;             // > 0: ph_inline holds this launch's block phases
    double ph_inline[32];
};
#define FE_PH_INLINE 32
struct FePhChunk { double v[FE_PH_INLINE]; int32_t n; };
__global__ void fe_set_phases_kernel(FePhChunk c, double* dst) {
    if ((int)threadIdx.x < c.n) dst[threadIdx.x] = c.v[threadIdx.x];
}
__device__ __forceinline__ double fe_block_phase(const FeArgs& a, int b) {
    if (!a.mix) return 0.0;
    if (a.ph_n > 0) return a.ph_inline[b];
    return a.phases ? a.phases[b] : 0.0;
}
#define FE_SRC_RT 99      // fe_block<FE_SRC_RT>: the block's base address from FeArgs.src

template <int SRC>
__device__ __forceinline__ float2 fe_load(const void* raw, int64_t i) {
    if (SRC == PRC_RAW_I8) {
        const signed char* p = (const signed char*)raw;
        return make_float2((float)p[2 * i], (float)p[2 * i + 1]);
    } else if (SRC == PRC_RAW_U8) {
        const unsigned char* p = (const unsigned char*)raw;
        return make_float2((float)p[2 * i], (float)p[2 * i + 1]);
    } else if (SRC == PRC_RAW_I16) {
        const short* p = (const short*)raw;
        return make_float2((float)p[2 * i], (float)p[2 * i + 1]);
    } else if (SRC == PRC_RAW_F32) {
        const float* p = (const float*)raw;
        return make_float2(p[2 * i], p[2 * i + 1]);
    } else {
        return ((const float2*)raw)[i];
    }
}

// sin / cos of a phase of up to ~1e9 rad to float32 accuracy: quadrant reduction in double (the argument is the
// reference's float32 ramp plus a double block phase, 6e5 rad at the end of a PRconfig.yaml block, so the reduction cannot
// be done in float32), polynomial kernels on |r| <= pi/4 in float32 (truncation 2e-9 / 1e-10, rounding 6e-8).  The
// reference keeps the tuned block in complex128 up to resample_poly; the FIR here runs on float32 samples, so the
// rotation is only observable to that rounding.  ~30 instructions against the ~300 of a double sincos with its
// large-argument path.
// Both polynomials advance in ONE packed instruction per Horner step (sin r = r + r r^2 S(r^2) in the low half, cos r =
// 1 + r^2 C(r^2) in the high half: five v_pk_fma_f32 for the ten scalar ones), the quadrant is applied as sign bits, and
// the complex product is the two packed instructions of fft_pk.h's pk_cmul (the same roundings as written out there).
typedef float v2f __attribute__((ext_vector_type(2)));   // a complex64 in an even VGPR pair: v_pk_*_f32 work on both halves
__device__ __forceinline__ v2f fe_sincos(double x) {     // (cos x, sin x)
    const double q = rint(x * 0.63661977236758134308);
    double r = fma(-q, 1.5707963267948965580, x);
    r = fma(-q, 6.123233995736766036e-17, r);
    const float rf = (float)r;
    const int qi = (int)q;
    const float r2 = rf * rf;
    const v2f r22 = v2f{r2, r2};
    v2f P = __builtin_elementwise_fma(v2f{2.7557314297e-6f, -2.7557314297e-7f}, r22, v2f{-1.9841270114e-4f, 2.4801587642e-5f});
    P = __builtin_elementwise_fma(P, r22, v2f{8.3333337680e-3f, -1.3888889225e-3f});
    P = __builtin_elementwise_fma(P, r22, v2f{-1.6666667163e-1f, 4.1666667908e-2f});
    P = __builtin_elementwise_fma(P, r22, v2f{0.f, -0.5f});          // low: r^2 S (an exact +0); high: the next step of C
    P = __builtin_elementwise_fma(P, v2f{rf, r2}, v2f{rf, 1.0f});    // (sin r, cos r)
    // x = r + q pi/2: odd q swaps the two; sin is negative in quadrants 2, 3, cos in 1, 2 (bit 1 of q, of q + 1)
    const unsigned t = (unsigned)qi << 30;
    const float ss = (qi & 1) ? P.y : P.x, cc = (qi & 1) ? P.x : P.y;
    return v2f{__uint_as_float(__float_as_uint(cc) ^ ((t + 0x40000000u) & 0x80000000u)),
               __uint_as_float(__float_as_uint(ss) ^ (t & 0x80000000u))};
}

// tuned sample i of the block (reference: complex128 after the array phase offset; rounded to float32 here, the
// type the resampler's FIR runs in)
// The rotation factor of sample i (it depends on the sample index and the block phase only: main.py:133-149 tunes the
// reference and the surveillance recording with the same phases, so two channels of a block share it) ...
__device__ __forceinline__ v2f fe_twiddle(const FeArgs& a, int64_t i, double blk_phase) {
    const float ph32 = (a.pr.a32 * (float)(int)i) * a.pr.rcp32;  // float32 ramp, as the reference (n_in < 2^31: plan creation)
    return fe_sincos((double)ph32 + blk_phase);
}
// ... and its product with a sample: x t = (x.x t.x - x.y t.y, x.y t.x + x.x t.y)
__device__ __forceinline__ float2 fe_apply(float2 v, v2f t) {
    const v2f x = v2f{v.x, v.y};
    v2f p, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(p) : "v"(x), "v"(t));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(d) : "v"(x), "v"(t), "v"(p));
    return make_float2(d.x, d.y);
}
template <int MIX = -1>     // 1: rotate, 0: do not, -1: a.mix decides (a wave-uniform branch per sample)
__device__ __forceinline__ float2 fe_rotate(const FeArgs& a, float2 v, int64_t i, double blk_phase) {
    if (MIX == 0 || (MIX < 0 && !a.mix)) return v;
    return fe_apply(v, fe_twiddle(a, i, blk_phase));
}
template <int SRC>
__device__ __forceinline__ float2 fe_tuned(const FeArgs& a, const void* raw, int64_t i, double blk_phase) {
    return fe_rotate(a, fe_load<SRC>(raw, i), i, blk_phase);
}

// 'line' extension of upfirdn: xe[i] = x[0] + i*slope (i < 0),  x[n-1] + (i-(n-1))*slope (i >= n)
struct FeLine {
    double2 x0, xl, slope;
};
template <int SRC>
__device__ __forceinline__ FeLine fe_line(const FeArgs& a, const void* raw, double blk_phase) {
    FeLine l;
    const float2 f0 = fe_tuned<SRC>(a, raw, 0, blk_phase), fl = fe_tuned<SRC>(a, raw, a.n_in - 1, blk_phase);
    l.x0 = make_double2(f0.x, f0.y);
    l.xl = make_double2(fl.x, fl.y);
    const double inv = a.n_in > 1 ? 1.0 / (double)(a.n_in - 1) : 0.0;
    l.slope = make_double2((l.xl.x - l.x0.x) * inv, (l.xl.y - l.x0.y) * inv);
    return l;
}
template <int SRC>
__device__ __forceinline__ float2 fe_extended(const FeArgs& a, const void* raw, int64_t i, double blk_phase, const FeLine& l) {
    if (i < 0) return make_float2((float)(l.x0.x + (double)i * l.slope.x), (float)(l.x0.y + (double)i * l.slope.y));
    if (i >= a.n_in) {
        const double d = (double)(i - (a.n_in - 1));
        return make_float2((float)(l.xl.x + d * l.slope.x), (float)(l.xl.y + d * l.slope.y));
    }
    return fe_tuned<SRC>(a, raw, i, blk_phase);
}

template <int SRC>
__device__ __forceinline__ const void* fe_block(const FeArgs& a, int b) {
    if (SRC == FE_SRC_RT) {
        const int64_t bytes = a.src <= PRC_RAW_U8 ? 1 : (a.src == PRC_RAW_I16 ? 2 : (a.src == PRC_RAW_F32 ? 4 : 8));
        return (const char*)a.raw + (int64_t)b * a.raw_stride * bytes;
    }
    if (SRC == PRC_RAW_I8 || SRC == PRC_RAW_U8) return (const char*)a.raw + (int64_t)b * a.raw_stride;
    if (SRC == PRC_RAW_I16) return (const short*)a.raw + (int64_t)b * a.raw_stride;
    if (SRC == PRC_RAW_F32) return (const float*)a.raw + (int64_t)b * a.raw_stride;
    return (const float2*)a.raw + (int64_t)b * a.raw_stride;
}

template <int SRC>
__global__ __launch_bounds__(FE_THREADS) void frontend_kernel(FeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* H = reinterpret_cast<float*>(smem_raw);                  // J * up
    float2* X = reinterpret_cast<float2*>(H + ((a.J * a.up + 1) & ~1));   // staged tuned inputs
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const void* raw = fe_block<SRC>(a, b);
    const double blk_phase = fe_block_phase(a, b);
    for (int i = tid; i < a.J * a.up; i += FE_THREADS) H[i] = a.taps[i];

    // outputs of this workgroup and the input span they touch
    const int64_t m0 = (int64_t)blockIdx.x * a.opw;
    int64_t m1 = m0 + a.opw;
    if (m1 > a.n_out) m1 = a.n_out;
    const int64_t i_hi = ((m1 - 1 + a.n_pre_remove) * a.dn) / a.up;      // newest input of the last output
    const int64_t i_lo = ((m0 + a.n_pre_remove) * a.dn) / a.up - (a.J - 1);
    const int span = (int)(i_hi - i_lo + 1);
    const FeLine line = fe_line<SRC>(a, raw, blk_phase);
    for (int k = tid; k < span; k += FE_THREADS) X[k] = fe_extended<SRC>(a, raw, i_lo + k, blk_phase, line);
    __syncthreads();
    const int64_t m = m0 + tid;
    if (tid >= a.opw || m >= a.n_out) return;
    const int64_t t = (m + a.n_pre_remove) * a.dn;
    const int phase = (int)(t % a.up);
    const int base = (int)(t / a.up - i_lo);          // X index of the newest input of this output
    float2 acc = make_float2(0.f, 0.f);
    const float* Hp = H + phase;
#pragma unroll 4
    for (int j = 0; j < a.J; ++j) {
        const float hj = Hp[j * a.up];
        const float2 x = X[base - j];
        acc.x = fmaf(hj, x.x, acc.x);
        acc.y = fmaf(hj, x.y, acc.y);
    }
    a.out[(int64_t)b * a.out_stride + m] = acc;
}


// ---- the group form (round 4): `up` consecutive outputs per thread -------------------------------------------------
// Output m = up N + q (N: group, q < up) is  y = sum_r hz[s_q - up r] xe[dn N + r],  s_q = (q + n_pre_remove) dn, the
// same sum as above with the tap index written against the input offset r from dn N.  Two things follow.  (1) The taps
// an input meets, hz[s_q - up r] for q = 0 .. up-1, depend on r alone, not on the group: a thread that owns a whole
// group reads each input ONCE from LDS and feeds `up` accumulators, and the taps are the same for every lane -- they
// come through the scalar unit from a table T[row][q] (row = r_hi - r, zero where the index leaves the filter) the
// plan made on the host.  Against one output per thread that is 1/up of the LDS reads and no tap reads at all (the
// per-output kernel is bound by its two LDS reads per multiply-add); the price is the multiply-adds on the zero
// corners of T, (dn (up-1)/up) of J + dn (up-1)/up rows (37 % at 13:119).  (2) A wavefront of 64 groups spans 64 dn
// inputs (61 KB at 13:119), so the eight wavefronts of a workgroup share ONE window and split its rows (two workgroups
// per CU = four wavefronts per SIMD); the partial sums meet in LDS, added in wavefront order every time, and leave as
// whole 128-byte lines.
// Inputs are staged tuned, as above.  dn even: a lane stride of dn complex samples would hit dn/gcd banks only; the
// window is stored with one pad sample per dn.
typedef float __attribute__((address_space(4))) fe_const_float;
typedef v2f __attribute__((address_space(4))) fe_const_v2f;     // two adjacent taps of a row: one aligned SGPR pair
#define FEG_G 64
#ifndef FEG_WAVES
#define FEG_WAVES 8        // wavefronts per workgroup: they share one window and split its rows
#endif
#define FEG_THREADS (64 * FEG_WAVES)
#define FEG_SEGS 10        // segments per wavefront (trips[w][s] == 0 ends the list)
struct FegArgs {
    const float* T;      // [rows_total][16]
    int32_t rows_total, r_first;      // r_first: input offset r of the window's first sample (= r_hi - (rows_total - 1))
    // Round 5: the tap table is a band -- a row meets only some of the `up` outputs of a group (a suffix of the columns in
    // the table's upper corner, a prefix in its lower one) -- so a wavefront works through a short list of SEGMENTS, runs of
    // trips (two rows each) that multiply the same compile-time window of columns, and the trips are dealt out so that
    // every wavefront has about the same number of multiply-adds.  code = width (1 .. NQ) | 0x100 for a suffix window.
    int16_t seg_row0[FEG_WAVES][FEG_SEGS], seg_trips[FEG_WAVES][FEG_SEGS], seg_code[FEG_WAVES][FEG_SEGS];
    int32_t lane_stride, pad, span;   // span: staged samples per workgroup (and channel)
    int32_t xstride;                  // two-channel form: LDS elements between the two channels' windows
    float inv_dn;
    // Round 6: FOLDED rows.  A column of the tap table (one of the `up` outputs of a group) meets a contiguous run of at most
    // M = max column length rows, and the runs start later for smaller q -- the table is a band: 37 % of T[rows][up] at
    // 13:119 are zeros in its two corners.  Rows rho and rho + M never both meet the same column, so they FOLD into one row
    // of M: T2[rho][q] = T[rho][q] or T[rho + M][q], whichever exists, with the input of row rho + M for the columns
    // q < q0(rho) and the input of row rho for the others.  Every folded row then carries `up` non-zero taps: M rows of
    // full width instead of rows_total (184 against 294), dealt to the wavefronts in equal runs -- no corner, no imbalance
    // (round 5's banded windows saved the same multiply-adds and lost them again to uneven wavefronts and odd widths).
    const float* T2;                  // [fold_M + 3][16]; nullptr: no folded form for this plan
    int32_t fold_M;                   // 0: the unfolded row loop
    int16_t fseg_row0[FEG_WAVES][FEG_SEGS], fseg_rows[FEG_WAVES][FEG_SEGS], fseg_q0[FEG_WAVES][FEG_SEGS];
};

#ifndef FEG_CHUNK
#define FEG_CHUNK 8      // raw samples a thread has in flight before it starts rotating them
#endif
template <int PAD = -1>     // 1: one pad sample per dn, 0: none, -1: g.pad decides
__device__ __forceinline__ int feg_at(const FeArgs& a, const FegArgs& g, int k) {
    if (PAD == 0 || (PAD < 0 && !g.pad)) return k;
    int nk = (int)((float)k * g.inv_dn);
    if (nk * a.dn > k) --nk;
    if ((nk + 1) * a.dn <= k) ++nk;
    return k + nk;
}
// Staging of one window.  FEG_CHUNK loads per thread are issued before the first rotation (a load-rotate-store chain per
// sample leaves the wavefront waiting on HBM once per sample).  The interior windows of a block -- all but two -- run
// straight-line code specialised on the sample type, the pad layout and whether the block is tuned: whole trips of
// FEG_CHUNK samples without a branch, so the compiler interleaves the eight rotation chains; the last, partial trip and
// the two windows that touch a block end (the 'line' extension) take the general form below (clamped indices, every
// choice a wave-uniform branch).
template <int SRC, int NCH>
__device__ __forceinline__ void feg_stage_general(const FeArgs& a, const FegArgs& g, float2* X, const void* const (&raw)[NCH], int64_t i_w,
                                                  double blk_phase, int k_first, bool edge) {
    FeLine line[NCH];
    if (edge) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) line[ch] = fe_line<SRC>(a, raw[ch], blk_phase);
    }
    const int last = g.span - 1;
    for (int k0 = k_first; k0 <= last; k0 += FEG_THREADS * FEG_CHUNK) {
        float2 v[NCH][FEG_CHUNK];
#pragma unroll
        for (int c = 0; c < FEG_CHUNK; ++c) {
            int64_t i = i_w + min(k0 + c * FEG_THREADS, last);
            if (edge) i = i < 0 ? 0 : (i >= a.n_in ? a.n_in - 1 : i);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) v[ch][c] = fe_load<SRC>(raw[ch], i);
        }
#pragma unroll
        for (int c = 0; c < FEG_CHUNK; ++c) {
            const int k = k0 + c * FEG_THREADS;
            const int64_t i = i_w + k;
            v2f tw = v2f{1.f, 0.f};
            if (a.mix) tw = fe_twiddle(a, i, blk_phase);
            const int at = feg_at<-1>(a, g, min(k, last));
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                float2 t = a.mix ? fe_apply(v[ch][c], tw) : v[ch][c];
                if (edge) {
                    const FeLine& ln = line[ch];
                    if (i < 0) t = make_float2((float)(ln.x0.x + (double)i * ln.slope.x), (float)(ln.x0.y + (double)i * ln.slope.y));
                    else if (i >= a.n_in) {
                        const double d = (double)(i - (a.n_in - 1));
                        t = make_float2((float)(ln.xl.x + d * ln.slope.x), (float)(ln.xl.y + d * ln.slope.y));
                    }
                }
                if (k <= last) X[ch * g.xstride + at] = t;
            }
        }
    }
}
template <int SRC, int PAD, int MIX, int NCH>
__device__ __forceinline__ void feg_stage_interior(const FeArgs& a, const FegArgs& g, float2* X, const void* const (&raw)[NCH], int64_t i_w,
                                                   double blk_phase, int tid) {
    // whole trips: the same number for every thread (a per-thread bound lets the first lanes of the workgroup take one trip
    // more than the rest: two of its wavefronts then run both forms, and everybody waits for them at the barrier)
    const int nfull = g.span / (FEG_THREADS * FEG_CHUNK);
    // The few samples after the last whole trip (6 of 4102 at 13:119 with two channels) ride in that trip as ONE more load
    // per thread (round 6): left to the general form below they cost every workgroup a second round trip to memory and
    // eight rotations on its first wavefront alone, with the other seven waiting at the barrier.
    const int rem = g.span - nfull * (FEG_THREADS * FEG_CHUNK);
#ifdef FEG_EXP_TAIL_GENERAL        // the form of rounds 4-5, for A/B runs
    const bool fold = false;
#else
    const bool fold = nfull > 0 && rem > 0 && rem <= FEG_THREADS;
#endif
    int k0 = tid;
    for (int trip = 0; trip < nfull; ++trip, k0 += FEG_THREADS * FEG_CHUNK) {
        float2 v[NCH][FEG_CHUNK];
        const bool extra = fold && trip == nfull - 1 && tid < rem;
        const int kx = nfull * (FEG_THREADS * FEG_CHUNK) + tid;
        float2 vx[NCH];
        if (extra) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) vx[ch] = fe_load<SRC>(raw[ch], i_w + kx);
        }
#pragma unroll
        for (int c = 0; c < FEG_CHUNK; ++c) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
#ifdef FEG_EXP_NOLOAD                 // timing ablation, never shipped: the staging without its global loads
                v[ch][c] = make_float2((float)(k0 + c), (float)ch);
#else
                v[ch][c] = fe_load<SRC>(raw[ch], i_w + k0 + c * FEG_THREADS);
#endif
            }
        }
        // the loads first, then everything that does not need them (the phases: most of the work) while they are in flight --
        // left alone the scheduler sinks the loads below the phase arithmetic
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < FEG_CHUNK; ++c) {
            const int k = k0 + c * FEG_THREADS;
            const int at = feg_at<PAD>(a, g, k);
#ifdef FEG_EXP_NOROT                  // timing ablation, never shipped: the staging without its rotations
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) X[ch * g.xstride + at] = v[ch][c];
#else
            if (MIX) {
                const v2f tw = fe_twiddle(a, i_w + k, blk_phase);      // ONE rotation factor for every channel of the block
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) X[ch * g.xstride + at] = fe_apply(v[ch][c], tw);
            } else {
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) X[ch * g.xstride + at] = v[ch][c];
            }
#endif
        }
        if (extra) {
            const int at = feg_at<PAD>(a, g, kx);
            if (MIX) {
                const v2f tw = fe_twiddle(a, i_w + kx, blk_phase);
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) X[ch * g.xstride + at] = fe_apply(vx[ch], tw);
            } else {
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) X[ch * g.xstride + at] = vx[ch];
            }
        }
    }
    if (!fold) feg_stage_general<SRC, NCH>(a, g, X, raw, i_w, blk_phase, k0, false);
}
template <int SRC, int NCH>
__device__ __forceinline__ void feg_stage(const FeArgs& a, const FegArgs& g, float2* X, const void* const (&raw)[NCH], int64_t i_w,
                                          double blk_phase, int tid) {
    if (i_w < 0 || i_w + g.span > a.n_in) feg_stage_general<SRC, NCH>(a, g, X, raw, i_w, blk_phase, tid, true);
    else if (g.pad) {
        if (a.mix) feg_stage_interior<SRC, 1, 1, NCH>(a, g, X, raw, i_w, blk_phase, tid);
        else feg_stage_interior<SRC, 1, 0, NCH>(a, g, X, raw, i_w, blk_phase, tid);
    } else {
        if (a.mix) feg_stage_interior<SRC, 0, 1, NCH>(a, g, X, raw, i_w, blk_phase, tid);
        else feg_stage_interior<SRC, 0, 0, NCH>(a, g, X, raw, i_w, blk_phase, tid);
    }
}

// One segment of a wavefront's rows: `ntrips` trips of two rows from `row0` on, columns [QA, QA + QN) of the tap table.
// Per trip: 2 x QN taps through the scalar unit, two inputs from LDS, 2 QN packed multiply-adds.  A row's input sits
// o = (rows_total - 1 - row) samples after the lane's first one (plus one pad sample per dn of them when dn is even): o,
// o / dn and o % dn are carried in scalar registers, nothing is looked up.
template <int QA, int QN, int NQ, bool PAD>
__device__ __forceinline__ void feg_trips(const FeArgs& a, const FegArgs& g, const float2* xl, int row0, int ntrips, v2f (&acc)[NQ]) {
    int o = g.rows_total - 1 - row0;
    int od = PAD ? o / a.dn : 0, om = PAD ? o - od * a.dn : 0;
    auto next_off = [&]() {                                      // LDS offset of the next row's input; rows only go down
        const int off = o + (PAD ? od : 0);
        --o;
        if (PAD && --om < 0) {                                   // dn even: one pad sample per dn of them (carried, not divided)
            om += a.dn;
            --od;
        }
        return off;
    };
    // wave-uniform address in the CONSTANT address space: the rows must come through the scalar unit (s_load into
    // SGPRs, the taps then ride as scalar operands of the packed multiply-adds).  Through a plain global pointer the
    // compiler only does that while it can prove that nothing in the kernel writes the table, gives up on a kernel
    // this size, and loads the taps with vector loads into VGPRs instead (measured: 8.4 -> 12.7 us per block)
    const fe_const_float* tr = (const fe_const_float*)(g.T + (size_t)row0 * 16);
    // Measured in round 5 (profiles/r05_frontend_ab.md): requesting a trip's taps and inputs one trip ahead of their use
    // (two register sets in turn, explicit lgkmcnt(0) waits) changes nothing, before and after the scalar work below was
    // removed, and neither does dealing the rows out by cost with narrower column windows in the table's corners
    // (PRC_OPT_FE_BALANCE: 37 % fewer multiply-adds), nor does a loop of four rows per iteration.
#ifdef FEG_EXP_SAMEROW                    // timing ablation, never shipped: every trip reads the same two tap rows (128 bytes of table)
#define FEG_TR_STEP 0
#else
#define FEG_TR_STEP 32
#endif
    // Taps as SGPR PAIRS (round 5).  A packed multiply-add takes a 64-bit scalar source and op_sel / op_sel_hi say which
    // half feeds each half of the result: the tap in the LOW half of a pair serves with op_sel_hi:[0,..], the one in the HIGH
    // half with op_sel:[1,..] op_sel_hi:[1,..] -- two adjacent taps of a row are one aligned pair as s_load delivers them.
    // Written as `fma(v2f{t, t}, x, acc)` the compiler wants every tap in the low half of a pair of its own and moves 26
    // scalars per trip into place: 161 M scalar instructions per launch against 184 M vector ones (SQ_INSTS_SALU /
    // SQ_INSTS_VALU), and the ONE scalar unit of a CU serves all sixteen wavefronts -- the row loop was bound by it, which
    // is why neither fewer multiply-adds nor earlier requests moved it (profiles/r05_frontend_ab.md).
    constexpr int Q0 = QA & ~1, Q1 = (QA + QN + 1) & ~1;         // whole pairs covering [QA, QA + QN); the extra columns are zeros
    static_assert(Q1 <= 16, "a tap row has 16 slots");
    constexpr int NP = (Q1 - Q0) / 2;                            // tap pairs per row
    auto use = [&](const v2f (&pa)[NP], const v2f (&pb)[NP], float2 x0, float2 x1) {
        const v2f xa = v2f{x0.x, x0.y}, xb = v2f{x1.x, x1.y};
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int q = Q0 + 2 * j;
            if (q < NQ) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[q < NQ ? q : 0]) : "s"(pa[j]), "v"(xa));
            if (q + 1 < NQ) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[q + 1 < NQ ? q + 1 : 0]) : "s"(pa[j]), "v"(xa));
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int q = Q0 + 2 * j;
            if (q < NQ) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[q < NQ ? q : 0]) : "s"(pb[j]), "v"(xb));
            if (q + 1 < NQ) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[q + 1 < NQ ? q + 1 : 0]) : "s"(pb[j]), "v"(xb));
        }
    };
    auto request = [&](v2f (&pa)[NP], v2f (&pb)[NP], float2& x0, float2& x1) {
        const fe_const_v2f* tp = (const fe_const_v2f*)tr;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            pa[j] = tp[(Q0 >> 1) + j];
            pb[j] = tp[8 + (Q0 >> 1) + j];
        }
        tr += FEG_TR_STEP;
        const int off0 = next_off(), off1 = next_off();
        x0 = xl[off0];
        x1 = xl[off1];
    };
#pragma unroll 1
    for (int t = 0; t < ntrips; ++t) {
        v2f pa[NP], pb[NP];
        float2 x0, x1;
        request(pa, pb, x0, x1);
        use(pa, pb, x0, x1);
    }
}
// One run of FOLDED rows (FegArgs.T2): `nrows` rows from `row0` on, the columns q < Q0 multiply the input of row + M (xb), the
// others the input of the row itself (xa).  Only for odd decimations (no pad samples: an input's LDS offset is its row).
template <int Q0, int NQ>
__device__ __forceinline__ void feg_rows_folded(const FegArgs& g, const float2* xl, int row0, int nrows, v2f (&acc)[NQ]) {
    constexpr int NP = (NQ + 1) / 2;                             // tap pairs per row
    const fe_const_float* tr = (const fe_const_float*)(g.T2 + (size_t)row0 * 16);
    const float2* xa_p = xl + (g.rows_total - 1 - row0);         // rows only go down: the input of the next row is one sample earlier
    const float2* xb_p = xa_p - g.fold_M;
    auto row = [&](const v2f (&pa)[NP], float2 fa, float2 fb) {
        const v2f xa = v2f{fa.x, fa.y}, xb = v2f{fb.x, fb.y};
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int q = 2 * j;
            if (q < NQ) {
                if (q < Q0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[q < NQ ? q : 0]) : "s"(pa[j]), "v"(xb));
                else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[q < NQ ? q : 0]) : "s"(pa[j]), "v"(xa));
            }
            if (q + 1 < NQ) {
                if (q + 1 < Q0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[q + 1 < NQ ? q + 1 : 0]) : "s"(pa[j]), "v"(xb));
                else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[q + 1 < NQ ? q + 1 : 0]) : "s"(pa[j]), "v"(xa));
            }
        }
    };
    // two rows per trip (their taps and inputs requested together), then the odd one
    int r = 0;
#pragma unroll 1
    for (; r + 2 <= nrows; r += 2) {
        const fe_const_v2f* tp = (const fe_const_v2f*)tr;
        v2f pa[NP], pb[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) { pa[j] = tp[j]; pb[j] = tp[8 + j]; }
        tr += 32;
        const float2 fa0 = xa_p[0], fa1 = xa_p[-1];
        float2 fb0 = fa0, fb1 = fa1;
        if (Q0 > 0) { fb0 = xb_p[0]; fb1 = xb_p[-1]; }
        xa_p -= 2;
        xb_p -= 2;
        row(pa, fa0, fb0);
        row(pb, fa1, fb1);
    }
    if (r < nrows) {
        const fe_const_v2f* tp = (const fe_const_v2f*)tr;
        v2f pa[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) pa[j] = tp[j];
        const float2 fa = xa_p[0];
        float2 fb = fa;
        if (Q0 > 0) fb = xb_p[0];
        row(pa, fa, fb);
    }
}
template <int Q_, int NQ>
__device__ __forceinline__ void feg_fold_segment(const FegArgs& g, const float2* xl, int row0, int nrows, int q0, v2f (&acc)[NQ]) {
    if constexpr (Q_ <= NQ) {
        if (q0 == Q_) feg_rows_folded<Q_, NQ>(g, xl, row0, nrows, acc);
        else feg_fold_segment<Q_ + 1, NQ>(g, xl, row0, nrows, q0, acc);
    }
}

// the column window of a segment is a compile-time choice among the prefixes [0, w) and the suffixes [NQ - w, NQ)
template <int W_, int NQ>
__device__ __forceinline__ void feg_segment(const FeArgs& a, const FegArgs& g, const float2* xl, int row0, int ntrips, int code,
                                            v2f (&acc)[NQ]) {
    if constexpr (W_ <= NQ) {
        if (code == W_) {
            if (g.pad) feg_trips<0, W_, NQ, true>(a, g, xl, row0, ntrips, acc);
            else feg_trips<0, W_, NQ, false>(a, g, xl, row0, ntrips, acc);
        } else if (code == (W_ | 0x100)) {
            if (g.pad) feg_trips<NQ - W_, W_, NQ, true>(a, g, xl, row0, ntrips, acc);
            else feg_trips<NQ - W_, W_, NQ, false>(a, g, xl, row0, ntrips, acc);
        }
        else feg_segment<W_ + 1, NQ>(a, g, xl, row0, ntrips, code, acc);
    }
}

// NCH = 2 (prc_frontend_execute2): both channels of a block in ONE workgroup.  A wavefront's lanes are 32 groups x 2
// channels (lane = 32 channel + group): the window per channel is half as long (32 dn inputs + the rows), the two
// windows together take the LDS one 64-group window took, and the row loop is the same instruction stream -- the taps are
// wave-uniform whatever channel a lane reads.  What is saved is the staging: one rotation factor per input sample serves
// both channels (31 of the ~40 instructions a staged sample costs).
template <int NQ, int NCH>
__global__ __launch_bounds__(FEG_THREADS, 2 * FEG_WAVES / 4) void frontend_group_kernel(FeArgs a, FegArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* X = reinterpret_cast<float2*>(smem_raw);
    constexpr int G = FEG_G / NCH;                                  // groups per workgroup
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const void* raw[NCH];
    raw[0] = fe_block<FE_SRC_RT>(a, b);
    if (NCH == 2) {
        FeArgs a2 = a;
        a2.raw = a.raw2;
        raw[NCH - 1] = fe_block<FE_SRC_RT>(a2, b);
    }
    const double blk_phase = fe_block_phase(a, b);
#ifdef FEG_EXP_STAGGER              // A/B, never shipped (round 6: 7.1-7.3 us per block-channel with 3 / 7 / 10 us of delay, 7.0-7.3 without):
    {                               // the second workgroup of every CU in the first dispatch round starts late, so that the two
                                    // workgroups of a CU are in different phases (staging against row loop) from then on
        const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
        if (lin >= 256u && lin < 512u)
            for (int i = 0; i < FEG_EXP_STAGGER; ++i) __builtin_amdgcn_s_sleep(64);
    }
#endif
    const int64_t N0 = (int64_t)blockIdx.x * G;                     // first group of the workgroup
    const int64_t i_w = N0 * a.dn + g.r_first;                      // input index of the window's first sample
#ifndef FEG_EXP_NOSTAGE              // timing ablation, never shipped: no staging at all (the row loop reads whatever LDS holds)
    switch (a.src) {            // one scalar branch per window, not per sample
        case PRC_RAW_I8: feg_stage<PRC_RAW_I8, NCH>(a, g, X, raw, i_w, blk_phase, tid); break;
        case PRC_RAW_U8: feg_stage<PRC_RAW_U8, NCH>(a, g, X, raw, i_w, blk_phase, tid); break;
        case PRC_RAW_I16: feg_stage<PRC_RAW_I16, NCH>(a, g, X, raw, i_w, blk_phase, tid); break;
        case PRC_RAW_F32: feg_stage<PRC_RAW_F32, NCH>(a, g, X, raw, i_w, blk_phase, tid); break;
        default: feg_stage<PRC_RAW_C64, NCH>(a, g, X, raw, i_w, blk_phase, tid);
    }
#endif
    __syncthreads();
    const float2* xl = NCH == 2 ? X + (lane >> 5) * g.xstride + (lane & 31) * g.lane_stride : X + lane * g.lane_stride;
    v2f acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = v2f{0.f, 0.f};
#ifdef FEG_EXP_NOFIR                      // timing ablation, never shipped: one trip
    feg_trips<0, NQ, NQ, false>(a, g, xl, 0, 1, acc);
#else
    if (g.fold_M > 0) {
#pragma unroll 1
        for (int sgm = 0; sgm < FEG_SEGS; ++sgm) {
            const int nrows = __builtin_amdgcn_readfirstlane((int)g.fseg_rows[w][sgm]);
            if (nrows == 0) break;
            feg_fold_segment<0, NQ>(g, xl, __builtin_amdgcn_readfirstlane((int)g.fseg_row0[w][sgm]), nrows,
                                    __builtin_amdgcn_readfirstlane((int)g.fseg_q0[w][sgm]), acc);
        }
    } else {
#pragma unroll 1
        for (int sgm = 0; sgm < FEG_SEGS; ++sgm) {
            const int ntrips = __builtin_amdgcn_readfirstlane((int)g.seg_trips[w][sgm]);
            if (ntrips == 0) break;
            feg_segment<1, NQ>(a, g, xl, __builtin_amdgcn_readfirstlane((int)g.seg_row0[w][sgm]), ntrips,
                               __builtin_amdgcn_readfirstlane((int)g.seg_code[w][sgm]), acc);
        }
    }
#endif
#ifdef FEG_EXP_NOEPI                  // timing ablation, never shipped: every wavefront stores its own partial sums, nothing is added up
    {
        float2* out = a.out + (int64_t)b * a.out_stride;
        const int64_t m = N0 * a.up + (int64_t)(lane & (G - 1)) * a.up;
        if (w == 0 && m + NQ <= a.n_out)
            for (int q = 0; q < NQ; ++q) out[m + q] = make_float2(acc[q].x, acc[q].y);
        return;
    }
#endif
    __syncthreads();                                                // the window is dead: its LDS takes the partial sums
    // [wave][lane][q] at an odd pitch: a lane's `up` sums are stored pitch samples from its neighbour's (b64 stores,
    // conflict-free for an odd pitch) and read back as what they are, consecutive outputs ([wave][q][lane] measured 68 %
    // of all LDS cycles of the kernel as bank conflicts: thirteen lanes of a read on one bank)
    float2* P = reinterpret_cast<float2*>(smem_raw);
    constexpr int pitch = NQ | 1;
#pragma unroll
    for (int q = 0; q < NQ; ++q) P[(w * FEG_G + lane) * pitch + q] = make_float2(acc[q].x, acc[q].y);
    __syncthreads();
    const int64_t M0 = N0 * a.up;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        float2* out = (ch ? a.out2 : a.out) + (int64_t)b * a.out_stride;
        for (int o = tid; o < G * a.up; o += FEG_THREADS) {
            const int64_t m = M0 + o;
            if (m >= a.n_out) break;
            const int n = o / a.up, q = o - n * a.up;
            const int at = (ch * G + n) * pitch + q;
            float2 sum = P[at];
#pragma unroll
            for (int v = 1; v < FEG_WAVES; ++v) {                   // wavefront 0's rows first, in order, every time
                const float2 pv = P[v * FEG_G * pitch + at];
                sum.x += pv.x;
                sum.y += pv.y;
            }
            out[m] = sum;
        }
    }
}

struct prc_frontend_plan {
    prc_frontend_desc desc;
    float* d_taps = nullptr;     // polyphase layout
    double* d_phases = nullptr;  // max_blocks
    float* d_T = nullptr;        // group form: tap rows (nullptr: up > 16 or the window does not fit LDS)
    float* d_T2 = nullptr;       // folded tap rows (nullptr: even decimation, or the fold does not apply)
    FegArgs g = {};
    size_t g_lds = 0;
    FegArgs g2 = {};             // two-channel form (32 groups per channel and workgroup); g2_lds = 0: not available
    size_t g2_lds = 0;
    int J = 0;
    int64_t n_in = 0, n_out = 0;
    std::mutex mtx;
};

extern "C" int prc_frontend_plan_destroy(prc_frontend_plan* p) {
    if (!p) return PRC_OK;
    if (p->d_taps) (void)hipFree(p->d_taps);
    if (p->d_phases) (void)hipFree(p->d_phases);
    if (p->d_T) (void)hipFree(p->d_T);
    if (p->d_T2) (void)hipFree(p->d_T2);
    delete p;
    return PRC_OK;
}

extern "C" int prc_frontend_plan_create(prc_frontend_plan** plan, const prc_frontend_desc* host_desc) {
    PRC_REQUIRE(plan && host_desc, PRC_EINVAL, "prc_frontend_plan_create: null argument");
    static_assert(sizeof(prc_frontend_desc) == PRC_FRONTEND_DESC_SIZE_600, "prc_frontend_desc grew: keep PRC_FRONTEND_DESC_SIZE_600, default the new fields to 0");
    prc_frontend_desc mine;
    if (int rc = prc_take_desc(&mine, host_desc, PRC_FRONTEND_DESC_SIZE_600, "prc_frontend_plan_create", "prc_frontend_desc")) return rc;
    const prc_frontend_desc* d = &mine;
    PRC_REQUIRE(d->n_in > 1 && d->n_in < ((int64_t)1 << 30) && d->up > 0 && d->down > 0 && d->ntaps > 0 && d->taps_host &&
                d->max_blocks > 0, PRC_EINVAL, "prc_frontend_plan_create: bad size (blocks of 2 .. 2^30 complex samples)");
    PRC_REQUIRE(d->raw_dtype >= PRC_RAW_I8 && d->raw_dtype <= PRC_RAW_C64, PRC_EINVAL,
                "prc_frontend_plan_create: unknown raw dtype %d", d->raw_dtype);
    prc_frontend_plan* p = new prc_frontend_plan();
    p->desc = *d;
    p->desc.taps_host = nullptr;
    p->n_in = d->n_in;
    int64_t no = d->n_in * d->up;
    p->n_out = no / d->down + (no % d->down ? 1 : 0);
    p->J = (d->ntaps + d->up - 1) / d->up;
    std::vector<float> poly((size_t)p->J * d->up, 0.f);
    for (int k = 0; k < d->ntaps; ++k) poly[(size_t)(k / d->up) * d->up + (k % d->up)] = d->taps_host[k];
    hipError_t e = hipMalloc(&p->d_taps, sizeof(float) * poly.size());
    if (e == hipSuccess) e = hipMemcpy(p->d_taps, poly.data(), sizeof(float) * poly.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&p->d_phases, sizeof(double) * d->max_blocks);
    // the group form's tables: row = r_hi - r, T[row][q] = hz[s_q - up r], s_q = (q + n_pre_remove) dn
    if (e == hipSuccess && d->up <= 16) {
        const int64_t up = d->up, dn = d->down;
        const int64_t s0 = (int64_t)d->n_pre_remove * dn, sl = (up - 1 + d->n_pre_remove) * dn;
        const int64_t r_hi = sl / up, r_lo = s0 / up - (p->J - 1);
        const int64_t nrows = r_hi - r_lo + 1;
        const int pad = (dn % 2 == 0) ? 1 : 0;
        const int64_t nq = up <= 4 ? 4 : (up <= 8 ? 8 : (up <= 13 ? 13 : 16));      // accumulators per thread (the kernel's NQ)
        // which columns a row reaches: T[row][q] != 0 iff 0 <= (q + n_pre_remove) dn - up (r_hi - row) < ntaps
        auto col_range = [&](int64_t row, int& lo, int& hi) {       // [lo, hi) over q < up; lo == hi: an all-zero row
            lo = (int)up;
            hi = 0;
            for (int64_t q = 0; q < up; ++q) {
                const int64_t idx = (q + d->n_pre_remove) * dn - up * (r_hi - row);
                if (idx >= 0 && idx < d->ntaps) {
                    if (q < lo) lo = (int)q;
                    if (q + 1 > hi) hi = (int)q + 1;
                }
            }
            if (hi < lo) lo = hi = 0;
        };
        // Trips (two consecutive rows) and the narrowest compile-time column window that holds what their rows reach: a
        // prefix [0, w) or a suffix [nq - w, nq).  A trip costs 2 w multiply-adds + ~8 instructions of addressing, LDS
        // reads and loop; consecutive trips with the same window form a segment (~25 instructions to set up).  The trips
        // are dealt to the FEG_WAVES wavefronts in order, cut where the running cost passes the next eighth of the total.
        // PRC_OPT_FE_BALANCE = 0: equal runs of rows at full width (rounds 3-4).
        const int64_t trip_overhead = prc_opt(PRC_OPT_FE_BALANCE);          // 0: equal runs; else the cost model's constant per trip
        const bool balance = trip_overhead != 0;
        const int64_t ntrip_all = balance ? (nrows + 1) / 2 : FEG_WAVES * ((((nrows + FEG_WAVES - 1) / FEG_WAVES + 1) & ~(int64_t)1) / 2);
        std::vector<int> tcode((size_t)ntrip_all);
        std::vector<int64_t> tcost((size_t)ntrip_all);
        int64_t cost_all = 0;
        for (int64_t t = 0; t < ntrip_all; ++t) {
            int lo = (int)nq, hi = 0;
            for (int64_t r = 2 * t; r < 2 * t + 2 && r < nrows; ++r) {
                int l, h;
                col_range(r, l, h);
                if (h > l) {
                    if (l < lo) lo = l;
                    if (h > hi) hi = h;
                }
            }
            int code = (int)nq;                                       // full width
            if (balance) {
                if (hi <= lo) code = 1;                                // a pair of all-zero rows (padding): the narrowest window
                else if (hi <= (int)nq - lo) code = hi;                // prefix [0, hi)
                else code = ((int)nq - lo) | 0x100;                    // suffix [lo, nq)
                if ((code & 0xff) >= (int)nq) code = (int)nq;
            }
            tcode[(size_t)t] = code;
            tcost[(size_t)t] = 2 * (code & 0xff) + trip_overhead;
            cost_all += tcost[(size_t)t];
        }
        int16_t seg_row0[FEG_WAVES][FEG_SEGS] = {}, seg_trips[FEG_WAVES][FEG_SEGS] = {}, seg_code[FEG_WAVES][FEG_SEGS] = {};
        bool seg_fit = true;
        {
            int64_t t = 0, run = 0;
            for (int w = 0; w < FEG_WAVES; ++w) {
                const int64_t until = balance ? cost_all * (w + 1) / FEG_WAVES : 0;
                const int64_t t_end_equal = (w + 1) * (ntrip_all / FEG_WAVES);
                int nseg = 0;
                while (t < ntrip_all) {
                    if (balance ? (run + tcost[(size_t)t] / 2 > until && w + 1 < FEG_WAVES) : (t >= t_end_equal)) break;
                    if (nseg > 0 && seg_code[w][nseg - 1] == tcode[(size_t)t]) {
                        ++seg_trips[w][nseg - 1];
                    } else if (nseg < FEG_SEGS) {
                        seg_row0[w][nseg] = (int16_t)(2 * t);
                        seg_trips[w][nseg] = 1;
                        seg_code[w][nseg] = (int16_t)tcode[(size_t)t];
                        ++nseg;
                    } else {
                        // more windows than a wavefront's list holds (never at the ratios in use): widen the last segment to
                        // the full width and let it take the rest of this wavefront's trips
                        seg_code[w][nseg - 1] = (int16_t)nq;
                        ++seg_trips[w][nseg - 1];
                    }
                    run += tcost[(size_t)t];
                    ++t;
                }
            }
            seg_fit = t == ntrip_all;
        }
        const int64_t rows_total = 2 * ntrip_all;
        const int64_t o_max = rows_total - 1;
        const int64_t span = dn * (FEG_G - 1) + rows_total;
        const int64_t x_elems = span + pad * ((span - 1) / dn) + 1;
        size_t lds = sizeof(float2) * (size_t)x_elems;
        const size_t lds_p = sizeof(float2) * FEG_WAVES * (size_t)(nq | 1) * FEG_G;
        if (lds < lds_p) lds = lds_p;
        if (lds <= 78 * 1024 && seg_fit) {                                 // two workgroups per CU
            std::vector<float> T((size_t)(rows_total + 3) * 16, 0.f);     // spare rows: the look-ahead of a wavefront's last trip
            for (int64_t row = 0; row < rows_total; ++row) {
                const int64_t r = r_hi - row;
                for (int64_t q = 0; q < up; ++q) {
                    const int64_t idx = (q + d->n_pre_remove) * dn - up * r;
                    if (idx >= 0 && idx < d->ntaps) T[(size_t)row * 16 + q] = d->taps_host[idx];
                }
            }
            e = hipMalloc(&p->d_T, sizeof(float) * T.size());
            if (e == hipSuccess) e = hipMemcpy(p->d_T, T.data(), sizeof(float) * T.size(), hipMemcpyHostToDevice);
            p->g.T = p->d_T;
            p->g.T2 = nullptr;
            p->g.fold_M = 0;
            memset(p->g.fseg_row0, 0, sizeof(p->g.fseg_row0));
            memset(p->g.fseg_rows, 0, sizeof(p->g.fseg_rows));
            memset(p->g.fseg_q0, 0, sizeof(p->g.fseg_q0));
            // folded rows (FegArgs): odd decimations only (no pad samples between a row's input and the next)
            if (e == hipSuccess && pad == 0) {
                auto tap_idx = [&](int64_t row, int64_t q) { return (q + d->n_pre_remove) * dn - up * (r_hi - row); };
                auto has = [&](int64_t row, int64_t q) { const int64_t i = tap_idx(row, q); return row < nrows && i >= 0 && i < d->ntaps; };
                int64_t M = 0;                                             // longest column
                for (int64_t q = 0; q < up; ++q) {
                    int64_t len = 0;
                    for (int64_t row = 0; row < nrows; ++row) len += has(row, q) ? 1 : 0;
                    if (len > M) M = len;
                }
                bool ok = M > 0 && nrows <= 2 * M && M + 3 <= 32000;
                std::vector<float> T2((size_t)(M + 3) * 16, 0.f);
                std::vector<int> q0s((size_t)(M > 0 ? M : 1), 0);
                for (int64_t rho = 0; rho < M && ok; ++rho) {
                    int q0 = 0;                                            // columns [0, q0) take row rho + M, [q0, up) row rho
                    for (int64_t q = 0; q < up; ++q) {
                        const bool lo_ = has(rho, q), hi_ = has(rho + M, q);
                        if (lo_ && hi_) ok = false;                        // cannot happen (a column is at most M rows long)
                        if (hi_) {
                            if (q != q0) ok = false;                       // the columns of row rho + M must be a prefix
                            q0 = (int)q + 1;
                            T2[(size_t)rho * 16 + q] = d->taps_host[tap_idx(rho + M, q)];
                        } else if (lo_) {
                            T2[(size_t)rho * 16 + q] = d->taps_host[tap_idx(rho, q)];
                        }
                    }
                    for (int64_t q = 0; q < q0; ++q)
                        if (has(rho, q)) ok = false;
                    q0s[(size_t)rho] = q0;
#ifdef FEG_EXP_FOLD_ONESEG                 // timing ablation, never shipped (wrong results): every folded row as a one-input row, so that
                    q0s[(size_t)rho] = 0; //   a wavefront's rows are ONE segment -- what the segment switches cost
#endif
                }
                // equal runs of folded rows per wavefront; inside a run, consecutive rows with the same q0 form a segment
                int16_t f_row0[FEG_WAVES][FEG_SEGS] = {}, f_rows[FEG_WAVES][FEG_SEGS] = {}, f_q0[FEG_WAVES][FEG_SEGS] = {};
                if (ok) {
                    // Contiguous runs of about equal COST: a row is one unit, a change of q0 inside a wavefront's run (another
                    // compile-time split: another loop, 26 accumulator registers copied across, an odd-row tail) about three --
                    // measured: one segment per wavefront would be 0.7 us per block-channel faster than three.  (Dealing every
                    // wavefront the same mix of two-input and one-input rows instead is SLOWER, 6.67 against 6.46: one more
                    // segment for everybody.)
#ifndef FEG_FOLD_SWITCH_COST
#define FEG_FOLD_SWITCH_COST 3
#endif
                    int64_t total_cost = M;
                    for (int64_t rho = 1; rho < M; ++rho) total_cost += q0s[(size_t)rho] != q0s[(size_t)rho - 1] ? FEG_FOLD_SWITCH_COST : 0;
                    int64_t rho = 0, run_cost = 0;
                    for (int w = 0; w < FEG_WAVES && ok; ++w) {
                        int nseg = 0;
                        const int64_t until = total_cost * (w + 1) / FEG_WAVES;
                        while (rho < M) {
                            const int64_t c = 1 + ((nseg > 0 && f_q0[w][nseg - 1] != q0s[(size_t)rho]) ? FEG_FOLD_SWITCH_COST : 0);
                            if (w + 1 < FEG_WAVES && nseg > 0 && run_cost + c / 2 > until) break;
                            if (nseg > 0 && f_q0[w][nseg - 1] == q0s[(size_t)rho]) ++f_rows[w][nseg - 1];
                            else if (nseg < FEG_SEGS) {
                                f_row0[w][nseg] = (int16_t)rho;
                                f_rows[w][nseg] = 1;
                                f_q0[w][nseg] = (int16_t)q0s[(size_t)rho];
                                ++nseg;
                            } else { ok = false; break; }                  // more runs than a wavefront's list holds: no fold
                            run_cost += c;
                            ++rho;
                        }
                    }
                    if (rho != M) ok = false;
                }
                if (ok) {
                    e = hipMalloc(&p->d_T2, sizeof(float) * T2.size());
                    if (e == hipSuccess) e = hipMemcpy(p->d_T2, T2.data(), sizeof(float) * T2.size(), hipMemcpyHostToDevice);
                    p->g.T2 = p->d_T2;
                    p->g.fold_M = (int32_t)M;
                    memcpy(p->g.fseg_row0, f_row0, sizeof(f_row0));
                    memcpy(p->g.fseg_rows, f_rows, sizeof(f_rows));
                    memcpy(p->g.fseg_q0, f_q0, sizeof(f_q0));
                }
            }
            p->g.rows_total = (int32_t)rows_total;
            memcpy(p->g.seg_row0, seg_row0, sizeof(seg_row0));
            memcpy(p->g.seg_trips, seg_trips, sizeof(seg_trips));
            memcpy(p->g.seg_code, seg_code, sizeof(seg_code));
            p->g.r_first = (int32_t)(r_hi - o_max);
            p->g.lane_stride = (int32_t)(dn + pad);
            p->g.pad = pad;
            p->g.span = (int32_t)span;
            p->g.inv_dn = 1.0f / (float)dn;
            p->g.xstride = 0;
            p->g_lds = lds;
            // two channels per workgroup: 32 groups each, the two windows side by side
            const int64_t span2 = dn * (FEG_G / 2 - 1) + rows_total;
            const int64_t x2 = (span2 + pad * ((span2 - 1) / dn) + 1 + 1) & ~(int64_t)1;
            size_t lds2 = sizeof(float2) * 2 * (size_t)x2;
            if (lds2 < lds_p) lds2 = lds_p;
            if (lds2 <= 78 * 1024) {
                p->g2 = p->g;
                p->g2.span = (int32_t)span2;
                p->g2.xstride = (int32_t)x2;
                p->g2_lds = lds2;
            }
        }
    }
    if (e != hipSuccess) {
        prc_set_error("prc_frontend_plan_create: device setup failed: %s", hipGetErrorString(e));
        prc_frontend_plan_destroy(p);
        return PRC_EHIP;
    }
    *plan = p;
    return PRC_OK;
}

extern "C" int prc_frontend_out_len(const prc_frontend_plan* p, int64_t* n_out) {
    PRC_REQUIRE(p && n_out, PRC_EINVAL, "prc_frontend_out_len: null argument");
    *n_out = p->n_out;
    return PRC_OK;
}

static int frontend_run(prc_frontend_plan* p, const void* raw, const void* raw2, int64_t raw_stride, int32_t mix,
                        double fc, double fs, const double* phases_host, void* out, void* out2,
                        int64_t out_stride, int32_t nblocks, void* stream_) {
    PRC_REQUIRE(p && raw && out, PRC_EINVAL, "prc_frontend_execute: null argument");
    PRC_REQUIRE(nblocks > 0 && nblocks <= p->desc.max_blocks, PRC_EINVAL,
                "prc_frontend_execute: nblocks=%d outside [1, %d]", nblocks, p->desc.max_blocks);
    PRC_REQUIRE(out_stride >= p->n_out, PRC_ESHAPE, "prc_frontend_execute: out_stride shorter than the output");
    hipStream_t stream = (hipStream_t)stream_;
    std::unique_lock<std::mutex> lk(p->mtx);
    FeArgs a;
    a.raw = raw;
    a.out = (float2*)out;
    a.raw2 = raw2;
    a.out2 = (float2*)out2;
    a.taps = p->d_taps;
    a.phases = nullptr;
    a.ph_n = 0;
#ifdef FE_EXP_PHASES_DIRECT     // the form of rounds 3-5, kept for A/B runs only: asynchronous copy straight from the caller's array
    if (mix && phases_host) {
        PRC_HIP(hipMemcpyAsync(p->d_phases, phases_host, sizeof(double) * nblocks, hipMemcpyHostToDevice, stream));
        a.phases = p->d_phases;
    }
#else
    if (mix && phases_host) {
        if (nblocks <= FE_PH_INLINE) {
            memcpy(a.ph_inline, phases_host, sizeof(double) * nblocks);
            a.ph_n = nblocks;
        } else {
            for (int b0 = 0; b0 < nblocks; b0 += FE_PH_INLINE) {
                FePhChunk c;
                c.n = nblocks - b0 < FE_PH_INLINE ? nblocks - b0 : FE_PH_INLINE;
                memcpy(c.v, phases_host + b0, sizeof(double) * c.n);
                hipLaunchKernelGGL(fe_set_phases_kernel, dim3(1), dim3(64), 0, stream, c, p->d_phases + b0);
            }
            PRC_LAUNCH_CHECK();
            a.phases = p->d_phases;
        }
    }
#endif
    a.raw_stride = raw_stride;
    a.out_stride = out_stride;
    a.n_in = p->n_in;
    a.n_out = p->n_out;
    a.up = p->desc.up;
    a.dn = p->desc.down;
    a.J = p->J;
    a.n_pre_remove = p->desc.n_pre_remove;
    a.mix = mix ? 1 : 0;
    a.pr.a32 = (float)(2.0 * 3.14159265358979323846 * fc);
    a.pr.rcp32 = 1.0f / (float)fs;
    a.pr.off32 = 0.f;
    a.pr.enabled = a.mix;
    a.src = p->desc.raw_dtype;
    a.opw = 0;
    const int64_t method = prc_opt(PRC_OPT_FE_METHOD);
    PRC_REQUIRE(method != 2 || p->d_T, PRC_EUNSUPPORTED,
                "prc_frontend_execute: the group form needs up <= 16 and a window of 64 down samples within 78 KB of LDS (up=%d, down=%d)",
                a.up, a.dn);
    const int nq = a.up <= 4 ? 4 : (a.up <= 8 ? 8 : (a.up <= 13 ? 13 : 16));   // accumulators per thread (the tap rows are zero beyond `up`)
    if (raw2 && p->d_T && p->g2_lds && method != 1) {
        // both channels of every block in one workgroup (one rotation factor per input sample for the two of them)
        dim3 grid((unsigned)ceil_div64(p->n_out, (int64_t)(FEG_G / 2) * a.up), (unsigned)nblocks);
        FegArgs gg = p->g2;
        if (!prc_opt(PRC_OPT_FE_FOLD)) gg.fold_M = 0;
#define PRC_FEG2_CASE(Q)                                                                             \
    case Q:                                                                                          \
        if (int rc_ = prc_lds_optin((const void*)frontend_group_kernel<Q, 2>, (int)p->g2_lds)) return rc_; \
        hipLaunchKernelGGL((frontend_group_kernel<Q, 2>), grid, dim3(FEG_THREADS), p->g2_lds, stream, a, gg); \
        break;
        switch (nq) { PRC_FEG2_CASE(4) PRC_FEG2_CASE(8) PRC_FEG2_CASE(13) PRC_FEG2_CASE(16) }
#undef PRC_FEG2_CASE
        PRC_LAUNCH_CHECK();
        return PRC_OK;
    }
    if (raw2) {
        // no two-channel form for this ratio (or it was switched off): the channels one after the other, same phases
        lk.unlock();
        int rc = frontend_run(p, raw, nullptr, raw_stride, mix, fc, fs, phases_host, out, nullptr, out_stride, nblocks, stream_);
        if (rc != PRC_OK) return rc;
        return frontend_run(p, raw2, nullptr, raw_stride, mix, fc, fs, phases_host, out2, nullptr, out_stride, nblocks, stream_);
    }
    if (p->d_T && method != 1) {
        dim3 grid((unsigned)ceil_div64(p->n_out, (int64_t)FEG_G * a.up), (unsigned)nblocks);
        FegArgs gg = p->g;
        if (!prc_opt(PRC_OPT_FE_FOLD)) gg.fold_M = 0;
#define PRC_FEG_CASE(Q)                                                                              \
    case Q:                                                                                          \
        if (int rc_ = prc_lds_optin((const void*)frontend_group_kernel<Q, 1>, (int)p->g_lds)) return rc_; \
        hipLaunchKernelGGL((frontend_group_kernel<Q, 1>), grid, dim3(FEG_THREADS), p->g_lds, stream, a, gg);    \
        break;
        switch (nq) { PRC_FEG_CASE(4) PRC_FEG_CASE(8) PRC_FEG_CASE(13) PRC_FEG_CASE(16) }
#undef PRC_FEG_CASE
        PRC_LAUNCH_CHECK();
        return PRC_OK;
    }
    // staged span per workgroup: opw outputs * dn/up inputs + J taps of history; opw shrinks (in steps of one
    // wavefront) until the span fits the CU's LDS, so any decimation ratio runs
    const size_t lds_taps = sizeof(float) * ((size_t)(a.J * a.up + 1) & ~(size_t)1);
    a.opw = FE_THREADS;
    size_t lds = 0;
    for (;;) {
        const int64_t span = ((int64_t)a.opw * a.dn) / a.up + a.J + 4;
        lds = lds_taps + sizeof(float2) * (size_t)span;
        if (lds <= 150 * 1024 || a.opw <= 1) break;
        a.opw = a.opw > 64 ? a.opw - 64 : a.opw / 2;
    }
    PRC_REQUIRE(lds <= 160 * 1024, PRC_EUNSUPPORTED, "prc_frontend_execute: resampling ratio %d/%d needs %zu B of LDS",
                a.up, a.dn, lds);
    dim3 grid((unsigned)ceil_div64(p->n_out, a.opw), (unsigned)nblocks);
#define PRC_FE_CASE(S)                                                                               \
    case S:                                                                                          \
        (void)hipFuncSetAttribute((const void*)frontend_kernel<S>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL(frontend_kernel<S>, grid, dim3(FE_THREADS), lds, stream, a);              \
        break;
    switch (p->desc.raw_dtype) {
        PRC_FE_CASE(PRC_RAW_I8)
        PRC_FE_CASE(PRC_RAW_U8)
        PRC_FE_CASE(PRC_RAW_I16)
        PRC_FE_CASE(PRC_RAW_F32)
        PRC_FE_CASE(PRC_RAW_C64)
    }
#undef PRC_FE_CASE
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}

extern "C" int prc_frontend_execute(prc_frontend_plan* p, const void* raw, int64_t raw_stride, int32_t mix,
                                    double fc, double fs, const double* phases_host, void* out,
                                    int64_t out_stride, int32_t nblocks, void* stream) {
    PRC_RANGE("prc_frontend_execute");
    return frontend_run(p, raw, nullptr, raw_stride, mix, fc, fs, phases_host, out, nullptr, out_stride, nblocks, stream);
}

extern "C" int prc_frontend_execute2(prc_frontend_plan* p, const void* raw_a, const void* raw_b, int64_t raw_stride,
                                     int32_t mix, double fc, double fs, const double* phases_host, void* out_a,
                                     void* out_b, int64_t out_stride, int32_t nblocks, void* stream) {
    PRC_RANGE("prc_frontend_execute2");
    PRC_REQUIRE(raw_b && out_b, PRC_EINVAL, "prc_frontend_execute2: null second channel");
    return frontend_run(p, raw_a, raw_b, raw_stride, mix, fc, fs, phases_host, out_a, out_b, out_stride, nblocks, stream);
}

// deinterleave_IQ alone (signal_utils.py:19-22): raw scalars -> complex64
template <int SRC>
__global__ void deinterleave_kernel(const void* raw, float2* out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = fe_load<SRC>(raw, i);
}

extern "C" int prc_deinterleave(const void* raw, int32_t raw_dtype, int64_t n_complex, void* out, void* stream) {
    PRC_RANGE("prc_deinterleave");
    PRC_REQUIRE(raw && out && n_complex > 0, PRC_EINVAL, "prc_deinterleave: bad argument");
    int64_t blocks = ceil_div64(n_complex, 256);
    if (blocks > 4096) blocks = 4096;
    switch (raw_dtype) {
        case PRC_RAW_I8: hipLaunchKernelGGL(deinterleave_kernel<PRC_RAW_I8>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, raw, (float2*)out, n_complex); break;
        case PRC_RAW_U8: hipLaunchKernelGGL(deinterleave_kernel<PRC_RAW_U8>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, raw, (float2*)out, n_complex); break;
        case PRC_RAW_I16: hipLaunchKernelGGL(deinterleave_kernel<PRC_RAW_I16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, raw, (float2*)out, n_complex); break;
        case PRC_RAW_F32: hipLaunchKernelGGL(deinterleave_kernel<PRC_RAW_F32>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, raw, (float2*)out, n_complex); break;
        default: prc_set_error("prc_deinterleave: unknown raw dtype %d", raw_dtype); return PRC_EINVAL;
    }
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}

// frequency_shift with an ARRAY phase offset (main.py:133-149): float32 ramp + double block phase,
// exponential in double, complex128 out (the reference's promotion).
__global__ void freq_shift_block_kernel(const float2* __restrict__ x, double2* __restrict__ y, int64_t n,
                                        PhaseRamp pr, double blk_phase) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float ph32 = (pr.a32 * (float)i) * pr.rcp32;
        double s, c;
        sincos((double)ph32 + blk_phase, &s, &c);
        const float2 v = x[i];
        y[i] = make_double2((double)v.x * c - (double)v.y * s, (double)v.x * s + (double)v.y * c);
    }
}

extern "C" int prc_frequency_shift_block(const void* x, void* y, int64_t n, double fc, double fs,
                                         double block_phase, void* stream) {
    PRC_RANGE("prc_frequency_shift_block");
    PRC_REQUIRE(x && y && n > 0 && fs != 0.0, PRC_EINVAL, "prc_frequency_shift_block: bad argument");
    PhaseRamp pr;
    pr.a32 = (float)(2.0 * 3.14159265358979323846 * fc);
    pr.rcp32 = 1.0f / (float)fs;
    pr.off32 = 0.f;
    pr.enabled = 1;
    int64_t blocks = ceil_div64(n, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(freq_shift_block_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)x, (double2*)y, n, pr, block_phase);
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}

// frequency_shift with ONE PHASE PER SAMPLE (signal_utils.py:24-27 broadcasts an array phase_offset of the signal's
// length as readily as main.py's one value per block).  NumPy's promotion decides the arithmetic: a float64 (or integer)
// array makes the sum double -- float32 ramp + double phase, exponential and product in double, complex128 out; a
// float32 array keeps the sum, the exponential and the product in float32, complex64 out.
template <bool F32>
__global__ void freq_shift_phases_kernel(const float2* __restrict__ x, void* __restrict__ y, int64_t n,
                                         PhaseRamp pr, const void* __restrict__ phase) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float ph32 = (pr.a32 * (float)i) * pr.rcp32;
        const float2 v = x[i];
        if (F32) {
            float s, c;
            sincosf(ph32 + ((const float*)phase)[i], &s, &c);
            ((float2*)y)[i] = cmul(v, make_float2(c, s));
        } else {
            double s, c;
            sincos((double)ph32 + ((const double*)phase)[i], &s, &c);
            ((double2*)y)[i] = make_double2((double)v.x * c - (double)v.y * s, (double)v.x * s + (double)v.y * c);
        }
    }
}

extern "C" int prc_frequency_shift_phases(const void* x, void* y, int64_t n, double fc, double fs,
                                          const void* phases, int32_t phases_f32, void* stream) {
    PRC_RANGE("prc_frequency_shift_phases");
    PRC_REQUIRE(x && y && phases && n > 0 && fs != 0.0, PRC_EINVAL, "prc_frequency_shift_phases: bad argument");
    PhaseRamp pr;
    pr.a32 = (float)(2.0 * 3.14159265358979323846 * fc);
    pr.rcp32 = 1.0f / (float)fs;
    pr.off32 = 0.f;
    pr.enabled = 1;
    int64_t blocks = ceil_div64(n, 256);
    if (blocks > 4096) blocks = 4096;
    if (phases_f32)
        hipLaunchKernelGGL(freq_shift_phases_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           (const float2*)x, y, n, pr, phases);
    else
        hipLaunchKernelGGL(freq_shift_phases_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           (const float2*)x, y, n, pr, phases);
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}
