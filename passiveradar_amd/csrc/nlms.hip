// Sample-recursive NLMS clutter canceller, one wavefront per independent stream.
//
// Replaces NLMS_filter, clutter_removal.py:189-249:
//   u_k[i] = ref[L+k+peek-i], i = 0..T-1;  e = srv[k+L] - w^H u;  w += mu u conj(e) / (u^H u)
// The recursion is strictly sequential in k; the parallelism is the T taps (spread over the 64
// lanes of ONE wavefront so that no workgroup barrier sits on the critical path) and the
// independent streams (one wavefront each).  This is a VALU-issue-bound wavefront dot-product +
// AXPY loop -- not HBM-bound (24 B/sample) and not MFMA-shaped.  A step is 68 packed FMAs (the 2 x 17
// complex MACs of T = 1034) + 41 other vector instructions: 274 ns alone on a SIMD, 223 ns per step with
// three streams per SIMD (round 2: 128 instructions, 374 / 280 ns).
//
// Placement matters more than anything else here: a SIMD that gets two of these wavefronts while its
// neighbour gets none runs 1.5x longer, and with one-wavefront workgroups the dispatcher does exactly that
// on the first launch after a different kernel.  So a workgroup is 4, 8 or 12 independent wavefronts
// (= 1, 2 or 3 per SIMD) and asks for more than half of the CU's LDS: exactly one workgroup per CU, its
// wavefronts dealt round-robin to the four SIMDs.  Wavefronts never share LDS data, so there is no
// workgroup barrier anywhere.
//
// Beyond the 2048 taps one wavefront's registers hold (NW = 2: up to 4096 taps, NW = 4: up to 8192; the reference
// takes any length, :189-249) a stream is ONE workgroup of NW wavefronts: consecutive tap ranges of T/NW taps each, one
// shared LDS window, and per step one exchange of the NW partial sums of w^H u through LDS (double-buffered slots, one
// workgroup barrier per step; every wavefront adds the partials in the same order, so all of them see the same error
// sample).  The step sizes of a window come from the first wavefront's prepass.  The split is for register space only: at
// T = 1034 a stream alone steps in 253 ns on one wavefront, 334 on two, 302 on four (tools/nlms_waves_probe.py).
#include "common.h"
#include <vector>

struct NlmsArgs {
    const float2* ref;
    const float2* srv;
    const float2* taps_in;   // [nstreams][T] or nullptr
    float2* out;
    float2* taps_out;        // [nstreams][T] or nullptr
    int64_t n, stride, out_stride;
    int32_t L, peek, T;
    int32_t nstreams, kt;    // kt: steps per staged LDS window
    float mu;
};

// All-lanes sum by DPP (no LDS crossbar): xor-1 / xor-2 inside quads, half-row mirror, row mirror
// -> every row of 16 lanes holds its row sum; the four row sums are read with v_readlane.
__device__ __forceinline__ float dpp_f(float x, const int which) {
    const int i = __float_as_int(x);
    int r;
    switch (which) {
        case 0: r = __builtin_amdgcn_mov_dpp(i, 0xB1, 0xF, 0xF, true); break;    // quad_perm [1,0,3,2]
        case 1: r = __builtin_amdgcn_mov_dpp(i, 0x4E, 0xF, 0xF, true); break;    // quad_perm [2,3,0,1]
        case 2: r = __builtin_amdgcn_mov_dpp(i, 0x141, 0xF, 0xF, true); break;   // row_half_mirror
        default: r = __builtin_amdgcn_mov_dpp(i, 0x140, 0xF, 0xF, true); break;  // row_mirror
    }
    return __int_as_float(r);
}
typedef float v2f __attribute__((ext_vector_type(2)));
#ifdef NLMS_PLAIN_FMA       // A/B (round 6): the same multiply-adds, same order, same roundings, as four plain v_fma_f32 instead of two packed
__device__ __forceinline__ void pk_cmac_conj(v2f& acc, v2f w, v2f u) {
    acc.x = fmaf(w.x, u.x, acc.x); acc.y = fmaf(w.x, u.y, acc.y);
    acc.x = fmaf(w.y, u.y, acc.x); acc.y = fmaf(w.y, -u.x, acc.y);
}
__device__ __forceinline__ v2f pk_cmul_conj(v2f w, v2f u) {
    v2f acc = {w.x * u.x, w.x * u.y};
    acc.x = fmaf(w.y, u.y, acc.x); acc.y = fmaf(w.y, -u.x, acc.y);
    return acc;
}
__device__ __forceinline__ void pk_cmac_bconj(v2f& w, v2f u, v2f c) {
    w.x = fmaf(c.x, u.x, w.x); w.y = fmaf(c.x, u.y, w.y);
    w.x = fmaf(c.y, u.y, w.x); w.y = fmaf(c.y, -u.x, w.y);
}
#else
// acc += conj(w) * u :  (acc.x, acc.y) += w.x (u.x, u.y);  (acc.x, acc.y) += w.y (u.y, -u.x)
__device__ __forceinline__ void pk_cmac_conj(v2f& acc, v2f w, v2f u) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n"
                 "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]"
                 : "+v"(acc) : "v"(w), "v"(u));
}
// acc = conj(w) * u (first product of a chain: no zero-initialised accumulator)
__device__ __forceinline__ v2f pk_cmul_conj(v2f w, v2f u) {
    v2f acc;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]\n"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]"
        : "=&v"(acc) : "v"(w), "v"(u));
    return acc;
}
// w += u * conj(c) :  (w.x, w.y) += c.x (u.x, u.y);  (w.x, w.y) += c.y (u.y, -u.x)
__device__ __forceinline__ void pk_cmac_bconj(v2f& w, v2f u, v2f c) {
    asm("v_pk_fma_f32 %0, %2, %1, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n"
                 "v_pk_fma_f32 %0, %2, %1, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]"
                 : "+v"(w) : "v"(u), "v"(c));
}
#endif

// a wavefront only orders its own LDS traffic (the hardware keeps one wavefront's LDS operations in order)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Both sums of a step (re, im of conj(w).u) in ONE reduction: v_permlane32_swap puts the real partials of the upper half
// next to the imaginary partials of the lower half, one add folds the halves, four DPP adds reduce the 16-lane rows (every
// lane of a row then holds the row sum), row_bcast:15 adds row 0 into row 1 and row 2 into row 3, and lanes 31 / 63 are
// read back -- 9 instructions where two separate all-lane sums took 22.
__device__ __forceinline__ void wave_allsum2(float& yr, float& yi) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_int(yr), __float_as_int(yi), false, false);
    float v = __int_as_float(sw[0]) + __int_as_float(sw[1]);     // lanes 0-31: re(l) + re(l+32); lanes 32-63: im
    v += dpp_f(v, 0);
    v += dpp_f(v, 1);
    v += dpp_f(v, 2);
    v += dpp_f(v, 3);
    // rows 1 and 3 (row_mask 0xA) receive lane 15 of the row before them; rows 0 and 2 add zero
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));
    yr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
    yi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// lane l <- lane l-1 across the whole wavefront; lane 0 keeps `lane0`
__device__ __forceinline__ float wave_shr1(float lane0, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lane0), __float_as_int(src), 0x138, 0xF, 0xF, false));
}

// Tap layout (round 3): consecutive taps per lane.  The first A = T - 64 (TPL - 1) lanes hold TPL taps, the others TPL - 1
// (A TPL + (64 - A)(TPL - 1) = T exactly: no ragged end anywhere), lane l's first tap being i0(l).  From one step to the next
// every tap's input sample moves to the next tap: inside a lane that is a register RENAME (the step loop is unrolled TPL-fold
// over a rotating register index, so nothing moves), across lanes it is one wave_shr DPP move of the lane's last valid
// sample, and the new sample enters lane 0 from a broadcast LDS read.  A short lane's spare register is re-zeroed every step
// (its index is a compile-time constant).  Round 2 spread taps lane-minor (tap = lane + 64 t) and re-read the whole window
// from LDS every step: 17 ds_read_b64 per step at T = 1034 against 1 now, and its two all-lane sums are one (wave_allsum2).
// NW = 1: MAXW independent streams per workgroup, one wavefront each.  NW > 1: one stream per workgroup of NW wavefronts.
template <int TPL, int MAXW, int NW = 1>
__global__ __launch_bounds__(64 * (NW > 1 ? NW : MAXW)) void nlms_kernel(NlmsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int KT = a.kt;
    const int wave_id = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int WIN = 64 * TPL * NW;                  // >= T: the window of all the stream's taps
    // per stream: KT + WIN float2 (reference window), KT float2 (srv in, error out), KT float (step sizes mu / u^H u)
    float2* Rw = reinterpret_cast<float2*>(smem_raw) + (NW > 1 ? (size_t)0 : (size_t)wave_id * (2 * KT + WIN + KT / 2));
    float2* D = Rw + KT + WIN;
    float* Sa = reinterpret_cast<float*>(D + KT);
    float2* S = D + KT + KT / 2;                        // NW > 1: partial sums of w^H u, [step parity][wavefront]
    const int lane = threadIdx.x & 63;
    const int tid = NW > 1 ? (int)threadIdx.x : lane;   // index and stride of the loops that stage / drain the LDS windows
    constexpr int NT = 64 * NW;
    const int b = NW > 1 ? (int)blockIdx.x : blockIdx.x * (int)(blockDim.x >> 6) + wave_id;
    if (b >= a.nstreams) return;                        // NW = 1: no barriers below, idle wavefronts just leave
    const float2* __restrict__ ref = a.ref + (int64_t)b * a.stride;
    const float2* __restrict__ srv = a.srv + (int64_t)b * a.stride;
    float2* __restrict__ out = a.out + (int64_t)b * a.out_stride;
    const int T = a.T;
    const int64_t nsteps = a.n - T;   // k = 0..nsteps-1 (may be <= 0)
    // this wavefront's taps: oq .. oq + Tq - 1 (an even split: 64 (TPL - 1) <= Tq <= 64 TPL for every wavefront)
    int Tq = T, oq = 0;
    if (NW > 1) {
        const int lo = T / NW, extra = T % NW;
        Tq = lo + (wave_id < extra ? 1 : 0);
        oq = wave_id * lo + (wave_id < extra ? wave_id : extra);
    }
    const int A = Tq - 64 * (TPL - 1);                  // 0 <= A <= 64 lanes with TPL taps
    const bool full = lane < A;
    const int i0 = full ? lane * TPL : A * TPL + (lane - A) * (TPL - 1);
    const int ntap = full ? TPL : TPL - 1;
    auto team_fence = [&]() { if (NW > 1) __syncthreads(); else wave_lds_fence(); };

    v2f w[TPL];
#pragma unroll
    for (int t = 0; t < TPL; ++t) {
        const float2 w0 = (a.taps_in && t < ntap) ? a.taps_in[(int64_t)b * T + oq + i0 + t] : make_float2(0.f, 0.f);
        w[t] = v2f{w0.x, w0.y};
    }
    // out[0:L] = 0 and out[n-peek:] = 0 (:231)
    for (int64_t i = tid; i < a.n; i += NT)
        if (i < a.L || i >= a.L + (nsteps > 0 ? nsteps : 0)) out[i] = make_float2(0.f, 0.f);

    for (int64_t k0 = 0; k0 < nsteps; k0 += KT) {
        const int64_t rem = nsteps - k0;
        const int cnt = rem < KT ? (int)rem : KT;
        team_fence();
        // window element x <-> ref[k0 + 1 - (WIN - T) + x];  u_kk[i] = Rw[kk + WIN - 1 - i]
        const int64_t base = k0 + 1 - (WIN - T);
        for (int x = tid; x < KT + WIN; x += NT) {
            const int64_t idx = base + x;
            Rw[x] = (idx >= 0 && idx < a.n) ? ref[idx] : make_float2(0.f, 0.f);
        }
        for (int x = tid; x < cnt; x += NT) D[x] = srv[k0 + x + a.L];
        team_fence();
        // u^H u is summed at the start of every staged window and then slid IN DOUBLE PRECISION:
        //   E(k+1) = E(k) + |ref[T+k+1]|^2 - |ref[k+1]|^2      (window element kk+WIN enters, kk+WIN-T leaves)
        // The reference re-sums at every step (:213), so its value never carries cancellation error.  A float32
        // slide does when the reference level falls sharply inside the tap window (a 50 dB drop leaves an energy
        // that is mostly rounding error, mu / en then blows up); the squares of float32 samples are exact in
        // double and the running sum is good to 1e-16 of the largest energy seen, so the slid value equals the
        // reference's per-step sum to float32 accuracy whatever the input does.
        // The slide does not depend on the taps, so it is not part of the recursion: all KT energies of the window are made
        // here, 64 lanes wide (a run of KT / 64 steps per lane, one exclusive scan of the lanes' totals in double), and the
        // step loop reads its step size mu / (u^H u) from LDS -- one broadcast ds_read_b32 where the sequential slide
        // was two ds_read_b64, four conversions, four f64 FMAs, a conversion back and the reciprocal (12 of ~105
        // instructions per step at T = 1034).
        v2f P[TPL];                                    // the lane's samples, rotating: tap t of step j is P[(t - j) mod TPL]
        {
            double e0 = 0.0;
#pragma unroll
            for (int t = 0; t < TPL; ++t) {
                float2 v = Rw[WIN - 1 - (oq + i0 + t)];
                if (t >= ntap) v = make_float2(0.f, 0.f);
                if (NW == 1) e0 = fma((double)v.x, (double)v.x, fma((double)v.y, (double)v.y, e0));
                P[t] = v2f{v.x, v.y};
            }
            if (NW == 1 || wave_id == 0) {
                if (NW > 1) {                          // the energy of ALL T taps
                    for (int i = lane; i < T; i += 64) {
                        const float2 v = Rw[WIN - 1 - i];
                        e0 = fma((double)v.x, (double)v.x, fma((double)v.y, (double)v.y, e0));
                    }
                }
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) e0 += __shfl_xor(e0, m, 64);
                // E(kk) = E(0) + sum_{m < kk} delta(m), delta(m) = |Rw[m + WIN]|^2 - |Rw[m + WIN - T]|^2
                const int Q = KT >> 6;                 // steps per lane (KT is a multiple of 64)
                double run = 0.0;
                for (int q = 0; q < Q; ++q) {
                    const int m = lane * Q + q;
                    const float2 vi = Rw[m + WIN], vo = Rw[m + WIN - T];
                    run = fma((double)vi.x, (double)vi.x, run);
                    run = fma((double)vi.y, (double)vi.y, run);
                    run = fma(-(double)vo.x, (double)vo.x, run);
                    run = fma(-(double)vo.y, (double)vo.y, run);
                }
                double incl = run;                     // inclusive scan of the lanes' totals
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const double up = __shfl_up(incl, d, 64);
                    if (lane >= d) incl += up;
                }
                double e = e0 + (incl - run);          // E at the lane's first step
                for (int q = 0; q < Q; ++q) {
                    const int m = lane * Q + q;
                    Sa[m] = a.mu * __builtin_amdgcn_rcpf((float)e);       // hardware reciprocal (1 ulp): the step size is a tuning
                    const float2 vi = Rw[m + WIN], vo = Rw[m + WIN - T];  // constant, its last bit is not the reference's either
                    e = fma((double)vi.x, (double)vi.x, e);
                    e = fma((double)vi.y, (double)vi.y, e);
                    e = fma(-(double)vo.x, (double)vo.x, e);
                    e = fma(-(double)vo.y, (double)vo.y, e);
                }
            }
            team_fence();
        }
        for (int kk0 = 0; kk0 < cnt; kk0 += TPL) {
#pragma unroll
            for (int j = 0; j < TPL; ++j) {
                const int kk = kk0 + j;
                if (kk < cnt) {                        // uniform (no break: the unrolled body keeps its constant register indices)
                    // the sample that enters this wavefront's first tap at the next step, the surveillance sample, the step size
                    const float2 vin = Rw[kk + WIN - oq];
                    const float2 d = D[kk];
                    const float sk = Sa[kk];
                    // conj(w) . u on packed FMAs, two chains opened by a product (no zeroed accumulators to set up)
                    v2f acc0 = pk_cmul_conj(w[0], P[(0 - j + TPL) % TPL]);
                    v2f acc1 = TPL > 1 ? pk_cmul_conj(w[1 % TPL], P[(1 - j + TPL) % TPL]) : v2f{0.f, 0.f};
#pragma unroll
                    for (int t = 2; t < TPL; ++t) pk_cmac_conj((t & 1) ? acc1 : acc0, w[t], P[(t - j + TPL) % TPL]);
                    const v2f ysum = acc0 + acc1;
                    float yr = ysum.x, yi = ysum.y;
                    wave_allsum2(yr, yi);
                    if (NW > 1) {
                        // partial sums of the NW tap ranges: slots of this step's parity (the other parity is still being read
                        // by a wavefront that has not left the previous step), one barrier, the same order of addition everywhere
                        float2* Sk = S + (kk & 1) * NW;
                        if (lane == 0) Sk[wave_id] = make_float2(yr, yi);
                        __syncthreads();
                        float sr = 0.f, si = 0.f;
#pragma unroll
                        for (int q = 0; q < NW; ++q) { const float2 pq = Sk[q]; sr += pq.x; si += pq.y; }
                        yr = sr;
                        yi = si;
                    }
                    const float er = d.x - yr, ei = d.y - yi;
                    const float s = sk;                 // mu / (u^H u)
                    const v2f c = {er * s, ei * s};
#pragma unroll
                    for (int t = 0; t < TPL; ++t) pk_cmac_bconj(w[t], P[(t - j + TPL) % TPL], c);
                    if (lane == 0 && (NW == 1 || wave_id == 0)) D[kk] = make_float2(er, ei);   // every wavefront read d before the barrier
                    // shift: a lane's last VALID sample (tap TPL-1, or TPL-2 in a short lane) goes to the next lane's
                    // tap 0, the new sample to lane 0's; the slot of the old last tap, P[TPL-1-j], is tap 0 of step j+1
                    v2f carry = P[TPL - 1 - j];
                    if (TPL > 1) {
                        const v2f prev = P[(2 * TPL - 2 - j) % TPL];      // tap TPL-2 of this step
                        carry = full ? carry : prev;
                    }
                    P[TPL - 1 - j] = v2f{wave_shr1(vin.x, carry.x), wave_shr1(vin.y, carry.y)};
                    if (TPL > 1) {
                        // a short lane's spare register: tap TPL-1 of step j+1 (what was tap TPL-2) stays zero
                        if (!full) P[(2 * TPL - 2 - j) % TPL] = v2f{0.f, 0.f};
                    } else if (!full) {
                        P[0] = v2f{0.f, 0.f};                               // one tap per lane: lanes beyond T hold nothing
                    }
                }
            }
        }
        team_fence();
        for (int x = tid; x < cnt; x += NT) out[a.L + k0 + x] = D[x];
    }
    if (a.taps_out) {
#pragma unroll
        for (int t = 0; t < TPL; ++t)
            if (t < ntap) a.taps_out[(int64_t)b * T + oq + i0 + t] = make_float2(w[t].x, w[t].y);
    }
}


// ---- any filter length (round 5) ---------------------------------------------------------------------------------------
// Beyond the 8192 taps the register-resident kernels hold, the reference (clutter_removal.py:189-249 takes any length)
// is served by the plain form: one workgroup of 1024 threads per stream, the taps in a global workspace (thread t owns
// taps t, t + 1024, ...: nobody else ever touches them), per step a block-wide reduction of conj(w).u and u^H u in
// double and the update.  Two workgroup barriers and ~2 T / 1024 global reads per thread and step: microseconds per step
// -- a fallback that works at any length, not a fast path.
#define NLG_THREADS 1024
__device__ __forceinline__ double nlg_wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__global__ __launch_bounds__(NLG_THREADS) void nlms_generic_kernel(NlmsArgs a, float2* __restrict__ wbuf) {
    __shared__ double red[2][3][NLG_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const float2* __restrict__ ref = a.ref + (int64_t)b * a.stride;
    const float2* __restrict__ srv = a.srv + (int64_t)b * a.stride;
    float2* __restrict__ out = a.out + (int64_t)b * a.out_stride;
    float2* __restrict__ w = wbuf + (int64_t)b * a.T;
    const int T = a.T;
    const int64_t nsteps = a.n - T;
    for (int i = tid; i < T; i += NLG_THREADS) w[i] = a.taps_in ? a.taps_in[(int64_t)b * T + i] : make_float2(0.f, 0.f);
    for (int64_t i = tid; i < a.n; i += NLG_THREADS)
        if (i < a.L || i >= a.L + (nsteps > 0 ? nsteps : 0)) out[i] = make_float2(0.f, 0.f);      // :231
    for (int64_t k = 0; k < nsteps; ++k) {
        const float2* top = ref + T + k;                          // u[i] = top[-i]   (:211-215)
        double yr = 0.0, yi = 0.0, en = 0.0;
        for (int i = tid; i < T; i += NLG_THREADS) {
            const float2 u = top[-i], wv = w[i];
            yr += (double)(wv.x * u.x + wv.y * u.y);             // conj(w) u
            yi += (double)(wv.x * u.y - wv.y * u.x);
            en += (double)u.x * (double)u.x + (double)u.y * (double)u.y;
        }
        yr = nlg_wave_sum(yr);
        yi = nlg_wave_sum(yi);
        en = nlg_wave_sum(en);
        const int par = (int)(k & 1);                             // two slot sets: a fast wavefront may already write step k + 1's
        if (lane == 0) {
            red[par][0][wave] = yr;
            red[par][1][wave] = yi;
            red[par][2][wave] = en;
        }
        __syncthreads();
        double sr = 0.0, si = 0.0, se = 0.0;
#pragma unroll
        for (int q = 0; q < NLG_THREADS / 64; ++q) {              // the same order in every thread: one error sample for all
            sr += red[par][0][q];
            si += red[par][1][q];
            se += red[par][2][q];
        }
        const float2 d = srv[k + a.L];
        const float er = d.x - (float)sr, ei = d.y - (float)si;
        const float enf = (float)se;
        for (int i = tid; i < T; i += NLG_THREADS) {
            const float2 u = top[-i];
            // w += (mu u) conj(e) / (u^H u), float32 as the reference's complex64 arithmetic (:214)
            const float mx = a.mu * u.x, my = a.mu * u.y;
            float2 wv = w[i];
            wv.x += (mx * er + my * ei) / enf;
            wv.y += (my * er - mx * ei) / enf;
            w[i] = wv;
        }
        if (tid == 0) out[a.L + k] = make_float2(er, ei);
        // no second barrier: the next step's partial sums go to the other slot set, and a thread rewrites only its own taps
    }
    if (a.taps_out)
        for (int i = tid; i < T; i += NLG_THREADS) a.taps_out[(int64_t)b * T + i] = w[i];
}

extern "C" int prc_nlms_execute(const void* ref, const void* srv, int64_t n, int64_t stride,
                                int32_t filter_len, int32_t peek, float mu, const void* taps_in,
                                void* out, int64_t out_stride, void* taps_out, int32_t nstreams,
                                void* stream) {
    PRC_RANGE("prc_nlms_execute");
    PRC_REQUIRE(ref && srv && out, PRC_EINVAL, "prc_nlms_execute: null argument");
    PRC_REQUIRE(n > 0 && filter_len > 0 && peek >= 0 && nstreams > 0, PRC_EINVAL,
                "prc_nlms_execute: non-positive size");
    PRC_REQUIRE(stride >= n && out_stride >= n, PRC_ESHAPE, "prc_nlms_execute: stride shorter than n");
    const int T = filter_len + peek;
    if (T > 8192) {
        // the plain form for any length.  Its tap workspace is a grow-only block per host thread and (device, stream),
        // kept between calls: nothing synchronises in the steady state and the call can be captured in a graph once the
        // block exists (it used to hipMalloc / synchronise / hipFree per call).  Calls on one stream are
        // ordered by the stream; only GROWING a block waits for that stream first (its kernels may still read the old one).
        NlmsArgs g;
        g.ref = (const float2*)ref;
        g.srv = (const float2*)srv;
        g.taps_in = (const float2*)taps_in;
        g.out = (float2*)out;
        g.taps_out = (float2*)taps_out;
        g.n = n;
        g.stride = stride;
        g.out_stride = out_stride;
        g.L = filter_len;
        g.peek = peek;
        g.T = T;
        g.mu = mu;
        g.nstreams = nstreams;
        g.kt = 0;
        struct Scratch { float2* p = nullptr; size_t cap = 0; int dev = -1; hipStream_t stream = nullptr; };
        static thread_local std::vector<Scratch> pool;
        int dev = 0;
        PRC_HIP(hipGetDevice(&dev));
        const size_t need = sizeof(float2) * (size_t)nstreams * T;
        Scratch* hit = nullptr;
        for (Scratch& c : pool)
            if (c.dev == dev && c.stream == (hipStream_t)stream) { hit = &c; break; }
        if (!hit) {
            size_t mine = 0;
            for (const Scratch& c : pool) mine += c.dev == dev;
            if (mine >= 8) {                                  // a host cycling through streams: start over on this device
                PRC_HIP(hipDeviceSynchronize());
                std::vector<Scratch> keep;
                for (Scratch& c : pool) {
                    if (c.dev == dev) (void)hipFree(c.p);
                    else keep.push_back(c);
                }
                pool.swap(keep);
            }
            pool.push_back(Scratch());
            hit = &pool.back();
            hit->dev = dev;
            hit->stream = (hipStream_t)stream;
        }
        if (hit->cap < need) {
            if (hit->p) {
                PRC_HIP(hipStreamSynchronize((hipStream_t)stream));
                (void)hipFree(hit->p);
                hit->p = nullptr;
                hit->cap = 0;
            }
            const hipError_t me = hipMalloc(&hit->p, need);
            if (me != hipSuccess) {
                (void)hipGetLastError();                      // the failed allocation must not stay behind as a sticky error
                hit->p = nullptr;
                prc_set_error("prc_nlms_execute: %zu bytes of tap workspace for %d streams of %d taps: %s", need, nstreams, T,
                              hipGetErrorString(me));
                return PRC_EHIP;
            }
            hit->cap = need;
        }
        hipLaunchKernelGGL(nlms_generic_kernel, dim3(nstreams), dim3(NLG_THREADS), 0, (hipStream_t)stream, g, hit->p);
        PRC_LAUNCH_CHECK();
        return PRC_OK;
    }
    int nwave = T <= 2048 ? 1 : (T <= 4096 ? 2 : 4);             // wavefronts per stream
    // latency experiments (tools/nlms_waves_probe.py): PRC_OPT_NLMS_WAVES = 2 or 4 splits a filter that fits one wavefront
    // over 2 or 4 as well (honoured for the config-3 filter length, T = 1034, which is what the multi-wavefront
    // instantiations below cover)
    {
        const int v = (int)prc_opt(PRC_OPT_NLMS_WAVES);
        if ((v == 2 || v == 4) && v > nwave && T == 1034) nwave = v;
    }
    const int tpl = ((T + nwave - 1) / nwave + 63) / 64;          // taps per lane: <= 32
    NlmsArgs a;
    a.ref = (const float2*)ref;
    a.srv = (const float2*)srv;
    a.taps_in = (const float2*)taps_in;
    a.out = (float2*)out;
    a.taps_out = (float2*)taps_out;
    a.n = n;
    a.stride = stride;
    a.out_stride = out_stride;
    a.L = filter_len;
    a.peek = peek;
    a.T = T;
    a.mu = mu;
    a.nstreams = nstreams;
    // wavefronts per workgroup = per CU: one per SIMD up to 1024 streams, then two, then three (register file permitting)
    const int maxw = tpl <= 17 ? 12 : (tpl <= 24 ? 8 : 4);
    // measured step time of a SIMD holding 1 / 2 / 3 of these wavefronts, relative: 1.0 / 1.68 / 2.44 (MI355X, round 3:
    // 274 / 459 / 669 ns per step at T = 1034)
    int nw = 4;
    double best = 1e30;
    for (int w = 4; w <= maxw; w += 4) {
        const double rounds = (double)ceil_div64(ceil_div64(nstreams, w), 256);
        const double cost = rounds * (w == 4 ? 1.0 : (w == 8 ? 1.68 : 2.44));
        if (cost < best) { best = cost; nw = w; }
    }
    // A/B runs (PRC_OPT_NLMS_WG_WAVES): a forced count; 16 = four wavefronts per SIMD, instantiated for the config-3
    // filter (17 taps per lane, 118 VGPRs) -- its LDS window shrinks to 64 steps (16 streams share the CU's 160 KB)
    const int forced = (int)prc_opt(PRC_OPT_NLMS_WG_WAVES);
    if (forced == 16 && tpl == 17) nw = 16;
    else if (forced >= 4 && forced <= maxw) nw = forced;
    const size_t lds_cu = 160 * 1024;
    int dev = 0;
    PRC_HIP(hipGetDevice(&dev));
    if (nwave > 1) {
        // one stream per workgroup of nwave wavefronts; two workgroups per CU where the window allows it
        int kt = 1024;
        const size_t win = (size_t)64 * tpl * nwave;
        auto team_lds = [&](int k) { return (2 * (size_t)k + win + k / 2 + 2 * nwave) * sizeof(float2); };
        while (kt > 128 && team_lds(kt) > 80 * 1024) kt >>= 1;
        a.kt = kt;
        const size_t lds = team_lds(kt);
#define PRC_NLMS_TEAM_CASE(G)                                                                   \
    case G: {                                                                                   \
        const void* fn = nwave == 2 ? reinterpret_cast<const void*>(&nlms_kernel<G, 1, 2>)      \
                                    : reinterpret_cast<const void*>(&nlms_kernel<G, 1, 4>);     \
        { int rc_ = prc_lds_optin(fn, (int)lds_cu); if (rc_) return rc_; }                      \
        if (nwave == 2) hipLaunchKernelGGL((nlms_kernel<G, 1, 2>), dim3(nstreams), dim3(128), lds, (hipStream_t)stream, a); \
        else hipLaunchKernelGGL((nlms_kernel<G, 1, 4>), dim3(nstreams), dim3(256), lds, (hipStream_t)stream, a); \
        PRC_LAUNCH_CHECK();                                                                     \
        return PRC_OK;                                                                          \
    }
        switch (tpl) {
            PRC_NLMS_TEAM_CASE(5) PRC_NLMS_TEAM_CASE(9)          // T = 1034 on four / two wavefronts (the experiment knob only)
            PRC_NLMS_TEAM_CASE(17) PRC_NLMS_TEAM_CASE(18) PRC_NLMS_TEAM_CASE(19) PRC_NLMS_TEAM_CASE(20)
            PRC_NLMS_TEAM_CASE(21) PRC_NLMS_TEAM_CASE(22) PRC_NLMS_TEAM_CASE(23) PRC_NLMS_TEAM_CASE(24)
            PRC_NLMS_TEAM_CASE(25) PRC_NLMS_TEAM_CASE(26) PRC_NLMS_TEAM_CASE(27) PRC_NLMS_TEAM_CASE(28)
            PRC_NLMS_TEAM_CASE(29) PRC_NLMS_TEAM_CASE(30) PRC_NLMS_TEAM_CASE(31) PRC_NLMS_TEAM_CASE(32)
            default: break;
        }
#undef PRC_NLMS_TEAM_CASE
        return PRC_EUNSUPPORTED;
    }
    // steps per staged window: a multiple of 64 (the energy prepass gives every lane a run of kt / 64 steps); per stream
    // kt + 64 tpl reference samples, kt surveillance / error samples (complex64) and kt step sizes (float32)
    auto wave_lds = [&](int k) { return (size_t)nw * (2 * (size_t)k + 64 * tpl + k / 2) * sizeof(float2); };
    int kt = 1024;
    for (const int k : {1024, 768, 512, 384, 256, 192, 128, 64}) { kt = k; if (wave_lds(k) <= lds_cu) break; }
    a.kt = kt;
    size_t lds = wave_lds(kt);
    PRC_REQUIRE(lds <= lds_cu, PRC_EUNSUPPORTED, "prc_nlms_execute: %d taps do not fit the LDS window", T);
    if (lds < 84 * 1024) lds = 84 * 1024;              // more than half a CU's LDS: one workgroup per CU
    const int grid = (int)ceil_div64(nstreams, nw);
    // one instantiation per taps-per-lane count: the kernel masks only the LAST 64-tap group against T, so the
    // group count must be exact (a coarser bucket list once left whole groups beyond T unmasked)
#define PRC_NLMS_CASE(G, W)                                                                     \
    case G: {                                                                                   \
        static bool attr_done[16] = {false};      /* the LDS opt-in is per device and per instantiation, not per launch */ \
        if (dev < 0 || dev >= 16 || !attr_done[dev]) {                                          \
            PRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&nlms_kernel<G, W>),      \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cu)); \
            if (dev >= 0 && dev < 16) attr_done[dev] = true;                                    \
        }                                                                                       \
        hipLaunchKernelGGL((nlms_kernel<G, W>), dim3(grid), dim3(64 * nw), lds, (hipStream_t)stream, a); \
        PRC_LAUNCH_CHECK();                                                                     \
        return PRC_OK;                                                                          \
    }
    if (nw == 16) {
        switch (tpl) {
            PRC_NLMS_CASE(17, 16)
            default: break;
        }
    }
    switch (tpl) {
        PRC_NLMS_CASE(1, 12) PRC_NLMS_CASE(2, 12) PRC_NLMS_CASE(3, 12) PRC_NLMS_CASE(4, 12)
        PRC_NLMS_CASE(5, 12) PRC_NLMS_CASE(6, 12) PRC_NLMS_CASE(7, 12) PRC_NLMS_CASE(8, 12)
        PRC_NLMS_CASE(9, 12) PRC_NLMS_CASE(10, 12) PRC_NLMS_CASE(11, 12) PRC_NLMS_CASE(12, 12)
        PRC_NLMS_CASE(13, 12) PRC_NLMS_CASE(14, 12) PRC_NLMS_CASE(15, 12) PRC_NLMS_CASE(16, 12)
        PRC_NLMS_CASE(17, 12) PRC_NLMS_CASE(18, 8) PRC_NLMS_CASE(19, 8) PRC_NLMS_CASE(20, 8)
        PRC_NLMS_CASE(21, 8) PRC_NLMS_CASE(22, 8) PRC_NLMS_CASE(23, 8) PRC_NLMS_CASE(24, 8)
        PRC_NLMS_CASE(25, 4) PRC_NLMS_CASE(26, 4) PRC_NLMS_CASE(27, 4) PRC_NLMS_CASE(28, 4)
        PRC_NLMS_CASE(29, 4) PRC_NLMS_CASE(30, 4) PRC_NLMS_CASE(31, 4) PRC_NLMS_CASE(32, 4)
        default: break;
    }
#undef PRC_NLMS_CASE
    return PRC_EUNSUPPORTED;
}
