// Block least-squares clutter cancellers (time-domain kernels).
//
// Replaces clutter_removal.py: LS_Filter_Toeplitz (:109-160), LS_Filter_Multiple (:162-187),
// LS_Filter (:6-56) and the helpers they call, signal_utils.xcorr (:29-32) and
// frequency_shift (:24-27).  Three kernels per Doppler bin:
//   1. corr_partial_kernel : per-tile lag products   sum_m r[m] conj(r[m+k]),  sum_m r[m] conj(s[m+k])
//      (scipy.signal.correlate 'valid' at :142-147), r = peek-rotated, phase-rotated reference
//      generated on the fly while staging LDS (the reference materialises roll() and
//      frequency_shift() arrays);
//   2. levinson_wave_kernel: fp64 reduction of the partials + Hermitian-Toeplitz Levinson solve on one
//      wavefront (scipy.linalg.solve_toeplitz at :150, complex128); the multi-bin chain on long
//      blocks replaces it by ONE Levinson-Durbin per block + parallel per-bin solves (see below);
//   3. fir_subtract_kernel : out = s - conv(r, w)[0:N] (np.convolve at :153-155), complex64.
// With circular=1 the same kernels wrap indices modulo N, which is LS_Filter's circulant data
// matrix: A^H A is exactly the circular-autocorrelation Toeplitz matrix, so no N x T matrix and
// no dense GEMM is ever formed.
#include "ls_internal.h"
#include <math.h>
#include <vector>

#define LSC_TILE 1024
#define LSC_TILES_PER_BLOCK 4
#define LSC_BLK (LSC_TILE * LSC_TILES_PER_BLOCK)
#define LS_THREADS 256

struct CorrArgs {
    const float2* p_src;    // P source (reference for LS)
    const float2* s1_src;   // first S source
    const float2* s2_src;   // second S source (DUAL) or nullptr
    int64_t p_stride, s1_stride, s2_stride;  // elements between batch items
    int64_t n;
    int32_t nlags;          // lags 0..nlags-1
    int32_t peek_p, peek_s1, peek_s2;  // logical index m reads src[(m + peek) mod n]
    int32_t rot_p, rot_s1, rot_s2;     // apply the phase ramp (on the source index) or not
    int32_t circular;       // S index wraps modulo n, else zero beyond n
    PhaseRamp pr;
    float2* partial;        // [batch][nblk][nS][nlags]
    int32_t nblk;
};

__device__ __forceinline__ float2 load_rot(const float2* __restrict__ src, int64_t m, int64_t n,
                                           int peek, int rot, const PhaseRamp& pr) {
    int64_t idx = m + peek;
    if (idx >= n) idx -= n;
    float2 v = src[idx];
    if (rot) v = cmul(v, phase_rot(pr, idx));
    return v;
}

template <int NLG, bool DUAL>
__global__ __launch_bounds__(LS_THREADS) void corr_partial_kernel(CorrArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* P = reinterpret_cast<float2*>(smem_raw);
    float2* S1 = P + LSC_TILE;
    float2* S2 = S1 + LSC_TILE + 64 * NLG;
    float2* red = DUAL ? S2 + LSC_TILE + 64 * NLG : S2;   // 4 waves * (DUAL?2:1) * NLG * 64

    constexpr int NS = DUAL ? 2 : 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int blk = blockIdx.x, b = blockIdx.y;
    const float2* __restrict__ psrc = a.p_src + (int64_t)b * a.p_stride;
    const float2* __restrict__ s1src = a.s1_src + (int64_t)b * a.s1_stride;
    const float2* __restrict__ s2src = DUAL ? a.s2_src + (int64_t)b * a.s2_stride : nullptr;
    const int64_t m_begin = (int64_t)blk * LSC_BLK;
    int64_t m_end = m_begin + LSC_BLK;
    if (m_end > a.n) m_end = a.n;

    for (int L0 = 0; L0 < a.nlags; L0 += 64 * NLG) {
        // float32 fused multiply-adds in runs of 32 samples, the runs summed in double: a single float32 accumulator over
        // the 1024 samples a wavefront sums per partial left 8e-7 in the taps, which at 1034 taps is 2e-4 of a cleaned
        // output a hundred times below its input (round 4; the FFT kernels sum per 1024- or 4096-point transform and the
        // solve kernels add the partials in double)
        float2 acc[NS][NLG];
        double2 dacc[NS][NLG];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int g = 0; g < NLG; ++g) { acc[s][g] = make_float2(0.f, 0.f); dacc[s][g] = make_double2(0.0, 0.0); }

        for (int64_t t0 = m_begin; t0 < m_end; t0 += LSC_TILE) {
            const int64_t rem = m_end - t0;
            const int cnt = rem < LSC_TILE ? (int)rem : LSC_TILE;
            for (int i = tid; i < LSC_TILE; i += LS_THREADS) {
                float2 v = make_float2(0.f, 0.f);
                if (i < cnt) v = load_rot(psrc, t0 + i, a.n, a.peek_p, a.rot_p, a.pr);
                P[i] = v;
            }
            for (int i = tid; i < LSC_TILE + 64 * NLG; i += LS_THREADS) {
                int64_t m = t0 + i + L0;
                bool ok = true;
                if (m >= a.n) {
                    if (a.circular) m %= a.n; else ok = false;
                }
                float2 v1 = make_float2(0.f, 0.f), v2 = make_float2(0.f, 0.f);
                if (ok) {
                    v1 = load_rot(s1src, m, a.n, a.peek_s1, a.rot_s1, a.pr);
                    if (DUAL) v2 = load_rot(s2src, m, a.n, a.peek_s2, a.rot_s2, a.pr);
                }
                S1[i] = v1;
                if (DUAL) S2[i] = v2;
            }
            __syncthreads();
            const int i0 = wave * (LSC_TILE / 4);
            int i1 = i0 + LSC_TILE / 4;
            if (i1 > cnt) i1 = cnt;
            const float2* S1l = S1 + lane;
            const float2* S2l = S2 + lane;
            for (int ia = i0; ia < i1; ia += 32) {
                const int ib = ia + 32 < i1 ? ia + 32 : i1;
#pragma unroll 2
                for (int i = ia; i < ib; ++i) {
                    const float2 p = P[i];
#pragma unroll
                    for (int g = 0; g < NLG; ++g) {
                        cmac_conj(acc[0][g], p, S1l[i + 64 * g]);
                        if (DUAL) cmac_conj(acc[NS - 1][g], p, S2l[i + 64 * g]);
                    }
                }
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int g = 0; g < NLG; ++g) {
                        dacc[s][g].x += (double)acc[s][g].x;
                        dacc[s][g].y += (double)acc[s][g].y;
                        acc[s][g] = make_float2(0.f, 0.f);
                    }
            }
            __syncthreads();
        }
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int g = 0; g < NLG; ++g)
                red[((wave * NS + s) * NLG + g) * 64 + lane] = make_float2((float)dacc[s][g].x, (float)dacc[s][g].y);
        __syncthreads();
        for (int t = tid; t < NS * NLG * 64; t += LS_THREADS) {
            float2 v = red[t];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float2 u = red[w * NS * NLG * 64 + t];
                v.x += u.x;
                v.y += u.y;
            }
            const int s = t / (NLG * 64);
            const int lag = L0 + (t - s * NLG * 64);
            if (lag < a.nlags)
                a.partial[(((int64_t)b * a.nblk + blk) * NS + s) * a.nlags + lag] = v;
        }
        __syncthreads();
    }
}

static int launch_corr(const CorrArgs& a, bool dual, int nbatch, hipStream_t stream) {
    int nlg = (a.nlags + 63) / 64;
    if (nlg > 8) nlg = 8;
    dim3 grid((unsigned)a.nblk, (unsigned)nbatch);
#define PRC_CORR_CASE(G)                                                                          \
    case G: {                                                                                     \
        if (dual) {                                                                               \
            size_t lds = sizeof(float2) * (LSC_TILE + 2 * (LSC_TILE + 64 * G) + 4 * 2 * G * 64); \
            hipLaunchKernelGGL((corr_partial_kernel<G, true>), grid, dim3(LS_THREADS), lds, stream, a); \
        } else {                                                                                  \
            size_t lds = sizeof(float2) * (LSC_TILE + (LSC_TILE + 64 * G) + 4 * G * 64);         \
            hipLaunchKernelGGL((corr_partial_kernel<G, false>), grid, dim3(LS_THREADS), lds, stream, a); \
        }                                                                                         \
    } break;
    switch (nlg) {
        PRC_CORR_CASE(1)
        PRC_CORR_CASE(2)
        PRC_CORR_CASE(3)
        PRC_CORR_CASE(4)
        PRC_CORR_CASE(5)
        PRC_CORR_CASE(6)
        PRC_CORR_CASE(7)
        PRC_CORR_CASE(8)
    }
#undef PRC_CORR_CASE
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}

// ---- Levinson on ONE wavefront (no workgroup barrier on the T-step critical path) -------------
// All-lanes sum of a double by DPP: xor-1, xor-2 inside quads, half-row mirror, row mirror (each
// row of 16 lanes then holds its row sum), then the four row sums are read with v_readlane.
__device__ __forceinline__ double dpp_add_d(double v, const int ctrl_is) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    int plo, phi;
    switch (ctrl_is) {
        case 0: plo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
                phi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true); break;   // quad xor 1
        case 1: plo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true);
                phi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true); break;   // quad xor 2
        case 2: plo = __builtin_amdgcn_mov_dpp(lo, 0x141, 0xF, 0xF, true);
                phi = __builtin_amdgcn_mov_dpp(hi, 0x141, 0xF, 0xF, true); break;  // row_half_mirror
        default: plo = __builtin_amdgcn_mov_dpp(lo, 0x140, 0xF, 0xF, true);
                 phi = __builtin_amdgcn_mov_dpp(hi, 0x140, 0xF, 0xF, true); break; // row_mirror
    }
    return v + __hiloint2double(phi, plo);
}
__device__ __forceinline__ double readlane_d(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_allsum_d(double v) {
    v = dpp_add_d(v, 0);
    v = dpp_add_d(v, 1);
    v = dpp_add_d(v, 2);
    v = dpp_add_d(v, 3);
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}

// LDS: c[T], a[T], w[T] (double2) -- 48 T bytes, so filters up to 3413 taps fit the 160 KB of a CU (the 4096-point FFT
// kernels carry 3073).  Same recursion as levinson_kernel with the predictor updated IN PLACE: step m turns a[j] and
// a[m-j] into a[j] + k conj(a[m-j]) and a[m-j] + k conj(a[j]) together, one lane per pair, so no second predictor
// buffer exists and a lane's loop is half as long; the right-hand side b is read from HBM one element per step
// (round 2 held five arrays in LDS and stopped at 2047 taps).
// C_GLOBAL (3414 .. 5120 taps, round 4): the autocorrelation c[] -- written once by the prologue, read-only in the
// recursion -- lives in a global workspace (c_ws, [block][T]) instead of LDS, which then holds only the two vectors the
// recursion rewrites (32 T bytes); its reads are ordinary cached loads behind one fence.
// MODE 2 (beyond 5120 taps, round 5: the reference takes any length): the two vectors the recursion rewrites live in a
// global workspace as well (aw_ws, [block][2 T]); one wavefront per block still, its lanes' stores made visible to each
// other's loads by one device-scope fence per step (write-through L1 + invalidate).  T^2 / 128 dependent global round
// trips: seconds at 10^4 taps -- a fallback that works, not a fast path.
template <int MODE>
__global__ __launch_bounds__(64) void levinson_wave_kernel(const float2* __restrict__ partial,
                                                           int nblk, int T, double reg,
                                                           double2* __restrict__ taps_out,
                                                           double2* __restrict__ rhs_ws, double2* __restrict__ c_ws,
                                                           double2* __restrict__ aw_ws) {
    constexpr bool C_GLOBAL = MODE >= 1, A_GLOBAL = MODE == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double2* lds0 = reinterpret_cast<double2*>(smem_raw);
    const int b = blockIdx.x;
    double2* c = C_GLOBAL ? c_ws + (int64_t)b * T : lds0;
    double2* a = A_GLOBAL ? aw_ws + (int64_t)b * 2 * T : (C_GLOBAL ? lds0 : lds0 + T);
    double2* w = a + T;
    const int lane = threadIdx.x;
    const float2* part = partial + (int64_t)b * nblk * 2 * T;
    double2* __restrict__ bb = rhs_ws + (int64_t)b * T;
    for (int k = lane; k < T; k += 64) {
        double cr = 0, ci = 0, br = 0, bi = 0;
#pragma unroll 4
        for (int blk = 0; blk < nblk; ++blk) {
            const float2 u = part[((int64_t)blk * 2 + 0) * T + k];
            const float2 v = part[((int64_t)blk * 2 + 1) * T + k];
            cr += (double)u.x; ci += (double)u.y;
            br += (double)v.x; bi += (double)v.y;
        }
        if (k == 0) cr += reg;
        c[k] = make_double2(cr, -ci);
        bb[k] = make_double2(br, -bi);
        a[k] = make_double2(k == 0 ? 1.0 : 0.0, 0.0);
        w[k] = make_double2(0.0, 0.0);
    }
    if (C_GLOBAL) __threadfence();                // c[] went to memory: visible to the other lanes' loads below
    __syncthreads();                              // one wavefront: orders the LDS and the global writes above
    double err = c[0].x;
    if (lane == 0) w[0] = zdiv(bb[0], c[0]);
    if (A_GLOBAL) __threadfence();
    __syncthreads();
    for (int m = 1; m < T; ++m) {
        const double2 bm = bb[m];                  // issued before the reductions: its latency sits under them
        double2 acc = make_double2(0, 0), dot = make_double2(0, 0);
        for (int i = lane; i < m; i += 64) {
            const double2 cm = c[m - i];
            acc = zadd(acc, zmul(a[i], cm));
            dot = zadd(dot, zmul(cm, w[i]));
        }
        acc.x = wave_allsum_d(acc.x);
        acc.y = wave_allsum_d(acc.y);
        dot.x = wave_allsum_d(dot.x);
        dot.y = wave_allsum_d(dot.y);
        const double rerr = 1.0 / err;
        const double2 k = make_double2(-acc.x * rerr, -acc.y * rerr);
        err = err * (1.0 - (k.x * k.x + k.y * k.y));
        const double2 res = zsub(bm, dot);
        const double rerr2 = 1.0 / err;
        const double2 g = make_double2(res.x * rerr2, res.y * rerr2);
        __builtin_amdgcn_wave_barrier();           // every lane has read a[] and w[] for the sums
        for (int j = lane; 2 * j <= m; j += 64) {
            const int mj = m - j;
            const double2 aj = a[j];
            const double2 amj = a[mj];
            const double2 nj = zadd(aj, zmul(k, zconj(amj)));
            const double2 nmj = zadd(amj, zmul(k, zconj(aj)));
            a[j] = nj;
            w[j] = zadd(w[j], zmul(g, zconj(nmj)));
            if (mj != j) {
                a[mj] = nmj;
                w[mj] = zadd(w[mj], zmul(g, zconj(nj)));
            }
        }
        if (A_GLOBAL) __threadfence();             // this step's a[], w[] reach memory before the next step's loads
        __builtin_amdgcn_wave_barrier();
    }
    for (int k = lane; k < T; k += 64) taps_out[(int64_t)b * T + k] = w[k];
}

// ---- shared-inverse solve for the Doppler-bin chain --------------------------------------------
// LS_Filter_Multiple (clutter_removal.py:178-187) solves Toeplitz(c_f) w = b_f once per Doppler bin
// with r_f = roll(ref e^{j phi_f}, -peek).  With rho = roll(ref, -peek), theta = 2 pi f/Fs and
// gamma = e^{-j theta N} (the phase ramp restarts for the `peek` samples that wrapped around):
//     c_f[k] = e^{j theta k} ( c_0[k] + (gamma - 1) S_e[k] ),
//     S_e[k] = sum_{n >= max(N-peek,k), n-k < N-peek} rho[n] conj(rho[n-k])          (<= peek terms)
// so the T sequential Levinson-Durbin steps run ONCE per block (ls_prepare_kernel: forward
// predictor -> dense T_0^{-1} by the Trench recurrence), and every bin is a fully parallel
//     w = D T_0^{-1} D^H b_f  (+ refinement steps against the exact c_f),   D = diag(e^{j theta k})
// (ls_solve_kernel).  Checked identity by identity in tools/ls_shared_inverse_model.py.
struct LsPrepArgs {
    const float2* partial;   // [block][nblk][2][T], slot 0 = conj(autocorrelation of the bin-0 reference)
    const float2* ref;
    int64_t ref_stride;
    int64_t n;
    int32_t nblk, T, peek;
    double reg;
    double theta0;           // rotation of the reference the autocorrelation was taken with
    double2* c0;             // [block][T]
    double2* se;             // [block][T]
    double2* tinv;           // [block][T][T] dense inverse (Trench), or nullptr: Gohberg-Semencul mode ...
    double2* apred;          // ... [block][T] forward predictor a (a[0] = 1)
    double* perr;            // ... [block] final prediction error
};

__global__ __launch_bounds__(LS_THREADS) void ls_prepare_kernel(LsPrepArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double2* c = reinterpret_cast<double2*>(smem_raw);
    double2* abuf0 = c + a.T;
    double2* abuf1 = abuf0 + a.T;
    double2* scratch = abuf1 + a.T;                    // [0] = (err, which buffer holds the predictor)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, T = a.T;
    const float2* part = a.partial + (int64_t)b * a.nblk * 2 * T;
    const float2* ref = a.ref + (int64_t)b * a.ref_stride;
    const int64_t N = a.n;
    const double g0x = cos(-a.theta0 * (double)N) - 1.0, g0y = sin(-a.theta0 * (double)N);   // gamma0 - 1
    for (int k = tid; k < T; k += LS_THREADS) {
        double cr = 0, ci = 0;
#pragma unroll 4
        for (int blk = 0; blk < a.nblk; ++blk) {
            const float2 u = part[((int64_t)blk * 2 + 0) * T + k];
            cr += (double)u.x;
            ci += (double)u.y;
        }
        // S_e[k]: n wrapped (rho[n] = ref[n+peek-N]), n-k not wrapped (rho[n-k] = ref[n-k+peek])
        double sr = 0, si = 0;
        for (int64_t n = (N - a.peek > k ? N - a.peek : k); n < N; ++n) {
            if (n - k >= N - a.peek) continue;
            const float2 x = ref[n + a.peek - N], y = ref[n - k + a.peek];
            sr += (double)x.x * y.x + (double)x.y * y.y;
            si += (double)x.y * y.x - (double)x.x * y.y;
        }
        // c_0[k] = e^{-j theta0 k} c_f0[k] - (gamma0 - 1) S_e[k]
        double sn, cs;
        sincos(-a.theta0 * (double)k, &sn, &cs);
        const double2 cf0 = make_double2(cr, -ci);
        double2 c0 = zmul(make_double2(cs, sn), cf0);
        c0 = zsub(c0, zmul(make_double2(g0x, g0y), make_double2(sr, si)));
        if (k == 0) c0.x += a.reg;
        c[k] = c0;
        a.c0[(int64_t)b * T + k] = c0;
        a.se[(int64_t)b * T + k] = make_double2(sr, si);
        abuf0[k] = make_double2(k == 0 ? 1.0 : 0.0, 0.0);
        abuf1[k] = make_double2(k == 0 ? 1.0 : 0.0, 0.0);
    }
    __syncthreads();
    // Levinson-Durbin forward recursion on wavefront 0 (T sequential steps, no workgroup barrier)
    if (wave == 0) {
        double err = c[0].x;
        double2* a_old = abuf0;
        double2* a_new = abuf1;
        for (int m = 1; m < T; ++m) {
            double2 acc = make_double2(0, 0);
            for (int i = lane; i < m; i += 64) acc = zadd(acc, zmul(a_old[i], c[m - i]));
            acc.x = wave_allsum_d(acc.x);
            acc.y = wave_allsum_d(acc.y);
            const double rerr = 1.0 / err;
            const double2 k = make_double2(-acc.x * rerr, -acc.y * rerr);
            err = err * (1.0 - (k.x * k.x + k.y * k.y));
            for (int j = lane; j <= m; j += 64)
                a_new[j] = zadd(a_old[j], zmul(k, zconj(a_old[m - j])));
            __builtin_amdgcn_wave_barrier();
            double2* t = a_old;
            a_old = a_new;
            a_new = t;
        }
        if (lane == 0) scratch[0] = make_double2(err, (a_old == abuf0) ? 0.0 : 1.0);
    }
    __syncthreads();
    const double2* af = scratch[0].y != 0.0 ? abuf1 : abuf0;   // forward predictor, af[0] = 1
    if (!a.tinv) {
        // Gohberg-Semencul mode: T_0^{-1} = (1/err) [ L(a) L(a)^H - L(z) L(z)^H ] is applied from the predictor itself
        // (ls_solve_gs_kernel); nothing dense is built
        for (int k = tid; k < T; k += LS_THREADS) a.apred[(int64_t)b * T + k] = af[k];
        if (tid == 0) a.perr[b] = scratch[0].x;
        return;
    }
    const double ie = 1.0 / scratch[0].x;              // x = af / err is the first column of T_0^{-1}
    double2* tinv = a.tinv + (int64_t)b * T * T;
    // Trench recurrence along the diagonals of the lower triangle (i = j + d), mirrored by symmetry:
    //   inv[i+1][j+1] = inv[i][j] + ( x[i+1] conj(x[j+1]) - conj(x[T-1-i]) x[T-1-j] ) / x[0]
    for (int d = tid; d < T; d += LS_THREADS) {
        double2 v = zscale(af[d], ie);                 // inv[d][0] = x[d]
        tinv[(int64_t)d * T] = v;
        tinv[d] = zconj(v);
        for (int j = 0; j + 1 + d < T; ++j) {
            const int i = j + d;
            const double2 t1 = zmul(af[i + 1], zconj(af[j + 1]));
            const double2 t2 = zmul(zconj(af[T - 1 - i]), af[T - 1 - j]);
            v = zadd(v, zscale(zsub(t1, t2), ie));     // (x x^H - xr xr^H)/x0 with x = af/err, x0 = 1/err
            tinv[(int64_t)(i + 1) * T + (j + 1)] = v;
            tinv[(int64_t)(j + 1) * T + (i + 1)] = zconj(v);
        }
    }
}

struct LsSolveArgs {
    const float2* partial;   // slot 1 = conj( sum_n s~[n] conj(rho[n-k]) ), s~ = s e^{-j theta (n+peek)}
    const double2* c0;
    const double2* se;
    const double2* tinv;
    const float2* ref;       // for the <= peek wrapped samples of rho
    const float2* srv;       // current surveillance stream (input of this bin)
    int64_t ref_stride, srv_stride;
    double2* taps;           // [block][T]  w
    double2* taps_t;         // [block][T]  w~[k] = w[k] e^{-j theta k}  (what the cached FIR applies)
    int64_t n;
    int32_t nblk, T, nref, peek;
    int32_t srv_rotated;     // the stream already is s~ = s e^{-j theta (n+peek)} (bins after the first of a chain)
    double theta;            // rotation of this bin (effective float32 ramp slope)
};

#define LSS_THREADS 1024
__global__ __launch_bounds__(LSS_THREADS) void ls_solve_kernel(LsSolveArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double2* bb = reinterpret_cast<double2*>(smem_raw);   // right-hand side
    double2* cf = bb + a.T;                                // exact first column for this bin
    double2* v = cf + a.T;                                 // D^H (rhs or residual)
    double2* x = v + a.T;                                  // solution
    double2* dd = x + a.T;                                 // D[k] = e^{j theta k}
    double2* pacc = dd + a.T;                              // [parts][RP] partial row sums
    const int tid = threadIdx.x, b = blockIdx.x, T = a.T;
    // rows are spread over RP = T rounded up to a wavefront; the column range is cut into `parts`
    const int RP = (T + 63) & ~63;
    const int parts = LSS_THREADS / RP > 0 ? LSS_THREADS / RP : 1;
    const int row = tid % RP, part = tid / RP;
    const int span = (T + parts - 1) / parts;
    const int j0 = part * span, j1 = (j0 + span < T) ? j0 + span : T;
    const bool active = part < parts && row < T;
    const float2* part_sum = a.partial + (int64_t)b * a.nblk * 2 * T;
    const double2* tinv = a.tinv + (int64_t)b * T * T;
    const double gx = cos(-a.theta * (double)a.n) - 1.0, gy = sin(-a.theta * (double)a.n);
    for (int k = tid; k < T; k += LSS_THREADS) {
        double br = 0, bi = 0;
#pragma unroll 4
        for (int blk = 0; blk < a.nblk; ++blk) {
            const float2 u = part_sum[((int64_t)blk * 2 + 1) * T + k];
            br += (double)u.x;
            bi += (double)u.y;
        }
        double sn, cs;
        sincos(a.theta * (double)k, &sn, &cs);
        const double2 D = make_double2(cs, sn);
        // b_f[k] = e^{j theta k} ( B~[k] + (conj(gamma)-1) E_b[k] ),
        // E_b[k] = sum_{m=N-peek}^{N-1-k} conj(rho[m]) s~[m+k]   (k < peek; rho[m] = ref[m+peek-N])
        double2 eb = make_double2(0, 0);
        if (a.theta != 0.0 && k < a.peek) {
            const float2* rr = a.ref + (int64_t)b * a.ref_stride;
            const float2* ss = a.srv + (int64_t)b * a.srv_stride;
            // phase of s~ at sample m+k: one sincos, then a one-sample rotation per term
            double s2, c2;
            sincos(-a.theta * (double)(a.n - a.peek + k + a.peek), &s2, &c2);
            double2 ph = make_double2(c2, s2);
            sincos(-a.theta, &s2, &c2);
            const double2 dph = make_double2(c2, s2);
            for (int64_t m = a.n - a.peek; m + k < a.n; ++m) {
                const float2 r = rr[m + a.peek - a.n];
                const float2 sraw = ss[m + k];
                const double2 st = a.srv_rotated ? make_double2(sraw.x, sraw.y) : zmul(make_double2(sraw.x, sraw.y), ph);
                eb = zadd(eb, zmul(make_double2(r.x, -r.y), st));
                ph = zmul(ph, dph);
            }
        }
        const double2 rhs = zmul(D, zadd(make_double2(br, -bi), zmul(make_double2(gx, -gy), eb)));
        bb[k] = rhs;
        dd[k] = D;
        v[k] = zmul(zconj(D), rhs);
        const double2 base = zadd(a.c0[(int64_t)b * T + k],
                                  zmul(make_double2(gx, gy), a.se[(int64_t)b * T + k]));
        cf[k] = zmul(D, base);
    }
    __syncthreads();
    for (int it = 0; it <= a.nref; ++it) {
        // x (+)= D T_0^{-1} v   (T_0^{-1} is Hermitian: column `row` is read as conj(row-major [j][row]),
        // consecutive threads on consecutive addresses; 8 independent loads in flight per thread)
        if (active) {
            double2 acc = make_double2(0, 0), acc2 = make_double2(0, 0);
            const double2* col = tinv + row;
            int j = j0;
            for (; j + 8 <= j1; j += 8) {
                double2 t[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) t[q] = col[(int64_t)(j + q) * T];
#pragma unroll
                for (int q = 0; q < 8; q += 2) {
                    acc = zadd(acc, zmul(zconj(t[q]), v[j + q]));
                    acc2 = zadd(acc2, zmul(zconj(t[q + 1]), v[j + q + 1]));
                }
            }
            for (; j < j1; ++j) acc = zadd(acc, zmul(zconj(col[(int64_t)j * T]), v[j]));
            pacc[part * RP + row] = zadd(acc, acc2);
        }
        __syncthreads();
        if (tid < T) {
            double2 acc = pacc[tid];
            for (int q = 1; q < parts; ++q) acc = zadd(acc, pacc[q * RP + tid]);
            acc = zmul(dd[tid], acc);
            x[tid] = it == 0 ? acc : zadd(x[tid], acc);
        }
        __syncthreads();
        if (it == a.nref) break;
        // residual against the exact Toeplitz(c_f):  v = D^H ( b - T_f x )
        if (active) {
            double2 acc = make_double2(0, 0), acc2 = make_double2(0, 0);
            int j = j0;
            for (; j + 2 <= j1; j += 2) {
                const int d0 = row - j, d1 = row - j - 1;
                const double2 c0 = d0 >= 0 ? cf[d0] : zconj(cf[-d0]);
                const double2 c1 = d1 >= 0 ? cf[d1] : zconj(cf[-d1]);
                acc = zadd(acc, zmul(c0, x[j]));
                acc2 = zadd(acc2, zmul(c1, x[j + 1]));
            }
            for (; j < j1; ++j) {
                const int d = row - j;
                acc = zadd(acc, zmul(d >= 0 ? cf[d] : zconj(cf[-d]), x[j]));
            }
            pacc[part * RP + row] = zadd(acc, acc2);
        }
        __syncthreads();
        if (tid < T) {
            double2 acc = bb[tid];
            for (int q = 0; q < parts; ++q) acc = zsub(acc, pacc[q * RP + tid]);
            v[tid] = zmul(zconj(dd[tid]), acc);
        }
        __syncthreads();
    }
    for (int k = tid; k < T; k += LSS_THREADS) {
        a.taps[(int64_t)b * T + k] = x[k];
        a.taps_t[(int64_t)b * T + k] = zmul(x[k], zconj(dd[k]));
    }
}


// ---- the same solve with T_0^{-1} in Gohberg-Semencul form ---------------------------------------------------
// T_0^{-1} = (1/err) [ L(a) L(a)^H - L(z) L(z)^H ],  a = forward predictor (a[0] = 1), err = final prediction error,
// z = (0, conj(a[T-1]), ..., conj(a[1])),  L(v) = lower-triangular Toeplitz matrix with first column v
// (identity, the refinement behaviour and the thread mapping below are checked in tools/gs_solver_model.py).
// A mat-vec is four triangular Toeplitz products on T-vectors that live in LDS -- 2 T^2 complex MACs in double
// from 17 KB of LDS instead of T^2 MACs on a 1.1 MB matrix streamed from L2/HBM per mat-vec and block -- and the
// residual against the exact Toeplitz(c_f) is two more.  Every product runs on all 1024 threads: a thread owns four
// consecutive outputs and one slice of the lag range, slides a four-element window over a zero-padded,
// four-way de-interleaved copy of the input (consecutive lanes read consecutive 16-byte words: no bank conflict)
// and the slices are summed through LDS in a fixed order (deterministic).
struct LsGsArgs {
    const float2* partial;   // slot 1 = conj( sum_n s~[n] conj(rho[n-k]) )
    const double2* c0;
    const double2* se;
    const double2* apred;    // [block][T]
    const double* perr;      // [block]
    const float2* ref;
    const float2* srv;
    int64_t ref_stride, srv_stride;
    double2* taps;
    double2* taps_t;
    int64_t n;
    int32_t nblk, T, nref, peek, srv_rotated;
    double theta;
    int32_t G, parts, span, Q;   // output groups of 4, lag slices, lags per slice, plane length of the padded vectors
};

struct TriTerm {
    const double2* c;        // coefficient vector in LDS, already conjugated / reversed / negated: coef(d) = c[d]
    const double2* pv;       // padded, de-interleaved input vector in LDS
    int dmin;                // first lag (1 skips the diagonal)
};

__device__ __forceinline__ int gs_slot(int e, int Q) { const int ep = e + 4; return (ep & 3) * Q + (ep >> 2); }

__device__ __forceinline__ void zfma(double2& acc, double2 a, double2 b) {      // acc += a * b, four FMAs
    acc.x = fma(a.x, b.x, acc.x);
    acc.x = fma(-a.y, b.y, acc.x);
    acc.y = fma(a.x, b.y, acc.y);
    acc.y = fma(a.y, b.x, acc.y);
}

// acc[j] += sum_d c[d] in[o0 + j + d]  (UP, d <= T-1-o0)   or   in[o0 + j - d]  (DOWN, d <= min(o0+3, T-1)); the
// group's OWN lag range [dmin, rmax] is cut into `parts` equal slices and this thread takes slice `c`, so no thread
// idles on the empty half of the triangle; entries of `in` outside [0, T) are zero
template <bool UP>
__device__ __forceinline__ void tri_accumulate(double2 (&acc)[4], const TriTerm& k, int o0, int c, int parts, int T,
                                               int Q) {
    const int rmax = UP ? (T - 1 - o0) : (o0 + 3 < T - 1 ? o0 + 3 : T - 1);
    const int len = rmax - k.dmin + 1;
    if (len <= 0) return;
    const int per = (len + parts - 1) / parts;
    const int dlo = k.dmin + c * per;
    int dhi = dlo + per - 1;
    if (dhi > rmax) dhi = rmax;
    if (dlo > dhi) return;
    double2 w0, w1, w2, w3;
    if (UP) {
        w0 = k.pv[gs_slot(o0 + dlo, Q)];     w1 = k.pv[gs_slot(o0 + 1 + dlo, Q)];
        w2 = k.pv[gs_slot(o0 + 2 + dlo, Q)]; w3 = k.pv[gs_slot(o0 + 3 + dlo, Q)];
    } else {
        w0 = k.pv[gs_slot(o0 - dlo, Q)];     w1 = k.pv[gs_slot(o0 + 1 - dlo, Q)];
        w2 = k.pv[gs_slot(o0 + 2 - dlo, Q)]; w3 = k.pv[gs_slot(o0 + 3 - dlo, Q)];
    }
    for (int d = dlo; d <= dhi; ++d) {
        const double2 co = k.c[d];
        zfma(acc[0], co, w0);
        zfma(acc[1], co, w1);
        zfma(acc[2], co, w2);
        zfma(acc[3], co, w3);
        if (UP) {
            w0 = w1; w1 = w2; w2 = w3;
            w3 = k.pv[gs_slot(o0 + 4 + d, Q)];
        } else {
            w3 = w2; w2 = w1; w1 = w0;
            w0 = k.pv[gs_slot(o0 - 1 - d, Q)];
        }
    }
}

__global__ __launch_bounds__(LSS_THREADS) void ls_solve_gs_kernel(LsGsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int T = a.T, Q = a.Q, G = a.G, parts = a.parts;
    double2* A = reinterpret_cast<double2*>(smem_raw);    // a[d]                        (y: sum_d a[d] p[i-d])
    double2* Ac = A + T;                                   // conj(a[d])                  (p: sum_d conj(a[d]) v[k+d])
    double2* Ar = Ac + T;                                  // a[T-d], d >= 1              (q: sum_d a[T-d] v[k+d])
    double2* Az = Ar + T;                                  // -conj(a[T-d]), d >= 1       (y: - sum_d conj(a[T-d]) q[i-d])
    double2* CF = Az + T;                                  // exact first column c_f[d] of this bin
    double2* CFc = CF + T;                                 // conj(c_f[d]), d >= 1
    double2* bb = CFc + T;                                 // right-hand side
    double2* dd = bb + T;                                  // D[k] = e^{j theta k}
    double2* PVv = dd + T;                                 // D^H (rhs or residual), padded / de-interleaved (4 Q)
    double2* PVp = PVv + 4 * Q;
    double2* PVq = PVp + 4 * Q;
    double2* PVx = PVq + 4 * Q;                            // solution
    double2* pacc = PVx + 4 * Q;                           // [parts][4 G] slice sums
    const int tid = threadIdx.x, b = blockIdx.x;
    const int g = tid % G, c = tid / G;
    const bool worker = c < parts;
    const int o0 = 4 * g;
    const float2* part_sum = a.partial + (int64_t)b * a.nblk * 2 * T;
    const double ierr = 1.0 / a.perr[b];
    const double gx = cos(-a.theta * (double)a.n) - 1.0, gy = sin(-a.theta * (double)a.n);
    for (int i = tid; i < 16 * Q; i += LSS_THREADS) PVv[i] = make_double2(0.0, 0.0);   // the four padded vectors
    // Set-up, arranged for latency (this kernel is one workgroup per block, a few microseconds of arithmetic: what
    // it waits for is global memory): the per-wave partial sums are reduced by all threads (S slices of the slot
    // range per lag, the loads of a slice independent of each other) and the <= 2 peek wrapped samples behind
    // E_b are staged through LDS in ONE round trip instead of being walked by ten threads one sample at a time.
    const int S = (LSS_THREADS / T) < 1 ? 1 : ((LSS_THREADS / T) > 8 ? 8 : (LSS_THREADS / T));
    const int per = (a.nblk + S - 1) / S;
    if (tid < S * T) {
        const int k = tid % T, sl = tid / T;
        const int b0 = sl * per, b1 = (b0 + per < a.nblk) ? b0 + per : a.nblk;
        double br = 0, bi = 0;
#pragma unroll 8
        for (int blk = b0; blk < b1; ++blk) {
            const float2 u = part_sum[((int64_t)blk * 2 + 1) * T + k];
            br += (double)u.x;
            bi += (double)u.y;
        }
        pacc[sl * T + k] = make_double2(br, bi);
    }
    double2* stg_r = pacc + (size_t)parts * 4 * G;         // conj(ref[i]),            i < peek
    double2* stg_s = stg_r + a.peek;                       // s~[n - peek + i],        i < peek
    if (a.theta != 0.0 && tid < 2 * a.peek) {
        if (tid < a.peek) {
            const float2 r = (a.ref + (int64_t)b * a.ref_stride)[tid];
            stg_r[tid] = make_double2(r.x, -r.y);
        } else {
            const int i = tid - a.peek;
            const int64_t j = a.n - a.peek + i;            // sample index; s~[j] = s[j] e^{-j theta (j + peek)}
            const float2 sraw = (a.srv + (int64_t)b * a.srv_stride)[j];
            double2 st = make_double2(sraw.x, sraw.y);
            if (!a.srv_rotated) {
                double s2, c2;
                sincos(-a.theta * (double)(j + a.peek), &s2, &c2);
                st = zmul(st, make_double2(c2, s2));
            }
            stg_s[i] = st;
        }
    }
    __syncthreads();
    for (int k = tid; k < T; k += LSS_THREADS) {
        double br = 0, bi = 0;
        for (int sl = 0; sl < S; ++sl) { br += pacc[sl * T + k].x; bi += pacc[sl * T + k].y; }
        double sn, cs;
        sincos(a.theta * (double)k, &sn, &cs);
        const double2 D = make_double2(cs, sn);
        // b_f[k] = e^{j theta k} ( B~[k] + (conj(gamma)-1) E_b[k] ),
        // E_b[k] = sum_{i=0}^{peek-1-k} conj(ref[i]) s~[n - peek + i + k]      (see ls_solve_kernel)
        double2 eb = make_double2(0, 0);
        if (a.theta != 0.0 && k < a.peek)
            for (int i = 0; i + k < a.peek; ++i) eb = zadd(eb, zmul(stg_r[i], stg_s[i + k]));
        const double2 rhs = zmul(D, zadd(make_double2(br, -bi), zmul(make_double2(gx, -gy), eb)));
        bb[k] = rhs;
        dd[k] = D;
        PVv[gs_slot(k, Q)] = zmul(zconj(D), rhs);
        const double2 base = zadd(a.c0[(int64_t)b * T + k], zmul(make_double2(gx, gy), a.se[(int64_t)b * T + k]));
        const double2 cfk = zmul(D, base);
        CF[k] = cfk;
        CFc[k] = zconj(cfk);
        const double2 ak = a.apred[(int64_t)b * T + k];
        A[k] = ak;
        Ac[k] = zconj(ak);
        const double2 arev = k >= 1 ? a.apred[(int64_t)b * T + (T - k)] : make_double2(0.0, 0.0);   // a[T-d]
        Ar[k] = arev;
        Az[k] = make_double2(-arev.x, arev.y);
    }
    __syncthreads();
    // one product: every worker accumulates its slice, the slices are summed per output
    auto product = [&](const TriTerm& t0, bool up0, const TriTerm* t1, bool up1) -> double2 {
        if (worker) {
            double2 acc[4] = {make_double2(0, 0), make_double2(0, 0), make_double2(0, 0), make_double2(0, 0)};
            if (up0) tri_accumulate<true>(acc, t0, o0, c, parts, T, Q);
            else tri_accumulate<false>(acc, t0, o0, c, parts, T, Q);
            if (t1) {
                if (up1) tri_accumulate<true>(acc, *t1, o0, c, parts, T, Q);
                else tri_accumulate<false>(acc, *t1, o0, c, parts, T, Q);
            }
            double2* pa = pacc + (size_t)c * 4 * G + o0;
            pa[0] = acc[0]; pa[1] = acc[1]; pa[2] = acc[2]; pa[3] = acc[3];
        }
        __syncthreads();
        double2 sum = make_double2(0, 0);
        if (tid < T)
            for (int q = 0; q < parts; ++q) sum = zadd(sum, pacc[(size_t)q * 4 * G + tid]);
        return sum;                                        // caller stores, then __syncthreads()
    };
    for (int it = 0; it <= a.nref; ++it) {
        // x (+)= D T_0^{-1} v,   T_0^{-1} v = ( L(a) [L(a)^H v] - L(z) [L(z)^H v] ) / err
        {
            const TriTerm tp = {Ac, PVv, 0};                                // p[k] = sum_d conj(a[d]) v[k+d]
            const double2 pk = product(tp, true, nullptr, false);
            if (tid < T) PVp[gs_slot(tid, Q)] = pk;
            __syncthreads();
            const TriTerm tq = {Ar, PVv, 1};                                // q[k] = sum_{d>=1} a[T-d] v[k+d]
            const double2 qk = product(tq, true, nullptr, false);
            if (tid < T) PVq[gs_slot(tid, Q)] = qk;
            __syncthreads();
            const TriTerm ty0 = {A, PVp, 0};                                // sum_d a[d] p[i-d]
            const TriTerm ty1 = {Az, PVq, 1};                               // - sum_{d>=1} conj(a[T-d]) q[i-d]
            const double2 yk = product(ty0, false, &ty1, false);
            if (tid < T) {
                const double2 acc = zmul(dd[tid], zscale(yk, ierr));
                const int sl = gs_slot(tid, Q);
                PVx[sl] = it == 0 ? acc : zadd(PVx[sl], acc);
            }
            __syncthreads();
        }
        if (it == a.nref) break;
        // residual against the exact Toeplitz(c_f):  v = D^H ( b - T_f x ),  (T_f x)[i] = sum_{d<=i} c_f[d] x[i-d] +
        // sum_{d>=1} conj(c_f[d]) x[i+d]
        {
            const TriTerm tr0 = {CF, PVx, 0};
            const TriTerm tr1 = {CFc, PVx, 1};
            const double2 tx = product(tr0, false, &tr1, true);
            if (tid < T) PVv[gs_slot(tid, Q)] = zmul(zconj(dd[tid]), zsub(bb[tid], tx));
            __syncthreads();
        }
    }
    for (int k = tid; k < T; k += LSS_THREADS) {
        const double2 xk = PVx[gs_slot(k, Q)];
        a.taps[(int64_t)b * T + k] = xk;
        a.taps_t[(int64_t)b * T + k] = zmul(xk, zconj(dd[k]));
    }
}

// ---- FIR apply: out[n] = s[n] - sum_k w[k] r[n-k] ----------------------------------------
#define FIR_OPT 4
#define FIR_SPAN (LS_THREADS * FIR_OPT)

struct FirArgs {
    const float2* ref;
    const float2* srv;
    float2* out;
    const double2* taps;   // [batch][T] complex128
    int64_t ref_stride, srv_stride, out_stride;
    int64_t n;
    int32_t T, peek, circular, rot;
    int32_t tile;          // taps per LDS tile (= T when the whole filter and its window fit the CU's LDS)
    PhaseRamp pr;
};

__global__ __launch_bounds__(LS_THREADS) void fir_subtract_kernel(FirArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* W = reinterpret_cast<float2*>(smem_raw);   // tile
    float2* Rt = W + a.tile;                             // FIR_SPAN + tile - 1
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int64_t n0 = (int64_t)blockIdx.x * FIR_SPAN;
    const float2* __restrict__ ref = a.ref + (int64_t)b * a.ref_stride;
    const float2* __restrict__ srv = a.srv + (int64_t)b * a.srv_stride;
    const double2* __restrict__ taps = a.taps + (int64_t)b * a.T;
    // Double-precision accumulation.  A float32 accumulator over a long filter (1034 taps at config 3) loses 4e-6 of
    // the INPUT level, which is 1e-3 of the hundred times smaller cleaned output (the reference's own complex64 matrix
    // product has that error; the FFT kernels do not); float32 blocks of 16 summed in double still left 2.5e-4.  This is the
    // fallback kernel (method = 1, or more taps than the FFT kernels carry): the products of the float32 taps and samples
    // are formed and summed in double (fp64 FMA, half the fp32 rate) and the 1e-4 bar holds on the output (round 4).
    double2 dacc[FIR_OPT];
#pragma unroll
    for (int o = 0; o < FIR_OPT; ++o) dacc[o] = make_double2(0.0, 0.0);
    // taps k0 .. k0 + tk - 1 per pass (one pass up to ~9000 taps; longer filters -- the reference takes any -- in tiles):
    // Rt[i] is reference sample n0 - k0 - halo + i, so tap k0 + k of output n0 + tid + o LS_THREADS reads Rl[o LS_THREADS - k]
    for (int k0 = 0; k0 < a.T; k0 += a.tile) {
    const int tk = a.T - k0 < a.tile ? a.T - k0 : a.tile;
    if (k0) __syncthreads();                            // the previous tile's window is dead
    for (int k = tid; k < tk; k += LS_THREADS) {
        const double2 t = taps[k0 + k];
        W[k] = make_float2((float)t.x, (float)t.y);
    }
    const int halo = tk - 1;
    for (int i = tid; i < FIR_SPAN + halo; i += LS_THREADS) {
        int64_t m = n0 - k0 - halo + i;
        float2 v = make_float2(0.f, 0.f);
        bool ok = m < a.n;
        if (m < 0) {
            if (a.circular) { m %= a.n; if (m < 0) m += a.n; } else ok = false;
        }
        if (ok) v = load_rot(ref, m, a.n, a.peek, a.rot, a.pr);
        Rt[i] = v;
    }
    __syncthreads();
    const float2* Rl = Rt + tid + halo;
#pragma unroll 2
    for (int k = 0; k < tk; ++k) {
        const float2 wf = W[k];
        const double wx = (double)wf.x, wy = (double)wf.y;
#pragma unroll
        for (int o = 0; o < FIR_OPT; ++o) {
            const float2 rf = Rl[o * LS_THREADS - k];
            const double rx = (double)rf.x, ry = (double)rf.y;
            dacc[o].x = fma(wx, rx, dacc[o].x);
            dacc[o].x = fma(-wy, ry, dacc[o].x);
            dacc[o].y = fma(wx, ry, dacc[o].y);
            dacc[o].y = fma(wy, rx, dacc[o].y);
        }
    }
    }
#pragma unroll
    for (int o = 0; o < FIR_OPT; ++o) {
        const int64_t n = n0 + tid + o * LS_THREADS;
        if (n < a.n) {
            const float2 s = srv[n];
            a.out[(int64_t)b * a.out_stride + n] = make_float2((float)((double)s.x - dacc[o].x), (float)((double)s.y - dacc[o].y));
        }
    }
}

// ---- plan -------------------------------------------------------------------------------
struct prc_ls_plan {
    prc_ls_desc desc;
    int T;
    int nblk;          // partial-sum slots per block (tiles of the direct kernel / waves of the FFT kernel)
    int method;        // 1 time-domain, 2 FFT
    int fft_waves;     // waves (1024-point kernels) or teams (4096-point kernels) per block in the FFT correlation kernel
    bool team = false; // the 4096-point team kernels of ls_fft_team.hip (770 .. 3073 taps, or method 4)
    bool team_chain = false;   // cached-spectrum chain on the 4096-point transform (ls_fft_team_cached.hip)
    int team_piece = 0;        // its samples per piece, fixed here once (PRC_OPT_LS_TEAM_ALIGN at plan creation)
    float2* d_partial = nullptr;
    double2* d_taps = nullptr;
    double2* d_rhs = nullptr;      // right-hand sides of the per-bin Levinson solve, [block][T] (one element read per step)
    double2* d_cws = nullptr;      // autocorrelations of that solve beyond 3413 taps, [block][T] (levinson_wave_kernel<1>, <2>)
    double2* d_aws = nullptr;      // predictor and taps of that solve beyond 5120 taps, [block][2 T] (levinson_wave_kernel<2>)
    float2* d_tmp[2] = {nullptr, nullptr};
    // shared-inverse path (non-circular FFT chain): c_0, S_e, dense T_0^{-1} per block
    double2* d_c0 = nullptr;
    double2* d_se = nullptr;
    double2* d_tinv = nullptr;      // dense T_0^{-1} per block (Trench) -- only when the Gohberg-Semencul form does not fit LDS
    double2* d_apred = nullptr;     // Gohberg-Semencul form: forward predictor per block ...
    double* d_perr = nullptr;       // ... and its final prediction error
    bool chain = false;             // shared-inverse chain available (non-circular 1024-point FFT kernels)
    int gs_G = 0, gs_parts = 0, gs_span = 0, gs_Q = 0;
    size_t gs_lds = 0;
    double2* d_taps_t = nullptr;   // w~ per block
    float2* d_cache = nullptr;     // FFT(rho block) per piece, reused by every bin
    // optional per-kernel timing (bench.py roofline): events around every launch of one execute
    int profiling = 0;
    std::vector<hipEvent_t> ev;     // 4 per Doppler bin: before corr, after corr, after levinson, after fir
    int ev_bins = 0;
    bool last_cached = false;     // the last execute ran the cached-spectrum chain
    std::mutex mtx;
};

extern "C" int prc_ls_plan_destroy(prc_ls_plan* p) {
    if (!p) return PRC_OK;
    if (p->d_partial) (void)hipFree(p->d_partial);
    if (p->d_taps) (void)hipFree(p->d_taps);
    if (p->d_rhs) (void)hipFree(p->d_rhs);
    if (p->d_cws) (void)hipFree(p->d_cws);
    if (p->d_aws) (void)hipFree(p->d_aws);
    if (p->d_tmp[0]) (void)hipFree(p->d_tmp[0]);
    if (p->d_tmp[1]) (void)hipFree(p->d_tmp[1]);
    if (p->d_c0) (void)hipFree(p->d_c0);
    if (p->d_se) (void)hipFree(p->d_se);
    if (p->d_tinv) (void)hipFree(p->d_tinv);
    if (p->d_apred) (void)hipFree(p->d_apred);
    if (p->d_perr) (void)hipFree(p->d_perr);
    if (p->d_taps_t) (void)hipFree(p->d_taps_t);
    if (p->d_cache) (void)hipFree(p->d_cache);
    for (hipEvent_t e : p->ev) (void)hipEventDestroy(e);
    delete p;
    return PRC_OK;
}

// c, a, w in LDS up to 3413 taps (mode 0); beyond that c moves to a global workspace and a, w stay (mode 1: 5120 taps);
// beyond that all three are global (mode 2: any length, slowly)
static int levinson_mode(int T) {
    return sizeof(double2) * ((size_t)3 * T) <= 160 * 1024 ? 0 : (sizeof(double2) * ((size_t)2 * T) <= 160 * 1024 ? 1 : 2);
}
static bool levinson_c_global(int T) { return levinson_mode(T) >= 1; }
static size_t levinson_lds(int T) { const int m = levinson_mode(T); return m == 2 ? 0 : sizeof(double2) * ((size_t)(m == 1 ? 2 : 3) * T); }
// time-domain FIR: the taps of one pass and the window they meet stay within 150 KB of LDS
static int fir_tile(int T) {
    const int cap = (int)((150 * 1024 / sizeof(float2) - FIR_SPAN) / 2);
    return T < cap ? T : cap;
}
static size_t fir_lds(int T) { const size_t t = (size_t)fir_tile(T); return sizeof(float2) * (t + FIR_SPAN + t - 1); }

// allow_cache = false: the retry after a mandatory allocation failed while the (optional) spectrum cache was held --
// the same plan without the cache, i.e. on the kernels that recompute the spectra
static int ls_plan_create_impl(prc_ls_plan** plan, const prc_ls_desc* d, bool allow_cache) {
    PRC_REQUIRE(plan && d, PRC_EINVAL, "prc_ls_plan_create: null argument");
    PRC_REQUIRE(d->n > 0 && d->filter_len > 0 && d->peek >= 0 && d->max_blocks > 0, PRC_EINVAL,
                "prc_ls_plan_create: non-positive size");
    PRC_REQUIRE(d->method >= 0 && d->method <= 4, PRC_EINVAL, "prc_ls_plan_create: method %d", d->method);
    const int T = d->filter_len + d->peek;
    PRC_REQUIRE(T < d->n, PRC_EINVAL, "prc_ls_plan_create: filter_len+peek (%d) >= n (%lld)", T,
                (long long)d->n);
    prc_ls_plan* p = new prc_ls_plan();
    p->desc = *d;
    p->T = T;
    p->method = d->method;
    if (p->method == 0) p->method = (ls_fft_supported(T) || ls_team_supported(T)) ? 2 : 1;
    if (p->method == 3) p->method = 2;          // same kernels; d->method == 3 only adds the spectrum cache
    // The cached-spectrum chain on 4096-point transforms (method 4; AUTO from 120 taps on blocks of >= 16 pieces): the
    // spectrum cache costs 32768 / (4097 - T) bytes per sample instead of 8192 / (1025 - T).  Measured on the five-bin
    // chain at 64 chunks of 1.2 M samples (tools/ls_chain_bench.py, end of round 4: packed first-bin kernel, aligned
    // pieces): T = 64 / 128 / 192 / 256 / 266 / 384 / 512 / 768 -> 1.05 / 0.92 / 0.97 / 0.92 / 0.91 / 0.90 / 0.82 / 0.64
    // of the 1024-point chain's time (round 3, when the switch sat at 250 taps: 1.06 / 1.04 / 1.01 / 0.99 / 0.96 / ...).
    const bool team_chain_ok = !d->circular && ls_fft_supported(T) && d->n >= 2 * 4096;
    if (d->method == 4 || (d->method == 0 && T >= 120 && d->n >= 16 * 4096)) {
        if (p->method == 4) p->method = 2;
        if (team_chain_ok) p->team = p->team_chain = true;
    }
    if (p->method == 2 && !ls_fft_supported(T)) {
        if (!ls_team_supported(T)) {
            prc_set_error("prc_ls_plan_create: FFT method supports at most 3073 taps, got %d", T);
            delete p;
            return PRC_EUNSUPPORTED;
        }
        p->team = true;                         // beyond the 1024-point transform: 4096-point team kernels
    }
    // Spectrum cache first: whether it can be had decides which kernels run, and with that the size of everything else.
    // A cache that cannot be allocated (or that PRC_OPT_LS_CACHE_LIMIT_MB rules out) sends the 4096-point chain back to
    // the 1024-point chain -- which recomputes the spectra when its own cache is missing too -- or, beyond 769 taps, to
    // the per-bin team kernels.
    const bool chain_ok = p->method == 2 && !d->circular;
    const int64_t limit_mb = prc_opt(PRC_OPT_LS_CACHE_LIMIT_MB);
    auto try_cache = [&](int64_t per_block) {
        const size_t bytes = sizeof(float2) * (size_t)d->max_blocks * (size_t)per_block;
        if (!allow_cache) return false;
        if (limit_mb > 0 && bytes > (size_t)limit_mb * 1048576u) return false;
        if (hipMalloc(&p->d_cache, bytes) != hipSuccess) {
            p->d_cache = nullptr;
            (void)hipGetLastError();
            return false;
        }
        return true;
    };
    if (p->team_chain) {
        p->team_piece = ls_team_piece(T, (int)prc_opt(PRC_OPT_LS_TEAM_ALIGN));
        if (!(chain_ok && d->method != 2 && try_cache(ls_team_cache_elems_per_block(d->n, p->team_piece)))) {
            p->team_chain = false;
            p->team = !ls_fft_supported(T);
            p->team_piece = 0;
        }
    }
    p->fft_waves = p->team_chain ? ls_team_chain_teams_per_block(d->n, p->team_piece, d->max_blocks, (int)prc_opt(PRC_OPT_LS_TEAM_PIECES))
                   : p->team     ? ls_team_teams_per_block(d->n, T) : ls_fft_waves_per_block(d->n, T);
    p->nblk = p->method == 2 ? p->fft_waves : (int)ceil_div64(d->n, LSC_BLK);
    hipError_t e = hipMalloc(&p->d_partial, sizeof(float2) * (size_t)d->max_blocks * p->nblk * 2 * T);
    if (e == hipSuccess) e = hipMalloc(&p->d_taps, sizeof(double2) * (size_t)d->max_blocks * T);
    if (e == hipSuccess) e = hipMalloc(&p->d_rhs, sizeof(double2) * (size_t)d->max_blocks * T);
    if (e == hipSuccess) e = hipMalloc(&p->d_tmp[0], sizeof(float2) * (size_t)d->max_blocks * d->n);
    if (e == hipSuccess) e = hipMalloc(&p->d_tmp[1], sizeof(float2) * (size_t)d->max_blocks * d->n);
    if (e == hipSuccess && p->method == 2 && !d->circular && (!p->team || p->team_chain)) {
        e = hipMalloc(&p->d_c0, sizeof(double2) * (size_t)d->max_blocks * T);
        if (e == hipSuccess) e = hipMalloc(&p->d_se, sizeof(double2) * (size_t)d->max_blocks * T);
        p->chain = true;
        // T_0^{-1} in Gohberg-Semencul form when its vectors fit the CU's LDS (they do up to the 769 taps of these
        // kernels); the dense Trench inverse is the fallback
        p->gs_G = (T + 3) / 4;
        p->gs_parts = LSS_THREADS / p->gs_G;
        if (p->gs_parts > 16) p->gs_parts = 16;
        if (p->gs_parts < 1) p->gs_parts = 1;
        p->gs_span = (T + p->gs_parts - 1) / p->gs_parts;
        p->gs_Q = (T + 8 + 3) / 4;
        p->gs_lds = sizeof(double2) * ((size_t)8 * T + 16 * (size_t)p->gs_Q +
                                       (size_t)p->gs_parts * 4 * p->gs_G + 2 * (size_t)d->peek);
        const bool use_gs = p->gs_lds <= 160 * 1024 && p->gs_G <= LSS_THREADS;
        if (use_gs) {
            if (e == hipSuccess) e = hipMalloc(&p->d_apred, sizeof(double2) * (size_t)d->max_blocks * T);
            if (e == hipSuccess) e = hipMalloc(&p->d_perr, sizeof(double) * (size_t)d->max_blocks);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)ls_solve_gs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        } else if (e == hipSuccess) {
            e = hipMalloc(&p->d_tinv, sizeof(double2) * (size_t)d->max_blocks * T * T);
        }
        if (e == hipSuccess) e = hipMalloc(&p->d_taps_t, sizeof(double2) * (size_t)d->max_blocks * T);
        // auto / method 3: keep FFT(rho block) in an HBM spectrum cache instead of recomputing it per
        // bin (measured: the fused kernel is HBM-bound at 2 FFTs per block and VALU-bound at 3).  If the
        // cache does not fit, the chain silently recomputes.  (The 4096-point chain took its cache above.)
        if (e == hipSuccess && d->method != 2 && !p->team_chain) (void)try_cache(ls_cache_elems_per_block(d->n, T));
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)ls_prepare_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)ls_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)levinson_wave_kernel<0>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)levinson_wave_kernel<1>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess && levinson_c_global(T)) e = hipMalloc(&p->d_cws, sizeof(double2) * (size_t)d->max_blocks * T);
    if (e == hipSuccess && levinson_mode(T) == 2) e = hipMalloc(&p->d_aws, sizeof(double2) * (size_t)d->max_blocks * 2 * T);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)fir_subtract_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
        const bool held_cache = allow_cache && p->d_cache != nullptr;
        prc_set_error("prc_ls_plan_create: device setup failed: %s", hipGetErrorString(e));
        prc_ls_plan_destroy(p);
        (void)hipGetLastError();
        // near the HBM limit the optional cache may have taken what a mandatory buffer needed: once more without it
        if (held_cache && e == hipErrorOutOfMemory) return ls_plan_create_impl(plan, d, false);
        return PRC_EHIP;
    }
    *plan = p;
    return PRC_OK;
}

extern "C" int prc_ls_plan_create(prc_ls_plan** plan, const prc_ls_desc* host_desc) {
    PRC_REQUIRE(plan && host_desc, PRC_EINVAL, "prc_ls_plan_create: null argument");
    static_assert(sizeof(prc_ls_desc) == PRC_LS_DESC_SIZE_600, "prc_ls_desc grew: keep PRC_LS_DESC_SIZE_600, default the new fields to 0");
    prc_ls_desc mine;
    if (int rc = prc_take_desc(&mine, host_desc, PRC_LS_DESC_SIZE_600, "prc_ls_plan_create", "prc_ls_desc")) return rc;
    return ls_plan_create_impl(plan, &mine, true);
}

static PhaseRamp make_ramp(double fc, double fs, double phase_offset) {
    PhaseRamp pr;
    pr.a32 = (float)(2.0 * 3.14159265358979323846 * fc);
    pr.rcp32 = 1.0f / (float)fs;
    pr.off32 = (float)phase_offset;
    pr.enabled = (fc != 0.0 || phase_offset != 0.0) ? 1 : 0;
    return pr;
}

static void fill_xa(LsFftArgs& xa, prc_ls_plan* p, const void* ref, int64_t stride, const float2* cur,
                    int64_t cur_stride, float2* dst, int64_t dst_stride, const PhaseRamp& pr) {
    xa.ref = (const float2*)ref;  xa.ref_stride = stride;
    xa.srv = cur;                 xa.srv_stride = cur_stride;
    xa.out = dst;                 xa.out_stride = dst_stride;
    xa.partial = p->d_partial;
    xa.taps = p->d_taps;
    xa.taps_t = p->d_taps_t;
    xa.cache = p->d_cache;
    xa.tab = nullptr;
    xa.n = p->desc.n;
    xa.T = p->T;
    xa.peek = p->desc.peek;
    xa.circular = p->desc.circular;
    xa.piece = p->team_piece;           // 4096-point chain; the other kernels set their own
    xa.rot = pr.enabled;
    xa.pr = pr;
    xa.has_next = 0;
    xa.rot2 = 0;
    xa.rot_in = 0;
    xa.pr2 = pr;
}

// Cached-spectrum chain (LS_Filter_Multiple on long linear blocks):
//   corr(bin 0) + cache -> prepare (Durbin + Trench, once) -> solve(0)
//   -> [ FIR(i) fused with corr(i+1) -> solve(i+1) ] for every bin.
static int run_cached_chain(prc_ls_plan* p, const void* ref, const void* srv, int64_t stride, void* out,
                            int64_t out_stride, int nblocks, double sample_rate, const double* bins,
                            int nbins, double reg, hipStream_t stream) {
    const int T = p->T;
    const int64_t n = p->desc.n;
    const int RP = (T + 63) & ~63;
    const int parts = LSS_THREADS / RP > 0 ? LSS_THREADS / RP : 1;
    const size_t solve_lds = sizeof(double2) * ((size_t)5 * T + (size_t)parts * RP);
    auto theta_exact = [&](int i) { return 2.0 * 3.14159265358979323846 * bins[i] / sample_rate; };
    auto launch_solve = [&](int i, const float2* cur, int64_t cur_stride) -> int {
        const PhaseRamp pr = make_ramp(bins[i], sample_rate, 0.0);
        LsSolveArgs sa;
        sa.partial = p->d_partial;  sa.c0 = p->d_c0;  sa.se = p->d_se;  sa.tinv = p->d_tinv;
        sa.ref = (const float2*)ref;  sa.ref_stride = stride;
        sa.srv = cur;  sa.srv_stride = cur_stride;
        sa.taps = p->d_taps;  sa.taps_t = p->d_taps_t;
        sa.n = n;  sa.nblk = p->nblk;  sa.T = T;  sa.peek = p->desc.peek;
        // effective slope of the reference's float32 ramp: fl32(2 pi f) * fl32(1/Fs)
        sa.theta = pr.enabled ? (double)pr.a32 * (double)pr.rcp32 : 0.0;
        sa.srv_rotated = i > 0 ? 1 : 0;                // fused(i-1) stored its output in this bin's frame
        // one refinement squares the wrap perturbation (~ a few peek/N); two for shorter blocks; none
        // when the ramp closes on itself over the block (gamma = e^{-j theta N} = 1: c_f = D c_0 exactly)
        const double gm1 = hypot(cos(sa.theta * (double)n) - 1.0, sin(sa.theta * (double)n));
        // relative size of the perturbation (gamma-1) S_e against c_0: ~ |gamma-1| * 10 peek / N;
        // k refinement steps leave ~ est^(k+1)
        const double est = pr.enabled ? gm1 * 10.0 * (double)p->desc.peek / (double)n : 0.0;
        sa.nref = 0;
        for (double left = est; left > 1e-9 && sa.nref < 4; left *= est) ++sa.nref;
        if (p->d_apred) {
            LsGsArgs ga;
            ga.partial = sa.partial;  ga.c0 = sa.c0;  ga.se = sa.se;  ga.apred = p->d_apred;  ga.perr = p->d_perr;
            ga.ref = sa.ref;  ga.srv = sa.srv;  ga.ref_stride = sa.ref_stride;  ga.srv_stride = sa.srv_stride;
            ga.taps = sa.taps;  ga.taps_t = sa.taps_t;
            ga.n = n;  ga.nblk = sa.nblk;  ga.T = T;  ga.nref = sa.nref;  ga.peek = sa.peek;
            ga.srv_rotated = sa.srv_rotated;  ga.theta = sa.theta;
            ga.G = p->gs_G;  ga.parts = p->gs_parts;  ga.span = p->gs_span;  ga.Q = p->gs_Q;
            hipLaunchKernelGGL(ls_solve_gs_kernel, dim3(nblocks), dim3(LSS_THREADS), p->gs_lds, stream, ga);
        } else {
            hipLaunchKernelGGL(ls_solve_kernel, dim3(nblocks), dim3(LSS_THREADS), solve_lds, stream, sa);
        }
        PRC_LAUNCH_CHECK();
        return PRC_OK;
    };
    const float2* cur = (const float2*)srv;
    int64_t cur_stride = stride;
    int rc;
    {
        const PhaseRamp pr0 = make_ramp(bins[0], sample_rate, 0.0);
        LsFftArgs xa;
        fill_xa(xa, p, ref, stride, cur, cur_stride, nullptr, 0, pr0);
        if (p->profiling) PRC_HIP(hipEventRecord(p->ev[0], stream));
        rc = p->team_chain ? ls_launch_corr_cached_team(xa, theta_exact(0), p->fft_waves, nblocks, stream)
                           : ls_launch_corr_cached(xa, theta_exact(0), p->fft_waves, nblocks, stream);
        if (rc) return rc;
        if (p->profiling) PRC_HIP(hipEventRecord(p->ev[1], stream));
        LsPrepArgs pa;
        pa.partial = p->d_partial;  pa.ref = (const float2*)ref;  pa.ref_stride = stride;
        pa.n = n;  pa.nblk = p->nblk;  pa.T = T;  pa.peek = p->desc.peek;  pa.reg = reg;
        pa.theta0 = 0.0;                // the chain correlates the unrotated reference
        pa.c0 = p->d_c0;  pa.se = p->d_se;  pa.tinv = p->d_tinv;  pa.apred = p->d_apred;  pa.perr = p->d_perr;
        hipLaunchKernelGGL(ls_prepare_kernel, dim3(nblocks), dim3(LS_THREADS),
                           sizeof(double2) * ((size_t)3 * T + 1), stream, pa);
        PRC_LAUNCH_CHECK();
        rc = launch_solve(0, cur, cur_stride);
        if (rc) return rc;
    }
    for (int ib = 0; ib < nbins; ++ib) {
        const PhaseRamp pr = make_ramp(bins[ib], sample_rate, 0.0);
        const bool has_next = ib + 1 < nbins;
        float2* dst = has_next ? p->d_tmp[ib & 1] : (float2*)out;
        const int64_t dst_stride = has_next ? n : out_stride;
        LsFftArgs xa;
        fill_xa(xa, p, ref, stride, cur, cur_stride, dst, dst_stride, pr);
        // the stream this bin reads is in its own rotated frame unless it is still the caller's raw input;
        // it is written in the next bin's frame (the last bin writes the true output, frame 0)
        xa.rot_in = (ib == 0 && pr.enabled) ? 1 : 0;
        double theta_out = 0.0;
        if (has_next) {
            const PhaseRamp prn = make_ramp(bins[ib + 1], sample_rate, 0.0);
            theta_out = theta_exact(ib + 1);
            xa.has_next = 1;
            xa.rot2 = prn.enabled;
            xa.pr2 = prn;
        }
        const double theta_eff = pr.enabled ? (double)pr.a32 * (double)pr.rcp32 : 0.0;
        if (p->profiling) PRC_HIP(hipEventRecord(p->ev[4 * ib + 2], stream));
        rc = p->team_chain ? ls_launch_fused_cached_team(xa, theta_exact(ib), theta_out, -theta_eff * (double)n,
                                                         p->fft_waves, nblocks, stream)
                           : ls_launch_fused_cached(xa, theta_exact(ib), theta_out, -theta_eff * (double)n, p->fft_waves,
                                                    nblocks, stream);
        if (rc) return rc;
        if (p->profiling) PRC_HIP(hipEventRecord(p->ev[4 * ib + 3], stream));
        cur = dst;
        cur_stride = dst_stride;
        if (has_next) {
            if (p->profiling) {
                PRC_HIP(hipEventRecord(p->ev[4 * (ib + 1) + 0], stream));
                PRC_HIP(hipEventRecord(p->ev[4 * (ib + 1) + 1], stream));
            }
            rc = launch_solve(ib + 1, cur, cur_stride);
            if (rc) return rc;
        }
    }
    return PRC_OK;
}

extern "C" int prc_ls_execute(prc_ls_plan* p, const void* ref, const void* srv, int64_t stride,
                              void* out, int64_t out_stride, int32_t nblocks, double sample_rate,
                              const double* bins, int32_t nbins, double reg, void* taps_out,
                              void* stream_) {
    PRC_RANGE("prc_ls_execute");
    PRC_REQUIRE(p && ref && srv && out && bins, PRC_EINVAL, "prc_ls_execute: null argument");
    PRC_REQUIRE(nblocks > 0 && nblocks <= p->desc.max_blocks, PRC_EINVAL,
                "prc_ls_execute: nblocks=%d outside [1, %d]", nblocks, p->desc.max_blocks);
    PRC_REQUIRE(nbins > 0, PRC_EINVAL, "prc_ls_execute: no Doppler bins");
    PRC_REQUIRE(stride >= p->desc.n && out_stride >= p->desc.n, PRC_ESHAPE,
                "prc_ls_execute: stride shorter than the block length");
    hipStream_t stream = (hipStream_t)stream_;
    std::lock_guard<std::mutex> lk(p->mtx);
    const int T = p->T;
    const int64_t n = p->desc.n;
    if (p->profiling) {
        while ((int)p->ev.size() < 4 * nbins) {
            hipEvent_t e;
            PRC_HIP(hipEventCreate(&e));
            p->ev.push_back(e);
        }
        p->ev_bins = nbins;
    }
    // The cached chain needs the wrap perturbation (gamma-1) S_e to be small against c_0 for the
    // refinement to converge fast (ratio ~ peek/N); short blocks and single-bin calls keep the
    // per-bin Levinson solve.
    p->last_cached = p->chain && nbins > 1 && n >= 2000LL * (p->desc.peek > 0 ? p->desc.peek : 1);
    if (p->last_cached) {
        int rc = run_cached_chain(p, ref, srv, stride, out, out_stride, nblocks, sample_rate, bins, nbins, reg,
                                  stream);
        if (rc) return rc;
    } else {
        const float2* cur = (const float2*)srv;
        int64_t cur_stride = stride;
        for (int ib = 0; ib < nbins; ++ib) {
            const PhaseRamp pr = make_ramp(bins[ib], sample_rate, 0.0);
            float2* dst = ib == nbins - 1 ? (float2*)out : p->d_tmp[ib & 1];
            const int64_t dst_stride = ib == nbins - 1 ? out_stride : n;
            const double theta = 2.0 * 3.14159265358979323846 * bins[ib] / sample_rate;
            if (p->method == 2 && pr.enabled) {
                PRC_REQUIRE(!p->desc.circular, PRC_EUNSUPPORTED,
                            "prc_ls_execute: Doppler-shifted bins with the circular (LS_Filter) form need method=1");
            }
            LsFftArgs xa;
            fill_xa(xa, p, ref, stride, cur, cur_stride, dst, dst_stride, pr);
            if (p->profiling) PRC_HIP(hipEventRecord(p->ev[4 * ib + 0], stream));
            int rc;
            if (p->method == 2) {
                rc = p->team ? ls_launch_corr_team(xa, theta, p->fft_waves, nblocks, true, stream)
                             : ls_launch_corr_fft(xa, theta, p->fft_waves, nblocks, true, stream);
            } else {
                CorrArgs ca;
                ca.p_src = (const float2*)ref;  ca.p_stride = stride;
                ca.s1_src = (const float2*)ref; ca.s1_stride = stride;
                ca.s2_src = cur;                ca.s2_stride = cur_stride;
                ca.n = n;
                ca.nlags = T;
                ca.peek_p = p->desc.peek;  ca.peek_s1 = p->desc.peek;  ca.peek_s2 = 0;
                ca.rot_p = pr.enabled;     ca.rot_s1 = pr.enabled;     ca.rot_s2 = 0;
                ca.circular = p->desc.circular;
                ca.pr = pr;
                ca.partial = p->d_partial;
                ca.nblk = p->nblk;
                rc = launch_corr(ca, true, nblocks, stream);
            }
            if (rc) return rc;
            if (p->profiling) PRC_HIP(hipEventRecord(p->ev[4 * ib + 1], stream));
            switch (levinson_mode(T)) {
                case 2:
                    hipLaunchKernelGGL(levinson_wave_kernel<2>, dim3(nblocks), dim3(64), 0, stream,
                                       p->d_partial, p->nblk, T, reg, p->d_taps, p->d_rhs, p->d_cws, p->d_aws);
                    break;
                case 1:
                    hipLaunchKernelGGL(levinson_wave_kernel<1>, dim3(nblocks), dim3(64), levinson_lds(T), stream,
                                       p->d_partial, p->nblk, T, reg, p->d_taps, p->d_rhs, p->d_cws, (double2*)nullptr);
                    break;
                default:
                    hipLaunchKernelGGL(levinson_wave_kernel<0>, dim3(nblocks), dim3(64), levinson_lds(T), stream,
                                       p->d_partial, p->nblk, T, reg, p->d_taps, p->d_rhs, (double2*)nullptr, (double2*)nullptr);
            }
            PRC_LAUNCH_CHECK();
            if (p->profiling) PRC_HIP(hipEventRecord(p->ev[4 * ib + 2], stream));
            if (p->method == 2) {
                rc = p->team ? ls_launch_fir_team(xa, theta, nblocks, stream) : ls_launch_fir_fft(xa, theta, nblocks, stream);
                if (rc) return rc;
            } else {
                FirArgs fa;
                fa.ref = (const float2*)ref;  fa.ref_stride = stride;
                fa.srv = cur;                 fa.srv_stride = cur_stride;
                fa.out = dst;                 fa.out_stride = dst_stride;
                fa.taps = p->d_taps;
                fa.n = n;
                fa.T = T;
                fa.peek = p->desc.peek;
                fa.circular = p->desc.circular;
                fa.rot = pr.enabled;
                fa.pr = pr;
                fa.tile = fir_tile(T);
                dim3 grid((unsigned)ceil_div64(n, FIR_SPAN), (unsigned)nblocks);
                hipLaunchKernelGGL(fir_subtract_kernel, grid, dim3(LS_THREADS), fir_lds(T), stream, fa);
                PRC_LAUNCH_CHECK();
            }
            if (p->profiling) PRC_HIP(hipEventRecord(p->ev[4 * ib + 3], stream));
            cur = dst;
            cur_stride = dst_stride;
        }
    }
    if (taps_out)
        PRC_HIP(hipMemcpyAsync(taps_out, p->d_taps, sizeof(double2) * (size_t)nblocks * T,
                               hipMemcpyDeviceToDevice, stream));
    return PRC_OK;
}

extern "C" int prc_ls_set_profiling(prc_ls_plan* p, int32_t enable) {
    PRC_REQUIRE(p, PRC_EINVAL, "prc_ls_set_profiling: null plan");
    std::lock_guard<std::mutex> lk(p->mtx);
    p->profiling = enable ? 1 : 0;
    p->ev_bins = 0;
    return PRC_OK;
}

// ms[0..2] = total milliseconds of the correlation / solve / FIR kernels of the LAST execute (summed
// over its Doppler bins), launches[0..2] = kernel launches behind each figure.  In the cached chain
// the correlation of bin i+1 runs inside the FIR kernel of bin i, so only bin 0 has a correlation
// launch of its own.  Synchronises on the recorded events.
extern "C" int prc_ls_get_profile(prc_ls_plan* p, double* ms, int32_t* launches) {
    PRC_REQUIRE(p && ms, PRC_EINVAL, "prc_ls_get_profile: null argument");
    std::lock_guard<std::mutex> lk(p->mtx);
    PRC_REQUIRE(p->profiling && p->ev_bins > 0, PRC_EINVAL, "prc_ls_get_profile: nothing recorded");
    ms[0] = ms[1] = ms[2] = 0.0;
    for (int ib = 0; ib < p->ev_bins; ++ib) {
        PRC_HIP(hipEventSynchronize(p->ev[4 * ib + 3]));
        for (int k = 0; k < 3; ++k) {
            float t = 0.f;
            PRC_HIP(hipEventElapsedTime(&t, p->ev[4 * ib + k], p->ev[4 * ib + k + 1]));
            ms[k] += t;
        }
    }
    if (launches) {
        launches[0] = p->last_cached ? 1 : p->ev_bins;
        launches[1] = p->ev_bins;
        launches[2] = p->ev_bins;
    }
    return PRC_OK;
}

// ---- xcorr (signal_utils.py:29-32) ---------------------------------------------------------
__global__ void xcorr_reduce_kernel(const float2* __restrict__ partial, int nblk, int nlags,
                                    float2* __restrict__ out, int out_base, int out_step,
                                    int lag_begin, int conj_flag) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x + lag_begin;
    if (k >= nlags) return;
    double re = 0, im = 0;
    for (int blk = 0; blk < nblk; ++blk) {
        const float2 v = partial[(int64_t)blk * nlags + k];
        re += (double)v.x;
        im += (double)v.y;
    }
    out[out_base + out_step * k] = make_float2((float)re, (float)(conj_flag ? -im : im));
}

extern "C" int prc_xcorr(const void* s1, const void* s2, int64_t n, int32_t nlead, int32_t nlag,
                         void* out, void* stream_) {
    PRC_RANGE("prc_xcorr");
    PRC_REQUIRE(s1 && s2 && out, PRC_EINVAL, "prc_xcorr: null argument");
    PRC_REQUIRE(n > 0 && nlead >= 0 && nlag >= 0, PRC_EINVAL, "prc_xcorr: bad size");
    // lags beyond the signal length are legal (signal_utils.py:29-32 pads s2 by nlag / nlead zeros): their sums are 0
    hipStream_t stream = (hipStream_t)stream_;
    const int nblk = (int)ceil_div64(n, LSC_BLK);
    const int maxl = (nlag > nlead ? nlag : nlead) + 1;
    float2* d_part = nullptr;
    PRC_HIP(hipMalloc(&d_part, sizeof(float2) * (size_t)nblk * maxl));
    int rc = PRC_OK;
    // d = 0..nlag:  z[nlead+d] = conj( sum_m s2[m] conj(s1[m+d]) )
    {
        CorrArgs ca = {};
        ca.p_src = (const float2*)s2;  ca.s1_src = (const float2*)s1;  ca.s2_src = nullptr;
        ca.n = n;  ca.nlags = nlag + 1;  ca.partial = d_part;  ca.nblk = nblk;
        rc = launch_corr(ca, false, 1, stream);
        if (rc == PRC_OK) {
            hipLaunchKernelGGL(xcorr_reduce_kernel, dim3((nlag + 1 + 63) / 64), dim3(64), 0, stream,
                               d_part, nblk, nlag + 1, (float2*)out, nlead, 1, 0, 1);
            if (hipGetLastError() != hipSuccess) rc = PRC_EHIP;
        }
    }
    // e = 1..nlead: z[nlead-e] = sum_n s1[n] conj(s2[n+e])
    if (rc == PRC_OK && nlead > 0) {
        CorrArgs ca = {};
        ca.p_src = (const float2*)s1;  ca.s1_src = (const float2*)s2;  ca.s2_src = nullptr;
        ca.n = n;  ca.nlags = nlead + 1;  ca.partial = d_part;  ca.nblk = nblk;
        rc = launch_corr(ca, false, 1, stream);
        if (rc == PRC_OK) {
            hipLaunchKernelGGL(xcorr_reduce_kernel, dim3((nlead + 63) / 64), dim3(64), 0, stream,
                               d_part, nblk, nlead + 1, (float2*)out, nlead, -1, 1, 0);
            if (hipGetLastError() != hipSuccess) rc = PRC_EHIP;
        }
    }
    hipError_t e = hipStreamSynchronize(stream);
    (void)hipFree(d_part);
    if (rc != PRC_OK) { prc_set_error("prc_xcorr: kernel launch failed"); return rc; }
    PRC_HIP(e);
    return PRC_OK;
}

// ---- frequency_shift (signal_utils.py:24-27) ------------------------------------------------
__global__ void freq_shift_kernel(const float2* __restrict__ x, float2* __restrict__ y, int64_t n,
                                  PhaseRamp pr) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        y[i] = cmul(x[i], phase_rot(pr, i));
}

extern "C" int prc_frequency_shift(const void* x, void* y, int64_t n, double fc, double fs,
                                   double phase_offset, void* stream) {
    PRC_RANGE("prc_frequency_shift");
    PRC_REQUIRE(x && y, PRC_EINVAL, "prc_frequency_shift: null argument");
    PRC_REQUIRE(n > 0 && fs != 0.0, PRC_EINVAL, "prc_frequency_shift: bad size or rate");
    PhaseRamp pr = make_ramp(fc, fs, phase_offset);
    pr.enabled = 1;
    int64_t blocks = ceil_div64(n, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(freq_shift_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)x, (float2*)y, n, pr);
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}
