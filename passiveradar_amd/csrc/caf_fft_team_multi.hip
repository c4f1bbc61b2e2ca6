// Cross-ambiguity segment sums of SEVERAL reference channels against one surveillance channel on the 4096-point
// team FFT (fft_team.h); single-reference kernel and the algorithm: caf_fft_team.hip.
//
// Round 4 form.  Two wavefronts per SIMD are given (V0, V1, the accumulator and one reference piece are 128 VGPRs), so
// the kernel spends the registers and the LDS that occupancy leaves: T2 twiddles in registers, two exchange buffers
// (FT_NBUF = 2: one workgroup barrier per forward transform, 71 KB of LDS, two workgroups per CU), every load issued one
// transform ahead of its use (the second surveillance window under the first surveillance transform, reference piece
// k + 1 under the transform of piece k), zero-padded reference pieces pruned as in the single-reference kernel (a piece
// of at most 2048 / 3072 samples loads 8 / 12 registers per thread and skips their first-pass additions), packed-f32
// butterflies (fft_pk.h).  Round 3's form (T2 twiddles in LDS, whole 16-register prefetch) spilled 27 VGPRs.
#ifndef FT_NBUF
#define FT_NBUF 2
#endif
#include "caf_internal.h"
#include "fft_team.h"
#include "caf_team_tail.h"
#include <type_traits>

// ---- several reference channels against ONE surveillance channel (BASELINE config 5: four illuminators) ----------
// fast_xambg is called once per (reference, surveillance) pair (range_doppler_processing.py:81-86 is the per-pair
// unit), so the surveillance pieces of a segment are the same for every illuminator.  When a segment is at most two
// pieces (configs 3 and 5), their spectra V0, V1 stay in registers while the illuminators take turns:
//     per illuminator  U0, U1 forward, acc = conj(U0) V0 + conj(U1) V1, one inverse
// = 2 + 3 nref transforms per segment instead of 5 nref (14 instead of 20 at config 5; 8 instead of 10 for a pair), the
// surveillance channel and its lag-extended pieces read once.  Segments of more pieces go through the single-reference
// kernel once per illuminator (same results, no sharing).
struct CafTeamMultiArgs {
    CafSegArgs s;                              // s.ref unused; s.y = surface block of illuminator 0
    const float2* gtab;
    const float2* refs[PRC_CAF_MAX_REFS];
    int64_t y_ref_stride;                      // elements between the illuminators' surface blocks in y
    int32_t nref;
    int32_t piece, lagblk, nlagblk, segs;
};

#ifndef CAFT_MULTI_WAVES
#define CAFT_MULTI_WAVES 2
#endif
// w = conj(u) * v
__device__ __forceinline__ float2 cmul_conj_a(float2 u, float2 v) {
    return make_float2(fmaf(u.y, v.y, u.x * v.x), fmaf(-u.y, v.x, u.x * v.y));
}

template <int N> using caft_int = std::integral_constant<int, N>;

// NZ0 / NZ1: registers per thread that a first / second reference piece can fill (256 NZ >= its samples), fixed per launch
// by the host from the piece length and the segment length: (8, 8) at config 5, (12, 8) at config 3, (16, 16) otherwise.
// A shorter piece (first and last segments of a frame) reads zeros beyond its end through the range check.
template <bool HAS_WIN, int NZ0, int NZ1>
__global__ __launch_bounds__(FT_THREADS, CAFT_MULTI_WAVES) void caf_fft_team_multi_kernel(CafTeamMultiArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);
    const FtLane f = ft_setup(lds, a.gtab);
    const int t = f.t;
    const int b = blockIdx.y;
    const float2* __restrict__ srv = a.s.srv + (int64_t)b * a.s.frame_stride;
    const float* __restrict__ win = a.s.window;
    const int N = (int)a.s.n, NV = (int)a.s.n_valid;
    const int R = a.s.range_bins;
    const int B = a.piece, LB = a.lagblk;
    const unsigned vo8 = (unsigned)t * 8u, vo4 = (unsigned)t * 4u;
    auto clampu = [](int x) { return x < 0 ? 0u : (unsigned)x; };
    const float sc = 1.0f / (float)FT_P;

    for (int sg = 0; sg < a.segs; ++sg) {
        const int64_t j = (int64_t)blockIdx.x * a.segs + sg;
        if (j >= a.s.freq_bins) break;                         // uniform
        const int64_t n_hi64 = j * a.s.q + a.s.half;
        const int64_t n_lo64 = n_hi64 - (a.s.ntaps - 1);
        const int lo = n_lo64 < 0 ? 0 : (int)n_lo64;
        const int hi = n_hi64 > N - 1 ? N - 1 : (int)n_hi64;
        const int len = hi - lo + 1;
        int tail = len % B;
        if (tail > CAFT_TAIL_MAX || len < B) tail = 0;
        const int hi_f = hi - tail;
        // at most two pieces (checked on the host): [lo, lo + cnt0) and [lo + B, hi_f]
        const int cnt0 = hi_f - lo + 1 < B ? hi_f - lo + 1 : B;
        const int n1p = lo + B;
        const int cnt1 = hi_f - n1p + 1;                       // <= B; <= 0: one piece
        const bool two = cnt1 > 0;

        for (int lb = 0; lb < a.nlagblk; ++lb) {
            auto issue_v = [&](float2 (&v)[16], int n0, int cnt) {
                int start = n0 + lb * LB;
                if (start >= N) start -= N;
                const int want = cnt + LB - 1;
                int c1 = want;
                if (N - start < c1) c1 = N - start;
                if (NV - start < c1) c1 = NV - start;
                const __amdgpu_buffer_rsrc_t rv = prc_rsrc(srv + start, clampu(c1) * 8u);
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = prc_buf_load_c64(rv, vo8, 2048u * r);
                const int over = start + want - N;
                if (over > 0) {
                    const __amdgpu_buffer_rsrc_t rw2 = prc_rsrc(srv, clampu(over < NV ? over : NV) * 8u);
                    const unsigned voff = vo8 - (unsigned)(N - start) * 8u;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float2 w2 = prc_buf_load_c64(rw2, voff + 2048u * r, 0u);
                        v[r].x += w2.x;
                        v[r].y += w2.y;
                    }
                }
            };
            // reference piece one transform ahead: the raw samples and the window land in un / wn; registers beyond the
            // piece (r >= NZ) are not loaded
            constexpr int NZM = NZ0 > NZ1 ? NZ0 : NZ1;
            // pieces of at most 2048 samples (config 5) are loaded one transform ahead (24 VGPRs in flight); longer ones
            // (config 3's 3072-sample piece is 36) would spill next to V0, V1, the accumulator and the piece in work, and
            // are loaded where they are used
#ifdef CAFT_MULTI_NO_AHEAD
            constexpr bool AHEAD = false;
#else
            constexpr bool AHEAD = NZM <= 8;
#endif
            float2 un[NZM];
            float wn[NZM];
            auto issue_u = [&](auto nzc, const float2* __restrict__ ref, int n0, int cnt) {
                constexpr int NZ = decltype(nzc)::value;
                if (NV - n0 < cnt) cnt = NV - n0;
                const __amdgpu_buffer_rsrc_t ru = prc_rsrc(ref + n0, clampu(cnt) * 8u);
#pragma unroll
                for (int r = 0; r < NZ; ++r) un[r] = prc_buf_load_c64(ru, vo8, 2048u * r);
                if (HAS_WIN) {
                    const __amdgpu_buffer_rsrc_t rw = prc_rsrc(win + n0, clampu(cnt) * 4u);
#pragma unroll
                    for (int r = 0; r < NZ; ++r) wn[r] = prc_buf_load_f32(rw, vo4, 1024u * r);
                }
            };
            // u = w * (landed piece), zero beyond it
            auto take_u = [&](auto nzc, float2 (&u)[16]) {
                constexpr int NZ = decltype(nzc)::value;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (r < NZ) u[r] = HAS_WIN ? make_float2(un[r].x * wn[r], un[r].y * wn[r]) : un[r];
                    else u[r] = make_float2(0.f, 0.f);
                }
            };
            float2 v0[16], v1[16];
            // buffer schedule (fft_team.h: transforms alternate exchange buffers 0, 1, 0, ...; a sequence that restarts
            // at buffer 0 is separated by ft_team_sync()):  V0<0> V1<1> sync | U0<0> U1<1> inv<0> sync | ...
            issue_v(v0, lo, cnt0);
            if (two) issue_v(v1, n1p, cnt1);                    // flies under the transform of V0
            __builtin_amdgcn_sched_barrier(0);
            ft4096_fwd<0>(v0, f);
            __builtin_amdgcn_sched_barrier(0);
            if (AHEAD) issue_u(caft_int<NZ0>(), a.refs[0] + (int64_t)b * a.s.frame_stride, lo, cnt0);   // ... and the first reference piece under V1
            __builtin_amdgcn_sched_barrier(0);
            if (two) ft4096_fwd<1>(v1, f);
            if (FT_NBUF == 2) ft_team_sync();
            const int L0 = lb * LB;
            for (int i = 0; i < a.nref; ++i) {
                const float2* __restrict__ ref = a.refs[i] + (int64_t)b * a.s.frame_stride;
                float2* __restrict__ ybase = a.s.y + (int64_t)i * a.y_ref_stride;     // illuminator i's surfaces
                const bool more = i + 1 < a.nref;
                const float2* __restrict__ refn = a.refs[more ? i + 1 : i] + (int64_t)b * a.s.frame_stride;
                float2 u[16], acc[16];
                // piece 0: take what landed, issue the next piece (this illuminator's second, or the next one's first;
                // past the last illuminator an empty descriptor), transform
                if (!AHEAD) issue_u(caft_int<NZ0>(), ref, lo, cnt0);
                take_u(caft_int<NZ0>(), u);
                if (AHEAD) {
                    if (two) issue_u(caft_int<NZ1>(), ref, n1p, cnt1);
                    else issue_u(caft_int<NZ0>(), refn, lo, more ? cnt0 : 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                ft4096_fwd<0, NZ0>(u, f);
#pragma unroll
                for (int m = 0; m < 16; ++m) acc[m] = cmul_conj_a(u[m], v0[m]);
                if (two) {
                    if (!AHEAD) issue_u(caft_int<NZ1>(), ref, n1p, cnt1);
                    take_u(caft_int<NZ1>(), u);
                    if (AHEAD) issue_u(caft_int<NZ0>(), refn, lo, more ? cnt0 : 0);
                    __builtin_amdgcn_sched_barrier(0);
                    ft4096_fwd<1, NZ1>(u, f);
#pragma unroll
                    for (int m = 0; m < 16; ++m) cmac_conj_a(acc[m], u[m], v1[m]);
                    ft4096_inv<0>(acc, f);
                } else {
                    ft4096_inv<1>(acc, f);                      // one piece: U0<0> is followed by buffer 1
                }
                ft_team_sync();
                __builtin_amdgcn_sched_barrier(0);
                if (tail > 0) caft_tail<HAS_WIN>(acc, ref, srv, win, hi_f, tail, L0, LB, R, N, NV, t);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int within = 256 * r + t;
                    const int lag = L0 + within;
                    if (within < LB && lag <= R) ybase[caf_y_off(a.s, b, j, R - lag)] = make_float2(acc[r].x * sc, -acc[r].y * sc);
                }
            }
        }
    }
}

// Blocking for nref illuminators sharing the surveillance transforms: lag blocks whose segments are at most two pieces,
// cost nlb (pieces + nref (pieces + 1)) transforms.  Returns a negative value when no such blocking exists.
double caf_team_multi_blocking(int64_t q1, int range_bins, int nref, int* nlb_out, int* lb_out) {
    double best = -1.0;
    for (int nlb = 1; nlb <= 64; ++nlb) {
        const int lb = (range_bins + nlb) / nlb;
        if (lb > 3073) continue;
        const int64_t Bp = FT_P + 1 - lb;
        int64_t pieces = q1 / Bp;
        const int64_t rest = q1 % Bp;
        if (rest > CAFT_TAIL_MAX || pieces == 0) ++pieces;
        if (pieces <= 2) {
            const double cost = (double)nlb * ((double)pieces + (double)nref * ((double)pieces + 1.0));
            if (best < 0 || cost < best) {
                best = cost;
                if (nlb_out) *nlb_out = nlb;
                if (lb_out) *lb_out = lb;
            }
        }
        if (lb <= 2) break;
    }
    return best;
}

bool caf_team_multi_supported(int64_t n, int range_bins, int freq_bins, int64_t q1, int nref) {
    return nref >= 1 && nref <= PRC_CAF_MAX_REFS && caf_team_supported(n, range_bins, freq_bins, 1) &&
           caf_team_multi_blocking(q1, range_bins, nref, nullptr, nullptr) > 0;
}

int caf_launch_fft_team_multi(const CafSegArgs& s, const float2* const* refs, int nref, int64_t y_ref_stride,
                              int nframes, hipStream_t stream) {
    CafTeamMultiArgs a;
    a.s = s;
    a.nref = nref;
    a.y_ref_stride = y_ref_stride;
    for (int i = 0; i < PRC_CAF_MAX_REFS; ++i) a.refs[i] = i < nref ? refs[i] : nullptr;
    PRC_REQUIRE(caf_team_multi_blocking(s.ntaps, s.range_bins, nref, &a.nlagblk, &a.lagblk) > 0, PRC_EUNSUPPORTED,
                "caf_launch_fft_team_multi: segments of more than two pieces");
    a.piece = FT_P + 1 - a.lagblk;
    int rc = ft_device_tables(&a.gtab);
    if (rc) return rc;
    const int64_t total = (int64_t)s.freq_bins * nframes;
    a.segs = total >= 16384 ? 4 : (total >= 4096 ? 2 : 1);
    dim3 grid((unsigned)((s.freq_bins + a.segs - 1) / a.segs), (unsigned)nframes);
    const size_t lds = sizeof(float2) * FT_LDS_ELEMS;
    // the longest first / second piece of a segment (s.ntaps samples: pieces of a.piece, a tail of <= CAFT_TAIL_MAX direct)
    auto nz_of = [](int64_t cnt) { return cnt <= 2048 ? 8 : (cnt <= 3072 ? 12 : 16); };
    int64_t rest = (int64_t)s.ntaps - a.piece;
    if (rest <= CAFT_TAIL_MAX) rest = 0;
    const int nz0 = nz_of(s.ntaps < a.piece ? s.ntaps : a.piece), nz1 = rest > 0 ? nz_of(rest) : 8;
    const dim3 block(FT_THREADS);
#define CAFT_MULTI_LAUNCH(W, A, B)                                                                                 \
    do {                                                                                                           \
        rc = prc_lds_optin(reinterpret_cast<const void*>(&caf_fft_team_multi_kernel<W, A, B>), (int)lds);          \
        if (rc) return rc;                                                                                         \
        hipLaunchKernelGGL((caf_fft_team_multi_kernel<W, A, B>), grid, block, lds, stream, a);                     \
    } while (0)
    if (s.window) {
        if (nz0 == 8 && nz1 == 8) CAFT_MULTI_LAUNCH(true, 8, 8);
        else if (nz0 == 12 && nz1 == 8) CAFT_MULTI_LAUNCH(true, 12, 8);
        else CAFT_MULTI_LAUNCH(true, 16, 16);
    } else {
        if (nz0 == 8 && nz1 == 8) CAFT_MULTI_LAUNCH(false, 8, 8);
        else if (nz0 == 12 && nz1 == 8) CAFT_MULTI_LAUNCH(false, 12, 8);
        else CAFT_MULTI_LAUNCH(false, 16, 16);
    }
#undef CAFT_MULTI_LAUNCH
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}

