// Cross-ambiguity segment sums for wide range spans: 4096-point transforms by a four-wavefront team.
//
// Same algorithm as caf_fft.hip (range_doppler_processing.py:81-86 for the boxcar decimator, :72) --
// per slow-time sample j the (q+1)-sample segment is cut into pieces, U = FFT(w ref piece), V = FFT(srv
// piece extended by the lag span), Wacc += conj(U) V in registers, ONE inverse transform per segment and
// lag block returns all its lags -- with the 4096-point team FFT of fft_team.h: a piece carries
// 4097 - LB samples for LB lags, so 1025 lags (config 3: 1024 x 1024) cost 5 transforms of 4096 points
// per 4883-sample segment instead of 32 of 1024, and 2049 lags (config 5) cost 5 instead of 58.
// A last piece of only a few samples (config 5: 4097 = 2 x 2048 + 1) is not worth two transforms: its
// lag products are added directly after the inverse transform (TAIL_MAX samples x 16 lags per thread).
//
// Occupancy (measured on MI355X, config 3 / config 5 segment kernels, A/B on one box): the first form of this
// kernel -- two exchange buffers (one barrier per forward transform), inputs prefetched one transform ahead,
// 236 VGPRs = 2 wavefronts per SIMD -- ran at 35.6 ms per 1024 config-3 frames; one exchange buffer (two barriers
// per transform, 37 KB of LDS per workgroup), no prefetch and <= 168 VGPRs = 3 wavefronts per SIMD runs at 30.7 ms
// (-14 %; config 5: 2.93 -> 2.49 ms per 32 surfaces).  Forcing 4 wavefronts per SIMD spills and is slower (40 ms);
// removing every barrier (wrong results, upper bound) would give 31.0 ms at 2 wavefronts per SIMD.
#ifndef FT_NBUF
#define FT_NBUF 1
#endif
#ifndef CAFT_WAVES_PER_SIMD
#define CAFT_WAVES_PER_SIMD 3
#define CAFT_NO_PREFETCH 1
#endif
#include "caf_internal.h"
#include "fft_team.h"
#include "caf_team_tail.h"
#include <math.h>

void ft_make_tables(float2* t) {
    const double PI = 3.14159265358979323846;
    for (int k1 = 0; k1 < 16; ++k1)
        for (int n2 = 0; n2 < 16; ++n2) {
            const double a = -2.0 * PI * (double)(k1 * n2) / 256.0;
            t[k1 * 16 + n2] = make_float2((float)cos(a), (float)sin(a));
        }
    for (int m = 0; m < FT_P; ++m) {
        const double a = -2.0 * PI * (double)m / (double)FT_P;
        t[FT_TW1 + m] = make_float2((float)cos(a), (float)sin(a));
    }
}

static float2* g_ft_tab[16] = {nullptr};
static std::mutex g_ft_mtx;

int ft_device_tables(const float2** out) {
    int dev = 0;
    PRC_HIP(hipGetDevice(&dev));
    PRC_REQUIRE(dev >= 0 && dev < 16, PRC_EINVAL, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_ft_mtx);
    if (!g_ft_tab[dev]) {
        float2* host = new float2[FT_GTAB];
        ft_make_tables(host);
        float2* d = nullptr;
        hipError_t e = hipMalloc(&d, sizeof(float2) * FT_GTAB);
        if (e == hipSuccess) e = hipMemcpy(d, host, sizeof(float2) * FT_GTAB, hipMemcpyHostToDevice);
        delete[] host;
        PRC_HIP(e);
        g_ft_tab[dev] = d;
    }
    *out = g_ft_tab[dev];
    return PRC_OK;
}


#ifndef CAFT_WAVES_PER_SIMD
#define CAFT_WAVES_PER_SIMD 2
#endif
template <bool HAS_WIN>
__global__ __launch_bounds__(FT_THREADS, CAFT_WAVES_PER_SIMD) void caf_fft_team_kernel(CafTeamArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);
    // Workgroup -> work.  Workgroups reach the eight XCDs round-robin in launch order and every XCD has its own L2: workgroup
    // L (XCD L & 7, slot L >> 3 there) takes channel (L >> 3) % nref of chunk 8 ((L >> 3) / nref) + (L & 7).  The channels of
    // a multi-illuminator frame read the SAME surveillance windows and sit in consecutive slots of ONE XCD: one of them
    // fetches a window from HBM, the others find it in that XCD's L2 (config 5, four channels: 219 -> 96 MB fetched per
    // surface, 301 -> 281 us per frame).  Chunks themselves keep going round the XCDs: giving every XCD a contiguous run of
    // segments instead (PRC_OPT_CAF_XCD_CONTIG = 1; neighbouring segments share half a window) measured 3-6 % SLOWER at
    // one channel and at four, configs 3 and 5 alike (profiles/r04_ab_log.md, call 12) -- eight distant streams instead of
    // one; it is the option's off position that ships.
    const int per_xcd = (a.nchunks + 7) >> 3;
    const int slot = (int)(blockIdx.x >> 3);
    const int ch = slot % a.nref, ci = slot / a.nref;
    int b, bx;
    if (a.pair_half > 0) {
        // 50 %-overlapped frames: position A of the stream (in chunks of half a frame) is covered by frame A / half (its
        // first half) and by the frame before (its second half) -- the two take consecutive slots, like the channels, and
        // share the reference and surveillance samples through the L2 (config 5, 16 frames: 285 -> 276 us per four-
        // illuminator frame, 74 -> 72.4 us per surface at one channel; config 3: no difference)
        const int k = ci & 1, A = (ci >> 1) * 8 + (int)(blockIdx.x & 7u);
        if (A >= (a.nframes + 1) * a.pair_half) return;         // uniform, before any barrier
        b = A / a.pair_half - k;
        bx = A % a.pair_half + k * a.pair_half;
        if (b < 0 || b >= a.nframes || bx >= a.chunks_x) return;
    } else {
        const int chunk = a.xcd_contig ? (int)(blockIdx.x & 7u) * per_xcd + ci : ci * 8 + (int)(blockIdx.x & 7u);
        if (ci >= per_xcd || chunk >= a.nchunks) return;        // uniform, before any barrier
        b = chunk / a.chunks_x;
        bx = chunk - b * a.chunks_x;
    }
    const float2* __restrict__ ref = a.refs[ch] + (int64_t)b * a.s.frame_stride;
    const float2* __restrict__ srv = a.s.srv + (int64_t)b * a.s.frame_stride;
    const float* __restrict__ win = a.s.window;
    float2* __restrict__ ych = a.s.y + (int64_t)ch * a.y_ref_stride;               // this channel's surfaces
    const FtLane f = ft_setup(lds, a.gtab);
    const int t = f.t;
    // frame-relative 32-bit arithmetic (n < 2^31); everything but t is workgroup-uniform
    const int N = (int)a.s.n, NV = (int)a.s.n_valid;
    const int R = a.s.range_bins;
    const int B = a.piece, LB = a.lagblk;
    const unsigned vo8 = (unsigned)t * 8u, vo4 = (unsigned)t * 4u;
    auto clampu = [](int x) { return x < 0 ? 0u : (unsigned)x; };
    const float sc = 1.0f / (float)FT_P;

    for (int sg = 0; sg < a.segs; ++sg) {
        const int64_t j = (int64_t)bx * a.segs + sg;
        if (j >= a.s.freq_bins) break;                         // uniform
        const int64_t n_hi64 = j * a.s.q + a.s.half;
        const int64_t n_lo64 = n_hi64 - (a.s.ntaps - 1);
        const int lo = n_lo64 < 0 ? 0 : (int)n_lo64;
        const int hi = n_hi64 > N - 1 ? N - 1 : (int)n_hi64;
        // a short remainder after the last full piece goes the direct way
        const int len = hi - lo + 1;
        int tail = len % B;
        if (tail > CAFT_TAIL_MAX || len < B) tail = 0;
        const int hi_f = hi - tail;                            // last sample that goes through the transforms

        for (int lb = 0; lb < a.nlagblk; ++lb) {
#ifdef CAFT_EXP_NOACC           // timing ablation, never shipped (wrong results): the accumulator shares the registers of the
            float2 v[16];       // surveillance spectrum -- what the kernel would run like if its state were 64 VGPRs, not 96
            float2(&acc)[16] = v;
#else
            float2 acc[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) acc[m] = make_float2(0.f, 0.f);
#endif

            // Software pipeline as in caf_fft.hip: raw buffer loads issued one transform ahead of their use; the
            // descriptor's num_records encodes "samples of this piece that exist" (zero padding of U, ragged
            // last piece, n_valid < n, the prefetch past the last piece).
            float2 un[16];
            float wn[16];
            auto issue_u = [&](int n0, int nz = 16) {
#ifdef CAFT_EXP_NOLOAD
#pragma unroll
                for (int r = 0; r < 16; ++r) { un[r] = make_float2((float)(t + n0), (float)r); wn[r] = 0.5f; }
                return;
#endif
                const int rem = hi_f - n0 + 1;
                int cnt = rem < B ? rem : B;
                if (NV - n0 < cnt) cnt = NV - n0;
                const __amdgpu_buffer_rsrc_t ru = prc_rsrc(ref + n0, clampu(cnt) * 8u);
                // registers beyond the piece (r >= nz: 256 r >= cnt) are zero for every thread: not even loaded
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (r < 8 || (r < 12 && nz > 8) || nz > 12) un[r] = prc_buf_load_c64(ru, vo8, 2048u * r);
                    else un[r] = make_float2(0.f, 0.f);
                }
                if (HAS_WIN) {
                    const __amdgpu_buffer_rsrc_t rw = prc_rsrc(win + n0, clampu(cnt) * 4u);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (r < 8 || (r < 12 && nz > 8) || nz > 12) wn[r] = prc_buf_load_f32(rw, vo4, 1024u * r);
                        else wn[r] = 0.f;
                    }
                }
            };
            // srv slots [0, cnt+LB-1) of this lag block: frame offsets start .. with circular wrap (:82)
            auto issue_v = [&](float2 (&v)[16], int n0, int cnt) {
#ifdef CAFT_EXP_NOLOAD
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = make_float2((float)(t - n0), (float)(r + cnt));
                return;
#endif
                int start = n0 + lb * LB;
                if (start >= N) start -= N;
                const int want = cnt + LB - 1;
                int c1 = want;
                if (N - start < c1) c1 = N - start;
                if (NV - start < c1) c1 = NV - start;
                const __amdgpu_buffer_rsrc_t rv = prc_rsrc(srv + start, clampu(c1) * 8u);
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = prc_buf_load_c64(rv, vo8, 2048u * r);
                const int over = start + want - N;              // slots that wrapped (uniform, rare)
                if (over > 0) {
                    const __amdgpu_buffer_rsrc_t rw2 = prc_rsrc(srv, clampu(over < NV ? over : NV) * 8u);
                    const unsigned voff = vo8 - (unsigned)(N - start) * 8u;   // threads before the wrap: out of range
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float2 w2 = prc_buf_load_c64(rw2, voff + 2048u * r, 0u);
                        v[r].x += w2.x;
                        v[r].y += w2.y;
                    }
                }
            };
#ifndef CAFT_NO_PREFETCH
            issue_u(lo);
            for (int n0 = lo; n0 <= hi_f; n0 += B) {
                const int rem = hi_f - n0 + 1;
                const int cnt = rem < B ? rem : B;
                float2 u[16], v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    u[r] = HAS_WIN ? make_float2(un[r].x * wn[r], un[r].y * wn[r]) : un[r];
                issue_v(v, n0, cnt);
                __builtin_amdgcn_sched_barrier(0);
                ft4096_fwd<0>(u, f);
                __builtin_amdgcn_sched_barrier(0);
                issue_u(n0 + B);                                // past the last piece: zero records -> zeros
                __builtin_amdgcn_sched_barrier(0);
                ft4096_fwd<1>(v, f);
#else
            // high-occupancy form: nothing is loaded a piece ahead (the other wavefronts of the SIMD cover the latency);
            // only u, v and the accumulator are ever live together, and the surveillance loads of a piece are issued
            // before the reference transform so that they fly under it (round 3, -2 %)
            for (int n0 = lo; n0 <= hi_f; n0 += B) {
                const int rem = hi_f - n0 + 1;
                const int cnt = rem < B ? rem : B;
#ifdef CAFT_EXP_NOACC
                float2 u[16];
#else
                float2 u[16], v[16];
#endif
                // zero-padded reference piece: a piece of at most 2048 (3072) samples leaves registers 8..15 (12..15) of
                // every thread zero -- their loads, window products and first-pass additions are skipped (uniform branch)
                const int nz = cnt <= 2048 ? 8 : (cnt <= 3072 ? 12 : 16);
                if (nz == 8) {
                    issue_u(n0, 8);
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        u[r] = r < 8 ? (HAS_WIN ? make_float2(un[r].x * wn[r], un[r].y * wn[r]) : un[r]) : make_float2(0.f, 0.f);
                    issue_v(v, n0, cnt);
                    __builtin_amdgcn_sched_barrier(0);
                    ft4096_fwd<0, 8>(u, f);
                } else if (nz == 12) {
                    issue_u(n0, 12);
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        u[r] = r < 12 ? (HAS_WIN ? make_float2(un[r].x * wn[r], un[r].y * wn[r]) : un[r]) : make_float2(0.f, 0.f);
                    issue_v(v, n0, cnt);
                    __builtin_amdgcn_sched_barrier(0);
                    ft4096_fwd<0, 12>(u, f);
                } else {
                    issue_u(n0, 16);
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        u[r] = HAS_WIN ? make_float2(un[r].x * wn[r], un[r].y * wn[r]) : un[r];
                    issue_v(v, n0, cnt);
                    __builtin_amdgcn_sched_barrier(0);
                    ft4096_fwd<0, 16>(u, f);
                }
                __builtin_amdgcn_sched_barrier(0);
                ft4096_fwd<1>(v, f);
#endif
#pragma unroll
                for (int m = 0; m < 16; ++m) cmac_conj_a(acc[m], u[m], v[m]);
            }
            ft4096_inv<0>(acc, f);
            ft_team_sync();                                     // the next pass starts at buffer 0 again
            const int L0 = lb * LB;
            __builtin_amdgcn_sched_barrier(0);
            if (tail > 0) caft_tail<HAS_WIN>(acc, ref, srv, win, hi_f, tail, L0, LB, R, N, NV, t);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int within = 256 * r + t;
                const int lag = L0 + within;
                if (within < LB && lag <= R) ych[caf_y_off(a.s, b, j, R - lag)] = make_float2(acc[r].x * sc, -acc[r].y * sc);
            }
        }
    }
}

// Lag blocking for 4096-point transforms: nlb blocks of LB = ceil((R+1)/nlb) lags, pieces of 4097-LB samples;
// a pass over one lag block costs 2 transforms per piece + 1 inverse.  Returns the cost in 4096-point transforms.
double caf_team_blocking(int64_t q1, int range_bins, int* nlb_out, int* lb_out) {
    double best = 1e300;
    int best_nlb = 1, best_lb = range_bins + 1;
    for (int nlb = 1; nlb <= 64; ++nlb) {
        const int lb = (range_bins + nlb) / nlb;               // ceil((R+1)/nlb)
        if (lb > 3073) continue;
        const int64_t Bp = FT_P + 1 - lb;
        int64_t pieces = q1 / Bp;
        const int64_t rest = q1 % Bp;
        if (rest > CAFT_TAIL_MAX || pieces == 0) ++pieces;
        const double cost = (double)nlb * (2.0 * (double)pieces + 1.0);
        if (cost < best) { best = cost; best_nlb = nlb; best_lb = lb; }
        if (lb <= 2) break;
    }
    if (nlb_out) *nlb_out = best_nlb;
    if (lb_out) *lb_out = best_lb;
    return best;
}

bool caf_team_supported(int64_t n, int range_bins, int freq_bins, int boxcar) {
    (void)freq_bins;
    // n >= 8192 keeps a piece's 4096 slots from wrapping around the frame more than once
    // n <= 2^28: a frame's samples are addressed through one 32-bit buffer descriptor (bytes)
    return boxcar && range_bins >= 1 && n >= 8192 && n <= ((int64_t)1 << 28) && range_bins < n / 2;
}

int caf_launch_fft_team_refs(const CafSegArgs& s, const float2* const* refs, int nref, int64_t y_ref_stride, int nframes,
                             hipStream_t stream) {
    PRC_REQUIRE(nref >= 1 && nref <= PRC_CAF_MAX_REFS, PRC_EINVAL, "caf_launch_fft_team_refs: nref = %d", nref);
    CafTeamArgs a;
    a.s = s;
    for (int i = 0; i < PRC_CAF_MAX_REFS; ++i) a.refs[i] = i < nref ? refs[i] : nullptr;
    a.y_ref_stride = y_ref_stride;
    caf_team_blocking(s.ntaps, s.range_bins, &a.nlagblk, &a.lagblk);
    a.piece = FT_P + 1 - a.lagblk;
    int rc = ft_device_tables(&a.gtab);
    if (rc) return rc;
    // several segments per workgroup amortise the table set-up once there is plenty of work
    const int64_t total = (int64_t)s.freq_bins * nframes * nref;
    a.segs = total >= 16384 ? 4 : (total >= 4096 ? 2 : 1);
    a.nref = nref;
    a.xcd_contig = (int)prc_opt(PRC_OPT_CAF_XCD_CONTIG);
    a.chunks_x = (s.freq_bins + a.segs - 1) / a.segs;
    a.nchunks = a.chunks_x * nframes;
    a.nframes = nframes;
    a.pair_half = 0;
    if (prc_opt(PRC_OPT_CAF_PAIR_FRAMES) && nframes >= 2 && (a.chunks_x & 1) == 0) {
        const double shift = (double)s.frame_stride / ((double)s.q * a.segs);     // frame to frame, in chunks
        if (shift > a.chunks_x / 2 - 1.0 && shift < a.chunks_x / 2 + 1.0) a.pair_half = a.chunks_x / 2;
    }
    dim3 grid(a.pair_half > 0 ? (unsigned)(8 * (((nframes + 1) * a.pair_half + 7) / 8) * 2 * nref)
                              : (unsigned)(8 * ((a.nchunks + 7) / 8) * nref));
    if (prc_opt(PRC_OPT_CAF_TEAM8)) return caf_launch_fft_team8(a, grid, s.window != nullptr, stream);
    const size_t lds = sizeof(float2) * FT_LDS_ELEMS;
    { int rc_ = prc_lds_optin(reinterpret_cast<const void*>(s.window ? &caf_fft_team_kernel<true> : &caf_fft_team_kernel<false>), (int)lds); if (rc_) return rc_; }
    if (s.window)
        hipLaunchKernelGGL((caf_fft_team_kernel<true>), grid, dim3(FT_THREADS), lds, stream, a);
    else
        hipLaunchKernelGGL((caf_fft_team_kernel<false>), grid, dim3(FT_THREADS), lds, stream, a);
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}

int caf_launch_fft_team(const CafSegArgs& s, int nframes, hipStream_t stream) {
    const float2* one[1] = {s.ref};
    return caf_launch_fft_team_refs(s, one, 1, 0, nframes, stream);
}
