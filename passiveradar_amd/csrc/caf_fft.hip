// FFT-domain cross-ambiguity segment sums: the HBM-bound form of fast_xambg's hot loop.
//
// Replaces range_doppler_processing.py:81-86 (roll * ref * window -> boxcar decimate) for the
// boxcar decimator (:72).  For slow-time sample j the (q+1)-sample segment is cut into pieces of
// B = 1024 - R samples; for each piece
//     U = FFT_1024( w*ref[piece], zero padded ),  V = FFT_1024( srv[piece .. piece+B+R) ),
//     Wacc += conj(U) V                                   (registers, permuted frequency layout)
// and ONE inverse FFT per segment returns all R+1 lags:  y[j, R-l] = conj( IFFT(Wacc)[l] ).
// One wavefront owns one segment end to end (fft_wave.h: 16 points per lane, private LDS tile,
// no workgroup barrier in the loop); ref, srv and the window are each read once with
// coalesced 512-byte wave loads (srv pieces overlap by R samples, served by L2), so the kernel
// moves ~20 N bytes per frame against ~0.4 GFLOP of butterflies: HBM-bound, not VALU-bound
// (the time-domain form in caf_direct.hip is 4.9 GFLOP per frame at config 2).
#include "caf_internal.h"
#include "fft_wave.h"
#include <math.h>

void fftw_make_tables(float2* t) {
    const double PI = 3.14159265358979323846;
    for (int k1 = 0; k1 < 16; ++k1)
        for (int n2 = 0; n2 < 64; ++n2) {
            const double a = -2.0 * PI * (double)(k1 * n2) / 1024.0;
            t[k1 * 64 + n2] = make_float2((float)cos(a), (float)sin(a));
        }
    for (int m = 0; m < 16; ++m)
        for (int j = 0; j < 4; ++j) {
            const double a = -2.0 * PI * (double)(m * j) / 64.0;
            t[FFTW_TW1 + 4 * m + j] = make_float2((float)cos(a), (float)sin(a));
            const float sg = (j == 0 || j == 3) ? 1.f : -1.f;          // sA_j sB_j, see fft_wave.h
            t[FFTW_TW1 + FFTW_TW2 + 4 * m + j] = make_float2(sg * (float)cos(a), sg * (float)sin(a));
        }
}

static float2* g_dev_tab[16] = {nullptr};
static std::mutex g_tab_mtx;

// Device copy of the twiddle tables for the current device (created once per device).
int fftw_device_tables(const float2** out) {
    int dev = 0;
    PRC_HIP(hipGetDevice(&dev));
    PRC_REQUIRE(dev >= 0 && dev < 16, PRC_EINVAL, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_tab_mtx);
    if (!g_dev_tab[dev]) {
        float2 host[FFTW_TABLE];
        fftw_make_tables(host);
        float2* d = nullptr;
        PRC_HIP(hipMalloc(&d, sizeof(host)));
        PRC_HIP(hipMemcpy(d, host, sizeof(host), hipMemcpyHostToDevice));
        g_dev_tab[dev] = d;
    }
    *out = g_dev_tab[dev];
    return PRC_OK;
}

// wavefronts per workgroup (they share the twiddle tables in LDS) and, for the single-lag-block form, wavefronts per SIMD
#ifndef CAFF_WAVES
#define CAFF_WAVES 4
#endif
#ifndef CAFF_OCC
#define CAFF_OCC 3
#endif

struct CafFftArgs {
    CafSegArgs s;
    const float2* tab;
    int32_t piece;     // B = 1025 - lags_per_block samples of ref per FFT
    int32_t lagblk;    // lags per block (<= 769)
    int32_t nlagblk;   // number of lag blocks covering 0..range_bins
};

// NLB lag blocks are accumulated per pass over the segment (U = FFT(w*ref piece) is shared by
// them; V_l = FFT(srv piece shifted by l*lagblk)); more lag blocks than NLB repeat the pass.
// Occupancy: with one lag block (NLB = 1: every span up to 769 lags, the headline config) the kernel does NOT load
// ahead -- u, v and the accumulator are the only arrays live together, 163 VGPRs, three wavefronts per SIMD, and the
// third wavefront hides the load latency better than a register-hungry prefetch did at two (config 2, MI355X, A/B
// on one box: 2.49 -> 2.32 ms per 256 frames).  With two lag blocks per pass the prefetching form at two
// wavefronts per SIMD stays (three would spill).
// HAS_TAPS: a decimation FIR other than the boxcar (shortFilt=False, range_doppler_processing.py:73-78: flat-top low-pass of
// 10 q + 1 taps): the same per-segment correlation with the weight of sample n of segment j being h[n_hi - n] w[n]
// instead of w[n] -- segments of 10 q + 1 samples, ten-fold overlapped, one more 4-byte stream (the reversed taps) next to
// the window.  Single-lag-block form only (up to 769 lags); wider spans stay on the time-domain kernel.
template <bool HAS_WIN, int NLB, bool HAS_TAPS = false>
__global__ __launch_bounds__(64 * CAFF_WAVES, NLB == 1 ? CAFF_OCC : 2) void caf_fft_kernel(CafFftArgs a) {
    static_assert(!HAS_TAPS || NLB == 1, "the long-FIR form is the single-lag-block kernel");
    constexpr bool PREFETCH = NLB != 1;
    // single-lag-block form: the surveillance loads of a piece are issued BEFORE the reference transform and fly under it
#if defined(CAFF_LATE_V) || CAFF_OCC > 3
    constexpr bool EARLY_V = false;                     // (four wavefronts per SIMD: 128 VGPRs, nothing in flight under a transform)
#else
    constexpr bool EARLY_V = NLB == 1 && !HAS_TAPS;     // (a third 4-byte stream in flight: the registers are taken)
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* tab = reinterpret_cast<float2*>(smem_raw);
    float2* tile = tab + FFTW_TABLE + (threadIdx.x >> 6) * FFTW_TILE;
    fft_load_tables(tab, a.tab);
    __syncthreads();

    const FftLane f = fft_lane_setup();
    const int lane = f.lane;
    // the wave index is wave-uniform: say so, or every buffer descriptor derived from it lands in
    // VGPRs and each load becomes a waterfall loop
    const int wave_id = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t j = (int64_t)blockIdx.x * CAFF_WAVES + wave_id;
    const int b = blockIdx.y;
    if (j >= a.s.freq_bins) return;
    const float2* __restrict__ ref = a.s.ref + (int64_t)b * a.s.frame_stride;
    const float2* __restrict__ srv = a.s.srv + (int64_t)b * a.s.frame_stride;
    const float* __restrict__ win = a.s.window;
    const float* __restrict__ trev = a.s.taps_rev;
    // frame-relative 32-bit arithmetic (n < 2^31); all of it is wave-uniform (SGPRs)
    const int N = (int)a.s.n, NV = (int)a.s.n_valid;
    const int R = a.s.range_bins;
    const int B = a.piece, LB = a.lagblk;

    const int64_t n_hi64 = j * a.s.q + a.s.half;
    const int64_t n_lo64 = n_hi64 - (a.s.ntaps - 1);
    const int lo = n_lo64 < 0 ? 0 : (int)n_lo64;
    const int hi = n_hi64 > N - 1 ? N - 1 : (int)n_hi64;
    const unsigned vo8 = (unsigned)lane * 8u, vo4 = (unsigned)lane * 4u;
    auto clampu = [](int x) { return x < 0 ? 0u : (unsigned)x; };
    const float sc = 1.0f / 1024.0f;

    for (int lb0 = 0; lb0 < a.nlagblk; lb0 += NLB) {
        float2 acc[NLB][16];
#pragma unroll
        for (int l = 0; l < NLB; ++l)
#pragma unroll
            for (int m = 0; m < 16; ++m) acc[l][m] = make_float2(0.f, 0.f);

        // Software pipeline: the loads of a piece are issued one FFT ahead of their use.  They are
        // raw buffer loads: the descriptor's num_records encodes "samples of this piece that
        // exist", so the zero padding of U, the ragged last piece, n_valid < n and the prefetch past
        // the last piece all come back as zeros from the hardware range check:
        //   [u(i) resident] issue v0(i) | FFT u(i) | issue v1(i) / u(i+1) | FFT v0 | ... | acc
        float2 un[16];
        float wn[16];
        float tn[HAS_TAPS ? 16 : 1];
        // nz: registers r >= nz lie beyond the piece for every lane (64 r >= cnt): not loaded, zero
        auto issue_u = [&](int n0, int nz = 16) {
#ifdef CAFF_EXP_NOLOAD      // timing ablation only (wrong results): no global loads
#pragma unroll
            for (int r = 0; r < 16; ++r) { un[r] = make_float2((float)(lane + n0), (float)r); wn[r] = 0.5f; }
            return;
#endif
            const int rem = hi - n0 + 1;
            int cnt = rem < B ? rem : B;
            if (NV - n0 < cnt) cnt = NV - n0;
            const __amdgpu_buffer_rsrc_t ru = prc_rsrc(ref + n0, clampu(cnt) * 8u);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r < 8 || (r < 12 && nz > 8) || nz > 12) un[r] = prc_buf_load_c64(ru, vo8, 512u * r);
                else un[r] = make_float2(0.f, 0.f);
            }
            if (HAS_WIN) {
#ifdef CAFF_EXP_WINSMALL              // timing ablation, never shipped (wrong results): the window read out of its first 8 KB only --
                const __amdgpu_buffer_rsrc_t rw = prc_rsrc(win + (n0 & 1023), clampu(cnt) * 4u);   // what its 4 N bytes per frame cost
#else
                const __amdgpu_buffer_rsrc_t rw = prc_rsrc(win + n0, clampu(cnt) * 4u);
#endif
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (r < 8 || (r < 12 && nz > 8) || nz > 12) wn[r] = prc_buf_load_f32(rw, vo4, 256u * r);
                    else wn[r] = 0.f;
                }
            }
            if (HAS_TAPS) {
                const __amdgpu_buffer_rsrc_t rt = prc_rsrc(trev + (n0 - (int)n_lo64), clampu(cnt) * 4u);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (r < 8 || (r < 12 && nz > 8) || nz > 12) tn[r] = prc_buf_load_f32(rt, vo4, 256u * r);
                    else tn[r] = 0.f;
                }
            }
        };
        // weight of register r: window, taps, or their product
        auto wgt = [&](int r) { return HAS_TAPS ? (HAS_WIN ? wn[r] * tn[r] : tn[r]) : wn[r]; };
        constexpr bool WEIGHTED = HAS_WIN || HAS_TAPS;
        // srv slots [0, cnt+LB-1) of lag block lb: frame offsets start .. with circular wrap (:82)
        auto issue_v = [&](float2 (&v)[16], int n0, int cnt, int lb) {
#ifdef CAFF_EXP_NOLOAD
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = make_float2((float)(lane - n0), (float)(r + cnt + lb));
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
            for (int r = 0; r < 16; ++r) v[r] = prc_buf_load_c64(rv, vo8, 512u * r);
            const int over = start + want - N;              // slots that wrapped (wave-uniform, rare)
            if (over > 0) {
                const __amdgpu_buffer_rsrc_t rw2 = prc_rsrc(srv, clampu(over < NV ? over : NV) * 8u);
                const unsigned voff = vo8 - (unsigned)(N - start) * 8u;   // lanes before the wrap: out of range
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float2 w2 = prc_buf_load_c64(rw2, voff + 512u * r, 0u);
                    v[r].x += w2.x;
                    v[r].y += w2.y;
                }
            }
        };
        if (PREFETCH) issue_u(lo);
        for (int n0 = lo; n0 <= hi; n0 += B) {
            const int rem = hi - n0 + 1;
            const int cnt = rem < B ? rem : B;
            float2 u[16], v[NLB][16];
            // zero-padded reference piece (every span above 256 lags makes it at most 768 samples): registers 12..15
            // (8..15 for a piece of at most 512) are zero in every lane -- loads, window products and the first-pass
            // additions of the transform are skipped (wave-uniform branch; the prefetching two-lag-block form loads whole)
            const int nz = PREFETCH ? 16 : (cnt <= 512 ? 8 : (cnt <= 768 ? 12 : 16));
            if (nz == 8) {
                issue_u(n0, 8);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    u[r] = r < 8 ? (WEIGHTED ? cscale(un[r], wgt(r)) : un[r]) : make_float2(0.f, 0.f);
                if (EARLY_V) issue_v(v[0], n0, cnt, lb0);
                __builtin_amdgcn_sched_barrier(0);
                fft1024_fwd<8>(u, tile, tab, f);
            } else if (nz == 12) {
                issue_u(n0, 12);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    u[r] = r < 12 ? (WEIGHTED ? cscale(un[r], wgt(r)) : un[r]) : make_float2(0.f, 0.f);
                if (EARLY_V) issue_v(v[0], n0, cnt, lb0);
                __builtin_amdgcn_sched_barrier(0);
                fft1024_fwd<12>(u, tile, tab, f);
            } else {
                if (!PREFETCH) issue_u(n0);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    u[r] = WEIGHTED ? cscale(un[r], wgt(r)) : un[r];
                if (PREFETCH || EARLY_V) issue_v(v[0], n0, cnt, lb0);
                __builtin_amdgcn_sched_barrier(0);
                fft1024_fwd(u, tile, tab, f);
            }
#pragma unroll
            for (int l = 0; l < NLB; ++l) {
                __builtin_amdgcn_sched_barrier(0);
                if (!PREFETCH) {
                    if (!EARLY_V) issue_v(v[l], n0, cnt, lb0 + l);
                } else if (l + 1 < NLB)
                    issue_v(v[l + 1 < NLB ? l + 1 : 0], n0, cnt, lb0 + l + 1);
                else
                    issue_u(n0 + B);                        // past the last piece: zero records -> zeros
                __builtin_amdgcn_sched_barrier(0);
                fft1024_fwd(v[l], tile, tab, f);
#pragma unroll
                for (int m = 0; m < 16; ++m) cmac_conj_a(acc[l][m], u[m], v[l][m]);
            }
        }
#pragma unroll
        for (int l = 0; l < NLB; ++l) {
            fft1024_inv(acc[l], tile, tab, f);
            const int L0 = (lb0 + l) * LB;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int within = 64 * r + lane;
                const int lag = L0 + within;
                if (within < LB && lag <= R) a.s.y[caf_y_off(a.s, b, j, R - lag)] = make_float2(acc[l][r].x * sc, -acc[l][r].y * sc);
            }
        }
    }
}

// y_jk[b][j][k] -> y_kj[b][k][j]
__global__ __launch_bounds__(256) void transpose_jk_kj_kernel(const float2* __restrict__ src,
                                                              float2* __restrict__ dst, int F,
                                                              int cols) {
    __shared__ float2 t[32][33];
    const int b = blockIdx.z;
    const int64_t base = (int64_t)b * F * cols;
    const int j0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int j = j0 + r, k = k0 + tx;
        if (j < F && k < cols) t[r][tx] = src[base + (int64_t)j * cols + k];
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, j = j0 + tx;
        if (k < cols && j < F) dst[base + (int64_t)k * F + j] = t[tx][r];
    }
}

// Lag blocking: nlb blocks of LB = ceil((R+1)/nlb) lags, pieces of B = 1025-LB samples.  One pass
// over a segment costs pieces*(1+NLB) forward FFTs (+NLB inverse); pick the cheapest split.
double caf_fft_blocking(int64_t q1, int range_bins, int* nlb_out, int* lb_out) {
    double best = 1e300;
    int best_nlb = 1, best_lb = range_bins + 1;
    for (int nlb = 1; nlb <= 64; ++nlb) {
        const int lb = (range_bins + nlb) / nlb;             // ceil((R+1)/nlb)
        if (lb > 769) continue;
        const int Bp = FFTW_P + 1 - lb;
        const double pieces = (double)((q1 + Bp - 1) / Bp);
        const int passes = (nlb + 1) / 2;                      // NLB = 2 per pass when nlb > 1
        const double cost = nlb == 1 ? pieces * 2 + 1 : passes * pieces * 1.0 + pieces * nlb + nlb;
        if (cost < best) { best = cost; best_nlb = nlb; best_lb = lb; }
        if (lb <= 2) break;
    }
    if (nlb_out) *nlb_out = best_nlb;
    if (lb_out) *lb_out = best_lb;
    return best;                                               // in 1024-point transforms per segment
}

bool caf_fft_supported(int64_t n, int range_bins, int freq_bins, int boxcar) {
    (void)freq_bins;
    // n >= 2048 keeps a piece's 1024 slots from wrapping around the frame more than once; a decimation FIR other than
    // the boxcar runs on the single-lag-block kernel (up to 769 lags)
    return range_bins >= 1 && n >= 2048 && range_bins < n / 2 && (boxcar || range_bins + 1 <= 769);
}

int caf_launch_fft(const CafSegArgs& s, int nframes, hipStream_t stream) {
    CafFftArgs a;
    a.s = s;
    caf_fft_blocking(s.ntaps, s.range_bins, &a.nlagblk, &a.lagblk);
    a.piece = FFTW_P + 1 - a.lagblk;
    int rc = fftw_device_tables(&a.tab);
    if (rc) return rc;
    dim3 grid((unsigned)((s.freq_bins + CAFF_WAVES - 1) / CAFF_WAVES), (unsigned)nframes);
    const size_t lds = sizeof(float2) * (FFTW_TABLE + CAFF_WAVES * FFTW_TILE);
    const dim3 block(64 * CAFF_WAVES);
#define CAFF_LAUNCH(...)                                                                                       \
    do {                                                                                                       \
        if (lds > 64 * 1024) {                                                                                 \
            rc = prc_lds_optin(reinterpret_cast<const void*>(&caf_fft_kernel<__VA_ARGS__>), (int)lds);         \
            if (rc) return rc;                                                                                 \
        }                                                                                                      \
        hipLaunchKernelGGL((caf_fft_kernel<__VA_ARGS__>), grid, block, lds, stream, a);                        \
    } while (0)
    if (s.taps_rev) {
        PRC_REQUIRE(s.range_bins + 1 <= 769, PRC_EUNSUPPORTED, "caf_launch_fft: the long-FIR form takes up to 769 lags");
        a.nlagblk = 1;
        a.lagblk = s.range_bins + 1;
        a.piece = FFTW_P + 1 - a.lagblk;
        if (s.window) CAFF_LAUNCH(true, 1, true);
        else CAFF_LAUNCH(false, 1, true);
    } else if (a.nlagblk == 1) {
        if (s.window) CAFF_LAUNCH(true, 1);
        else CAFF_LAUNCH(false, 1);
    } else {
        if (s.window) CAFF_LAUNCH(true, 2);
        else CAFF_LAUNCH(false, 2);
    }
#undef CAFF_LAUNCH
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}

int caf_launch_transpose_jk_kj(const float2* src, float2* dst, int freq_bins, int cols, int nframes,
                               hipStream_t stream) {
    dim3 grid((freq_bins + 31) / 32, (cols + 31) / 32, nframes);
    hipLaunchKernelGGL(transpose_jk_kj_kernel, grid, dim3(256), 0, stream, src, dst, freq_bins, cols);
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}
