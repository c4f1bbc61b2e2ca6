// Cached-spectrum chain of LS_Filter_Multiple (clutter_removal.py:162-187) on the 4096-point team transform.
//
// Same algebra and the same edge handling as ls_corr_cached_kernel / ls_fused_cached_kernel of ls_fft.hip (read the
// comment block there first); the only change is the transform length.  Why it pays: both kernels are bound by HBM
// traffic, and the spectrum cache costs P complex64 per piece of P - (T-1) new samples.  For the T = 266 taps of the
// headline configuration that is 8192 B / 759 samples = 10.8 B per sample with P = 1024 and 32768 B / 3824 samples
// = 8.6 B per sample with P = 4096, next to the 16 B per sample of reading the stream and writing it back.
//
// One workgroup (four wavefronts, fft_team.h) per piece.  Slot idx = 256 r + t of a piece at n0 holds rho[n0 - ext + idx]
// for the block spectrum X_p and the (surveillance / cleaned) sample n0 + idx - ext in slots [ext, ext + cnt) for the
// correlation inputs, where the slot origin ext = T - 1 rounded up to 16 samples (ltc_piece: pieces then start on
// 128-byte lines; the extra history slots are harmless to the overlap-save FIR and to the lags 0 .. T-1).  Compiled with one exchange buffer (FT_NBUF = 1, Makefile): 36 KB of exchange + 32 KB of
// per-thread spectrum (autocorrelation accumulator / tap spectrum) = 68 KB per workgroup, two workgroups per CU.
// This file: the fused FIR + correlation kernel; the first-bin kernel sits in ls_fft_team_corr_cached.hip (its own unit so
// that the two can be built with different FT_* options: A/B runs with tools/build_variant.sh + tools/ab_ls_team.sh).  Both
// keep the T2 twiddles in registers: the first-bin kernel is VALU-bound (13 % slower with the factored form), and the fused
// kernel has the registers since the next block's prefetch moved behind the inverse transform (with the prefetch in front of
// it the twiddles spilled: 2.02 ms; factored twiddles 1.53-1.55 ms; this arrangement 1.49 ms per 256 chunk-bins).
#include "ls_team_cached.h"

// FIR of bin i fused with the cross-correlation of bin i+1: per piece the team reads X_p (cache, 32 KB) and the
// surveillance piece once, writes the cleaned piece once and keeps it in registers as the next bin's correlation input
// (one inverse and one forward transform per piece).
template <bool ROT_IN>
__global__ __launch_bounds__(FT_THREADS, 2) void ls_fused_cached_team_kernel(LsFftArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);
    const FtLane f = ft_setup(lds, a.tab);
    const int t = f.t;
    const int team = blockIdx.x, nteams = gridDim.x;
    const int b = blockIdx.y;
    const float2* __restrict__ ref = a.ref + (int64_t)b * a.ref_stride;
    const float2* __restrict__ srv = a.srv + (int64_t)b * a.srv_stride;
    float2* __restrict__ out = a.out + (int64_t)b * a.out_stride;
    const double2* __restrict__ taps = a.taps_t + (int64_t)b * a.T;
    const int n = (int)a.n;
    const int T = a.T, B = a.piece, peek = a.peek;
    const int ext = FT_P - B;                                 // slot origin of a piece: >= T - 1 (ltc_piece)
    const int nblocks = (n + B - 1) / B;
    const float2* __restrict__ cache = a.cache + (int64_t)b * nblocks * FT_P;
    const unsigned vo8 = (unsigned)t * 8u;
    const unsigned vslot = vo8 - (unsigned)ext * 8u;

    float2 xn[16];
    auto issue_x = [&](int p) {
        const bool live = p < nblocks;
#ifdef LTC_EXP_NOLOAD       // timing ablation only (wrong results): no global loads
#pragma unroll
        for (int r = 0; r < 16; ++r) xn[r] = make_float2((float)(t + p), (float)r);
        return;
#endif
        const __amdgpu_buffer_rsrc_t rc = prc_rsrc(cache + (int64_t)(live ? p : 0) * FT_P, live ? FT_P * 8u : 0u);
#pragma unroll
        for (int m = 0; m < 8; ++m) prc_buf_load_2c64(rc, (unsigned)t * 16u, 4096u * m, xn[2 * m], xn[2 * m + 1]);
    };
    // a contiguous run of pieces per team (stream4.hip: +3 % over pieces team, team + nteams, ...)
#ifdef LTC_STRIDED
    const int p0 = team, pstep = nteams, pend = nblocks;
#else
    const int per = (nblocks + nteams - 1) / nteams;
    const int p0 = team * per, pstep = 1, pend = p0 + per < nblocks ? p0 + per : nblocks;
#endif
    issue_x(p0 < pend ? p0 : nblocks);                        // flies under the transform of the taps

    // H~ = FFT(taps) / 4096 of this block (frequency layout); element (r, t) is private to thread t and parked in LDS
    float2* Hs = lds + FT_LDS_ELEMS + t;
    const float sc = 1.0f / (float)FT_P;
    {
        float2 h[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int idx = FT_THREADS * r + t;
            const double2 tp = taps[idx < T ? idx : 0];
            h[r] = idx < T ? make_float2((float)tp.x, (float)tp.y) : make_float2(0.f, 0.f);
        }
        ft4096_fwd<1>(h, f);
#pragma unroll
        for (int r = 0; r < 16; ++r) Hs[FT_THREADS * r] = make_float2(h[r].x * sc, h[r].y * sc);
    }

    float2 wrs[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) wrs[m] = make_float2(0.f, 0.f);

    ltc_loads_landed();
    for (int p = p0; p < pend; p += pstep) {
        const int n0 = p * B;
        const int cnt = (n - n0) < B ? (n - n0) : B;
        float2 xc[16], y[16], sv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) xc[r] = xn[r];
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] = cmul(xc[r], Hs[FT_THREADS * r]);
        __builtin_amdgcn_sched_barrier(0);
#ifdef LTC_EARLY_PREFETCH    // A/B: the next block's loads issued before the inverse transform (32 more VGPRs live through it)
        issue_x(p + pstep < pend ? p + pstep : nblocks);
#endif
        {
#ifdef LTC_EXP_NOLOAD
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[r] = make_float2((float)(t - n0), (float)(r + cnt));
#else
            const __amdgpu_buffer_rsrc_t rs = prc_rsrc(srv + n0, ltc_clampu(cnt) * 8u);
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[r] = prc_buf_load_c64(rs, vslot + 2048u * r, 0u);
#endif
        }
        // one rotation on the way out: from this bin's frame to the frame of whoever reads the stream next
        const bool rot_out = a.rot || a.rot2;
        float2 obase = make_float2(1.f, 0.f), ibase = make_float2(1.f, 0.f);
        if (rot_out) {
            const int64_t idx = (int64_t)n0 - ext + t + peek;
            const float2 p1 = a.rot ? phase_rot(a.pr, idx) : make_float2(1.f, 0.f);
            float2 p2 = a.rot2 ? phase_rot(a.pr2, idx) : make_float2(1.f, 0.f);
            p2.y = -p2.y;
            obase = cmul(p1, p2);
            if (ROT_IN) ibase = make_float2(p1.x, -p1.y);
        }
        __builtin_amdgcn_sched_barrier(0);
#ifndef LTC_EXP_NOFFT       // timing ablation only (wrong results): no transforms, the memory pattern alone
        ft4096_inv<0>(y, f);
#endif
        // last `peek` outputs of the block: rho samples whose ramp restarted carry gamma instead of 1
        if (a.rot && peek > 0 && n0 + cnt > n - peek) {
            const float2 g1 = a.gamma_m1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = n0 + FT_THREADS * r + t - ext;
                const int over = nn - (n - peek);            // 0..peek-1 for affected outputs
                if (over >= 0 && nn < n) {
                    float2 acc = make_float2(0.f, 0.f);
                    for (int k = 0; k <= over && k < T; ++k) {
                        const double2 wk = taps[k];
                        cmac(acc, make_float2((float)wk.x, (float)wk.y), ref[over - k]);   // rho[nn-k] = ref[nn-k+peek-n]
                    }
                    const float2 c = cmul(g1, acc);
                    y[r].x += c.x;
                    y[r].y += c.y;
                }
            }
        }
        const __amdgpu_buffer_rsrc_t ro = prc_rsrc(out + n0, ltc_clampu(cnt) * 8u);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float2 sin_ = sv[r];
            if (ROT_IN) {                                   // raw input, rotated bin: s~ = s e^{-j phi_i}
                float2 st = a.step2[r];
                st.y = -st.y;
                sin_ = cmul(sin_, cmul(ibase, st));
            }
            float2 o = make_float2(sin_.x - y[r].x, sin_.y - y[r].y);
            if (rot_out) o = cmul(o, cmul(obase, a.step[r]));
            prc_buf_store_c64(ro, vslot + 2048u * r, 0u, o);
            // the stored piece (already in the next bin's frame) stays in registers as that bin's correlation
            // input: slots [ext, ext+cnt) only (the other slots of y are circular-convolution garbage)
            const int idx = FT_THREADS * r + t;
            const bool in = idx >= ext && idx < ext + cnt;
            y[r] = in ? o : make_float2(0.f, 0.f);
        }
#ifndef LTC_EARLY_PREFETCH
        // the next block's spectrum flies under the forward transform: issued here and not before the inverse, its 32 landing
        // registers are free during the inverse, which is what lets the T2 twiddles stay in registers without spills
        // (measured: early prefetch + factored twiddles 1.53-1.55 ms, this 1.49 ms per 256 chunk-bins)
        __builtin_amdgcn_sched_barrier(0);
        issue_x(p + pstep < pend ? p + pstep : nblocks);
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (a.has_next) {
#ifndef LTC_EXP_NOFFT
            ft4096_fwd<1>(y, f);
#endif
#pragma unroll
            for (int m = 0; m < 16; ++m) ltc_cmac_bconj(wrs[m], y[m], xc[m]);
        }
    }
    if (a.has_next) {
        ft4096_inv<0>(wrs, f);
        float2* __restrict__ part = a.partial + ((int64_t)b * nteams + team) * 2 * T;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lag = FT_THREADS * r + t;
            if (lag < T) part[T + lag] = make_float2(wrs[r].x * sc, -wrs[r].y * sc);
        }
    }
}

int ls_team_piece(int T, int align) { return ltc_piece(T, align); }

int64_t ls_team_cache_elems_per_block(int64_t n, int piece) {
    return ((n + piece - 1) / piece) * FT_P;
}

int ls_team_chain_teams_per_block(int64_t n, int piece, int max_blocks, int per) {
    // ~32 pieces per team (the tap transform at the start and the inverse of the partial sums at the end are two
    // transforms without HBM traffic behind them; 8 / 16 / 32 / 64 measured 1.596 / 1.573 / 1.571 / 1.578 ms for the fused
    // kernel at 256 chunks), but at least ~2048 workgroups per launch (four rounds of the 512 resident ones) while a team
    // keeps four pieces.  per: PRC_OPT_LS_TEAM_PIECES as the plan read it.
    if (per < 1) per = 32;
    const int64_t pieces = (n + piece - 1) / piece;
    int64_t teams = pieces / per;
    int64_t fill = (2048 + max_blocks - 1) / max_blocks;
    if (fill > pieces / 4) fill = pieces / 4;
    if (teams < fill) teams = fill;
    if (teams < 1) teams = 1;
    if (teams > 128) teams = 128;
    return (int)teams;
}

int ls_launch_fused_cached_team(LsFftArgs a, double theta, double theta_out, double gamma_angle, int teams_per_block,
                                int nblocks, hipStream_t stream) {
    ltc_fill(a, theta);
    for (int r = 0; r < 16; ++r) {
        const double ang_in = theta * (double)FT_THREADS * r, ang = (theta - theta_out) * (double)FT_THREADS * r;
        a.step2[r] = make_float2((float)cos(ang_in), (float)sin(ang_in));
        a.step[r] = make_float2((float)cos(ang), (float)sin(ang));
    }
    a.gamma_m1 = make_float2((float)(cos(gamma_angle) - 1.0), (float)sin(gamma_angle));
    int rc = ft_device_tables(&a.tab);
    if (rc) return rc;
    const dim3 grid((unsigned)teams_per_block, (unsigned)nblocks), block(FT_THREADS);
    if (a.rot_in) {
        rc = prc_lds_optin(reinterpret_cast<const void*>(&ls_fused_cached_team_kernel<true>), (int)LTC_LDS);
        if (rc) return rc;
        hipLaunchKernelGGL(ls_fused_cached_team_kernel<true>, grid, block, LTC_LDS, stream, a);
    } else {
        rc = prc_lds_optin(reinterpret_cast<const void*>(&ls_fused_cached_team_kernel<false>), (int)LTC_LDS);
        if (rc) return rc;
        hipLaunchKernelGGL(ls_fused_cached_team_kernel<false>, grid, block, LTC_LDS, stream, a);
    }
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}
