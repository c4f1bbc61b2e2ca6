// First-bin kernel of the cached-spectrum chain on the 4096-point team transform (see ls_fft_team_cached.hip for the
// chain and the slot conventions).  Three forward transforms per piece make it VALU-bound: T2 twiddles in registers,
// one exchange buffer (FT_NBUF = 1, Makefile).
#include "ls_team_cached.h"

// First bin of the chain: X_p = FFT(rho block p) -> spectrum cache; partial sums of the autocorrelation of rho and of
// the cross-correlation with the rotated surveillance stream (three forward transforms per piece).
__global__ __launch_bounds__(FT_THREADS, 2) void ls_corr_cached_team_kernel(LsFftArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);
    const FtLane f = ft_setup(lds, a.tab);
    const int t = f.t;
    const int team = blockIdx.x, nteams = gridDim.x;
    const int b = blockIdx.y;
    const float2* __restrict__ ref = a.ref + (int64_t)b * a.ref_stride;
    const float2* __restrict__ srv = a.srv + (int64_t)b * a.srv_stride;
    const int n = (int)a.n;
    const int T = a.T, B = a.piece, peek = a.peek;
    const int ext = FT_P - B;                                 // slot origin of a piece: >= T - 1 (ltc_piece)
    const int npieces = (n + B - 1) / B;
    float2* __restrict__ cache = a.cache + (int64_t)b * npieces * FT_P;
    const unsigned vo8 = (unsigned)t * 8u;
    const unsigned vslot = vo8 - (unsigned)ext * 8u;          // slot idx -> piece sample idx-ext (idx<ext: out of range)
    const __amdgpu_buffer_rsrc_t rx = prc_rsrc(ref + peek, ltc_clampu(n - peek) * 8u);

    // autocorrelation accumulator: element (r, t) is private to thread t, it only lives in LDS to save 32 VGPRs
    float2* Wrr = lds + FT_LDS_ELEMS + t;
    float2 wrs[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) { Wrr[FT_THREADS * m] = make_float2(0.f, 0.f); wrs[m] = make_float2(0.f, 0.f); }

    float2 xn[16];                                            // block of the next piece (prefetched)
    auto issue_x = [&](int p) {
        const bool live = p < npieces;
        const int mstart = live ? p * B - ext : n;            // dead prefetch (past the team's run): every slot out of range
        const unsigned voff = vo8 + (unsigned)mstart * 8u;
#pragma unroll
        for (int r = 0; r < 16; ++r) xn[r] = prc_buf_load_c64(rx, voff + 2048u * r, 0u);
    };
#ifdef LTC_STRIDED
    const int p0 = team, pstep = nteams, pend = npieces;
#else
    const int per = (npieces + nteams - 1) / nteams;          // a contiguous run of pieces per team
    const int p0 = team * per, pstep = 1, pend = p0 + per < npieces ? p0 + per : npieces;
#endif
    issue_x(p0 < pend ? p0 : npieces);
    ltc_loads_landed();
    for (int p = p0; p < pend; p += pstep) {
        const int n0 = p * B;
        const int cnt = (n - n0) < B ? (n - n0) : B;
        const int mstart = n0 - ext;
        float2 x[16], u[16], up[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = xn[r];
        // surveillance piece in slots [ext, ext+cnt), rotated below by e^{-j theta (n+peek)}
        {
            const __amdgpu_buffer_rsrc_t rs = prc_rsrc(srv + n0, ltc_clampu(cnt) * 8u);
#pragma unroll
            for (int r = 0; r < 16; ++r) u[r] = prc_buf_load_c64(rs, vslot + 2048u * r, 0u);
        }
        float2 sbase = make_float2(1.f, 0.f);
        if (a.rot) {
            sbase = phase_rot(a.pr, (int64_t)mstart + t + peek);
            sbase.y = -sbase.y;
        }
        // wrapped tail of rho (source index restarts at ref[0]); unrotated, so no phase here
        const int wstart = n - peek - mstart;
        if (peek > 0 && wstart < FT_P) {
            int cw = FT_P - wstart;
            if (cw > peek) cw = peek;
            const __amdgpu_buffer_rsrc_t rw = prc_rsrc(ref, ltc_clampu(cw) * 8u);
            const unsigned voff = vo8 - (unsigned)wstart * 8u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float2 w = prc_buf_load_c64(rw, voff + 2048u * r, 0u);
                x[r].x += w.x;
                x[r].y += w.y;
            }
        }
        // rho piece in slots [ext, ext+cnt): the same samples, masked by the range check
        {
            int cu = cnt;
            if (n - peek - n0 < cu) cu = n - peek - n0;
            const __amdgpu_buffer_rsrc_t ru = prc_rsrc(ref + peek + n0, ltc_clampu(cu) * 8u);
#pragma unroll
            for (int r = 0; r < 16; ++r) up[r] = prc_buf_load_c64(ru, vslot + 2048u * r, 0u);
            const int wst = n - peek - n0;                    // first wrapped sample of the piece
            if (peek > 0 && wst < cnt) {
                // a last piece shorter than peek starts inside the wrapped run (wst < 0): its source starts at ref[-wst]
                const int w0 = wst > 0 ? wst : 0;
                const __amdgpu_buffer_rsrc_t rw = prc_rsrc(ref + (w0 - wst), ltc_clampu(cnt - w0) * 8u);
                const unsigned voff = vslot - (unsigned)w0 * 8u;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float2 w = prc_buf_load_c64(rw, voff + 2048u * r, 0u);
                    up[r].x += w.x;
                    up[r].y += w.y;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        ft4096_fwd<0>(x, f);
        {
            // cache layout [register pair m][t][2]: registers 2m, 2m+1 of a thread are 16 contiguous bytes
            float4* __restrict__ cp = reinterpret_cast<float4*>(cache + (int64_t)p * FT_P);
#pragma unroll
            for (int m = 0; m < 8; ++m)
                cp[FT_THREADS * m + t] = make_float4(x[2 * m].x, x[2 * m].y, x[2 * m + 1].x, x[2 * m + 1].y);
        }
#ifndef LTC_EXP_NOUP                  // timing ablation, never shipped: without the reference piece's transform (an upper
        ft4096_fwd<1>(up, f);         // bound for what pruning it -- FFT(piece) = X_p - FFT(history) -- could save)
#endif
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            float2 w = Wrr[FT_THREADS * m];
            ltc_cmac_bconj(w, up[m], x[m]);
            Wrr[FT_THREADS * m] = w;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (a.rot) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float2 st = a.step[r];
                st.y = -st.y;
                u[r] = cmul(u[r], cmul(sbase, st));
            }
        }
        issue_x(p + pstep < pend ? p + pstep : npieces);
        __builtin_amdgcn_sched_barrier(0);
        ft4096_fwd<0>(u, f);
        if (FT_NBUF == 2) ft_team_sync();                     // three transforms per piece: the next piece starts at buffer 0 again
#pragma unroll
        for (int m = 0; m < 16; ++m) ltc_cmac_bconj(wrs[m], u[m], x[m]);
    }
    ft4096_inv<0>(wrs, f);
    float2 wrr[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) wrr[m] = Wrr[FT_THREADS * m];
    ft4096_inv<1>(wrr, f);
    // partial[b][team][0/1][lag] holds conj(g): the prepare / solve prologues conjugate back
    float2* __restrict__ part = a.partial + ((int64_t)b * nteams + team) * 2 * T;
    const float sc = 1.0f / (float)FT_P;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int lag = FT_THREADS * r + t;
        if (lag < T) {
            part[lag] = make_float2(wrr[r].x * sc, -wrr[r].y * sc);
            part[T + lag] = make_float2(wrs[r].x * sc, -wrs[r].y * sc);
        }
    }
}

int ls_launch_corr_cached_team(LsFftArgs a, double theta, int teams_per_block, int nblocks, hipStream_t stream) {
    ltc_fill(a, theta);
    int rc = ft_device_tables(&a.tab);
    if (rc) return rc;
    rc = prc_lds_optin(reinterpret_cast<const void*>(&ls_corr_cached_team_kernel), (int)LTC_LDS);
    if (rc) return rc;
    hipLaunchKernelGGL(ls_corr_cached_team_kernel, dim3((unsigned)teams_per_block, (unsigned)nblocks), dim3(FT_THREADS),
                       LTC_LDS, stream, a);
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}

