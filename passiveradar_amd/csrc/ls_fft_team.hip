// Block least-squares canceller for long filters: the FFT-domain kernels of ls_fft.hip on the 4096-point
// team transform of fft_team.h (770 .. 3073 taps; config 3's T = 1034 -- clutter_removal.py:142-155 at
// NLMS-sized filter lengths).  Same contract and the same arithmetic as ls_corr_fft_kernel / ls_fir_fft_kernel:
//   ls_corr_team_kernel : lags 0..T-1 of sum_m conj(r[m]) r[m+k] and sum_m conj(r[m]) s[m+k] by pieces of
//                         B = 4097-T samples (U = FFT(r piece, zero padded), V = FFT(r / s piece extended by
//                         T-1), conj(U) V accumulated in registers, one inverse per team at the end);
//   ls_fir_team_kernel  : overlap-save FIR, out = s - IFFT(FFT(r block) FFT(taps))             (:153-155).
// One workgroup (four wavefronts) per piece; linear (LS_Filter_Toeplitz / _Multiple) and circular (LS_Filter)
// boundaries; r is the peek-rotated, Doppler-rotated reference generated on the fly as in ls_fft.hip.
#include "ls_internal.h"
#include "fft_team.h"
#include <math.h>

// exp(j x) (see ls_fft.hip::small_rot)
__device__ __forceinline__ float2 lst_rot(float x) {
    if (fabsf(x) > 0.3f) {
        float s, c;
        sincosf(x, &s, &c);
        return make_float2(c, s);
    }
    const float x2 = x * x;
    const float c = 1.f + x2 * (-0.5f + x2 * (1.f / 24.f + x2 * (-1.f / 720.f)));
    const float s = x * (1.f + x2 * (-1.f / 6.f + x2 * (1.f / 120.f + x2 * (-1.f / 5040.f))));
    return make_float2(c, s);
}

struct TeamSlot {      // one register slot of the rotated reference
    bool ok;           // slot carries a sample (else zero)
    bool wr;           // source index wrapped around the block end
    int off;           // clamped source offset into ref
};

// logical r[m] = ref[(m+peek) mod n] * exp(j phi((m+peek) mod n)),  m may lie outside [0, n)
__device__ __forceinline__ TeamSlot team_slot(int m, int n, int peek, bool circular, bool want) {
    TeamSlot s;
    s.wr = false;
    bool ok = want;
    if (m >= n) { if (circular) { m -= n; s.wr = true; } else ok = false; }
    if (m < 0) { if (circular) { m += n; s.wr = true; } else ok = false; }
    int off = m + peek;
    if (off >= n) { off -= n; s.wr = true; }
    s.ok = ok;
    s.off = ok ? off : 0;
    return s;
}

__device__ __forceinline__ float2 team_finish(float2 raw, const TeamSlot& s, int rot, float theta32, float2 base,
                                              float2 step) {
    float2 v = raw;
    if (rot) {
        const float2 cont = cmul(base, step);
        const float2 wrapped = lst_rot(theta32 * (float)(s.wr ? s.off : 0));
        v = cmul(v, s.wr ? wrapped : cont);
    }
    return s.ok ? v : make_float2(0.f, 0.f);
}

template <bool AUTO>
__global__ __launch_bounds__(FT_THREADS, 2) void ls_corr_team_kernel(LsFftArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);
    const FtLane f = ft_setup(lds, a.tab);
    const int t = f.t;
    const int team = blockIdx.x, nteams = gridDim.x;
    const int b = blockIdx.y;
    const float2* __restrict__ ref = a.ref + (int64_t)b * a.ref_stride;
    const float2* __restrict__ srv = a.srv + (int64_t)b * a.srv_stride;
    const int n = (int)a.n;
    const int T = a.T, B = a.piece, ext = T - 1;
    const bool circ = a.circular != 0;

    float2 wrr[16], wrs[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) { wrr[m] = make_float2(0.f, 0.f); wrs[m] = make_float2(0.f, 0.f); }

    const int npieces = (n + B - 1) / B;
    for (int p = team; p < npieces; p += nteams) {
        const int m0 = p * B;
        const int rem = n - m0;
        const int cnt = rem < B ? rem : B;
        float2 u[16], v[16];
        float2 ebase = make_float2(1.f, 0.f);
        if (a.rot) ebase = phase_rot(a.pr, (int64_t)m0 + t + a.peek);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int idx = 256 * r + t;
            const TeamSlot es = team_slot(m0 + idx, n, a.peek, circ, idx < cnt + ext);
            v[r] = team_finish(ref[es.off], es, a.rot, a.theta32, ebase, a.step[r]);
            u[r] = idx < cnt ? v[r] : make_float2(0.f, 0.f);
        }
        ft4096_fwd<0>(u, f);
        if (AUTO) {
            ft4096_fwd<1>(v, f);
#pragma unroll
            for (int m = 0; m < 16; ++m) cmac_conj_a(wrr[m], u[m], v[m]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int idx = 256 * r + t;
            int m = m0 + idx;
            bool ok = idx < cnt + ext;
            if (m >= n) { if (circ) m -= n; else ok = false; }
            const float2 sv = srv[ok ? m : 0];
            v[r] = ok ? sv : make_float2(0.f, 0.f);
        }
        if (AUTO) {
            ft4096_fwd<0>(v, f);
            ft_team_sync();                   // three transforms per piece: the next piece starts at buffer 0 again
        } else {
            ft4096_fwd<1>(v, f);
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) cmac_conj_a(wrs[m], u[m], v[m]);
    }
    if (AUTO) ft4096_inv<0>(wrr, f);
    ft4096_inv<1>(wrs, f);
    // partial[b][team][0/1][lag] holds conj(g) so that the Levinson prologue's conj() restores g
    float2* __restrict__ part = a.partial + ((int64_t)b * nteams + team) * 2 * T;
    const float sc = 1.0f / (float)FT_P;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int lag = 256 * r + t;
        if (lag < T) {
            if (AUTO) part[lag] = make_float2(wrr[r].x * sc, -wrr[r].y * sc);
            part[T + lag] = make_float2(wrs[r].x * sc, -wrs[r].y * sc);
        }
    }
}

__global__ __launch_bounds__(FT_THREADS, 2) void ls_fir_team_kernel(LsFftArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);
    const FtLane f = ft_setup(lds, a.tab);
    const int t = f.t;
    const int team = blockIdx.x, nteams = gridDim.x;
    const int b = blockIdx.y;
    const float2* __restrict__ ref = a.ref + (int64_t)b * a.ref_stride;
    const float2* __restrict__ srv = a.srv + (int64_t)b * a.srv_stride;
    float2* __restrict__ out = a.out + (int64_t)b * a.out_stride;
    const double2* __restrict__ taps = a.taps + (int64_t)b * a.T;
    const int n = (int)a.n;
    const int T = a.T, B = a.piece, ext = T - 1;
    const bool circ = a.circular != 0;

    // H = FFT(taps zero padded) / 4096, once per team (frequency layout, registers)
    float2 h[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int idx = 256 * r + t;
        const double2 tp = taps[idx < T ? idx : 0];
        h[r] = idx < T ? make_float2((float)tp.x, (float)tp.y) : make_float2(0.f, 0.f);
    }
    ft4096_fwd<1>(h, f);
    const float sc = 1.0f / (float)FT_P;
#pragma unroll
    for (int r = 0; r < 16; ++r) { h[r].x *= sc; h[r].y *= sc; }

    const int nblocks = (n + B - 1) / B;
    for (int p = team; p < nblocks; p += nteams) {
        const int n0 = p * B;
        const int mstart = n0 - ext;                          // input index of register slot 0
        float2 x[16], sv[16];
        float2 xbase = make_float2(1.f, 0.f);
        if (a.rot) xbase = phase_rot(a.pr, (int64_t)mstart + t + a.peek);   // may be negative: e^{j theta i0}
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mstart + 256 * r + t;
            const TeamSlot xs = team_slot(m, n, a.peek, circ, m < n);
            x[r] = team_finish(ref[xs.off], xs, a.rot, a.theta32, xbase, a.step[r]);
            sv[r] = srv[(m >= 0 && m < n) ? m : 0];
        }
        ft4096_fwd<0>(x, f);
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = cmul(x[r], h[r]);
        ft4096_inv<1>(x, f);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int idx = 256 * r + t;
            const int nn = n0 + idx - ext;
            if (idx >= ext && nn < n) out[nn] = make_float2(sv[r].x - x[r].x, sv[r].y - x[r].y);
        }
    }
}

bool ls_team_supported(int T) { return T >= 2 && T - 1 <= 3072; }

int ls_team_teams_per_block(int64_t n, int T) {
    // >= ~8 pieces per team so the inverse transforms at the end stay in the noise
    const int64_t B = FT_P - (T - 1);
    const int64_t pieces = (n + B - 1) / B;
    int64_t teams = pieces / 8;
    if (teams < 1) teams = 1;
    if (teams > 64) teams = 64;
    return (int)teams;
}

static void team_fill(LsFftArgs& a, double theta) {
    a.piece = FT_P - (a.T - 1);
    a.theta32 = (float)theta;
    for (int r = 0; r < 16; ++r) {
        const double ang = theta * 256.0 * r;
        a.step[r] = make_float2((float)cos(ang), (float)sin(ang));
    }
}

int ls_launch_corr_team(LsFftArgs a, double theta, int teams_per_block, int nblocks, bool with_autocorr,
                        hipStream_t stream) {
    team_fill(a, theta);
    int rc = ft_device_tables(&a.tab);
    if (rc) return rc;
    dim3 grid((unsigned)teams_per_block, (unsigned)nblocks);
    const size_t lds = sizeof(float2) * FT_LDS_ELEMS;
    if (with_autocorr) {
        { int rc_ = prc_lds_optin(reinterpret_cast<const void*>(&ls_corr_team_kernel<true>), (int)lds); if (rc_) return rc_; }
        hipLaunchKernelGGL(ls_corr_team_kernel<true>, grid, dim3(FT_THREADS), lds, stream, a);
    } else {
        { int rc_ = prc_lds_optin(reinterpret_cast<const void*>(&ls_corr_team_kernel<false>), (int)lds); if (rc_) return rc_; }
        hipLaunchKernelGGL(ls_corr_team_kernel<false>, grid, dim3(FT_THREADS), lds, stream, a);
    }
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}

int ls_launch_fir_team(LsFftArgs a, double theta, int nblocks, hipStream_t stream) {
    team_fill(a, theta);
    int rc = ft_device_tables(&a.tab);
    if (rc) return rc;
    const int64_t B = a.piece;
    const int64_t pieces = (a.n + B - 1) / B;
    int64_t teams = (pieces + 7) / 8;                          // ~8 blocks per team: the FFT of the taps costs 1/17
    if (teams < 1) teams = 1;
    dim3 grid((unsigned)teams, (unsigned)nblocks);
    const size_t lds = sizeof(float2) * FT_LDS_ELEMS;
    { int rc_ = prc_lds_optin(reinterpret_cast<const void*>(&ls_fir_team_kernel), (int)lds); if (rc_) return rc_; }
    hipLaunchKernelGGL(ls_fir_team_kernel, grid, dim3(FT_THREADS), lds, stream, a);
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}
