// FFT-domain kernels of the block least-squares canceller (the HBM-bound form).
//
// Same contract as the time-domain kernels in ls.hip (clutter_removal.py:142-155), built on the
// one-wavefront 1024-point FFT of fft_wave.h:
//   ls_corr_fft_kernel : lags 0..T-1 of  sum_m conj(r[m]) r[m+k]  and  sum_m conj(r[m]) s[m+k]
//                        by pieces of B = 1025-T samples: U = FFT(r piece, zero padded),
//                        V = FFT(r / s piece extended by T-1), accumulate conj(U) V per wave in
//                        registers, one inverse FFT per wave at the end -> per-wave partial lags
//                        (scipy.signal.correlate at :142-147; 3 FFTs per piece instead of 2 T
//                        complex MACs per sample);
//   ls_fir_fft_kernel  : overlap-save FIR, out = s - IFFT( FFT(r block) * FFT(taps) )   (:153-155).
// r is the peek-rotated, Doppler-rotated reference generated on the fly: one sincosf per lane per
// piece at the reference's float32 phase (signal_utils.py:24-27) times a per-register constant
// step e^{j theta 64 r}; the <= peek samples that wrapped around the block end (np.roll at :139)
// restart the ramp at index 0 and take their own phase theta*index (Taylor for small arguments).  All loads are branch-free (clamped address + select) and issued one FFT ahead of
// their use, so each loop body is a single straight-line block the scheduler can overlap.
#include "ls_internal.h"
#include "fft_wave.h"

int fftw_device_tables(const float2** out);   // caf_fft.hip

#define LSF_WAVES 4

// exp(j x): 7th-order Taylor for |x| <= 0.3 (error < 2e-9, the usual case: Doppler bins of a few Hz), sincosf
// beyond (bins of kHz at a few hundred kHz of sample rate; only ever reached in the rarely taken wrap branches)
__device__ __forceinline__ float2 small_rot(float x) {
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
// index of a wrapped sample inside [0, peek]: lanes outside the wrapped run carry zeros, keep their phase argument small
__device__ __forceinline__ float wrap_index(int k, int peek) { return (float)(k < 0 ? 0 : (k > peek ? peek : k)); }

struct RefSlot {       // one register slot of the rotated reference, before the data arrived
    bool ok;           // slot carries a sample (else zero)
    bool wr;           // source index wrapped around the block end
    int off;           // clamped source offset into ref
};

// logical r[m] = ref[(m+peek) mod n] * exp(j phi((m+peek) mod n)),  m may lie outside [0, n)
__device__ __forceinline__ RefSlot ref_slot(int m, int n, int peek, bool circular, bool want) {
    RefSlot s;
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

__device__ __forceinline__ float2 ref_finish(float2 raw, const RefSlot& s, int rot, float theta32,
                                             float2 base, float2 step) {
    float2 v = raw;
    if (rot) {
        const float2 cont = cmul(base, step);
        const float2 wrapped = small_rot(theta32 * (float)(s.wr ? s.off : 0));
        v = cmul(v, s.wr ? wrapped : cont);
    }
    return s.ok ? v : make_float2(0.f, 0.f);
}

// AUTO: also accumulate the reference autocorrelation (3 FFTs per piece); otherwise only the
// cross-correlation with the surveillance stream (2 FFTs per piece, shared-inverse chain).
template <bool AUTO>
__global__ __launch_bounds__(64 * LSF_WAVES, AUTO ? 1 : 2) void ls_corr_fft_kernel(LsFftArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* tab = reinterpret_cast<float2*>(smem_raw);
    float2* tile = tab + FFTW_TABLE + (threadIdx.x >> 6) * FFTW_TILE;
    fft_load_tables(tab, a.tab);
    __syncthreads();
    const FftLane f = fft_lane_setup();
    const int lane = f.lane;
    const int wg = blockIdx.x * LSF_WAVES + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * LSF_WAVES;
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
    // pipeline state: raw reference slots of the NEXT piece to process
    float2 en[16];
    RefSlot es[16];
    float2 ebase = make_float2(1.f, 0.f);
    int ecnt = 0;
    auto issue_e = [&](int p) {
        const bool live = p < npieces;
        const int m0 = live ? p * B : 0;
        const int rem = n - m0;
        ecnt = live ? (rem < B ? rem : B) : 0;
        if (a.rot) ebase = phase_rot(a.pr, (int64_t)m0 + lane + a.peek);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int idx = 64 * r + lane;
            es[r] = ref_slot(m0 + idx, n, a.peek, circ, idx < ecnt + ext);
            en[r] = ref[es[r].off];
        }
    };
    issue_e(wg);
    for (int p = wg; p < npieces; p += nwaves) {
        const int m0 = p * B;
        const int cnt = ecnt;
        float2 u[16], v[16], sv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v[r] = ref_finish(en[r], es[r], a.rot, a.theta32, ebase, a.step[r]);
            u[r] = (64 * r + lane) < cnt ? v[r] : make_float2(0.f, 0.f);
        }
        __builtin_amdgcn_sched_barrier(0);
        fft1024_fwd(u, tile, tab, f);
        __builtin_amdgcn_sched_barrier(0);
        // issue the surveillance slots of this piece (consumed one or two FFTs later)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int idx = 64 * r + lane;
            int m = m0 + idx;
            bool ok = idx < cnt + ext;
            if (m >= n) { if (circ) m -= n; else ok = false; }
            sv[r] = srv[ok ? m : 0];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (AUTO) {
            fft1024_fwd(v, tile, tab, f);
#pragma unroll
            for (int m = 0; m < 16; ++m) cmac_conj_a(wrr[m], u[m], v[m]);
        }
        __builtin_amdgcn_sched_barrier(0);
        issue_e(p + nwaves);                      // past the end: all slots masked, reads ref[0]
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int idx = 64 * r + lane;
            const bool ok = idx < cnt + ext && (circ || m0 + idx < n);
            v[r] = ok ? sv[r] : make_float2(0.f, 0.f);
        }
        fft1024_fwd(v, tile, tab, f);
#pragma unroll
        for (int m = 0; m < 16; ++m) cmac_conj_a(wrs[m], u[m], v[m]);
    }
    if (AUTO) fft1024_inv(wrr, tile, tab, f);
    fft1024_inv(wrs, tile, tab, f);
    // partial[b][wave][0/1][lag] holds conj(g) so that the Levinson prologue's conj() restores g
    float2* __restrict__ part = a.partial + ((int64_t)b * nwaves + wg) * 2 * T;
    const float sc = 1.0f / 1024.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int lag = 64 * r + lane;
        if (lag < T) {
            if (AUTO) part[lag] = make_float2(wrr[r].x * sc, -wrr[r].y * sc);
            part[T + lag] = make_float2(wrs[r].x * sc, -wrs[r].y * sc);
        }
    }
}

__global__ __launch_bounds__(64 * LSF_WAVES, 2) void ls_fir_fft_kernel(LsFftArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* tab = reinterpret_cast<float2*>(smem_raw);
    float2* tile = tab + FFTW_TABLE + (threadIdx.x >> 6) * FFTW_TILE;
    fft_load_tables(tab, a.tab);
    __syncthreads();
    const FftLane f = fft_lane_setup();
    const int lane = f.lane;
    const int wg = blockIdx.x * LSF_WAVES + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * LSF_WAVES;
    const int b = blockIdx.y;
    const float2* __restrict__ ref = a.ref + (int64_t)b * a.ref_stride;
    const float2* __restrict__ srv = a.srv + (int64_t)b * a.srv_stride;
    float2* __restrict__ out = a.out + (int64_t)b * a.out_stride;
    const double2* __restrict__ taps = a.taps + (int64_t)b * a.T;
    const int n = (int)a.n;
    const int T = a.T, B = a.piece, ext = T - 1;
    const bool circ = a.circular != 0;

    // H = FFT(taps zero padded) / 1024, once per wave
    float2 h[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int idx = 64 * r + lane;
        const double2 t = taps[idx < T ? idx : 0];
        h[r] = idx < T ? make_float2((float)t.x, (float)t.y) : make_float2(0.f, 0.f);
    }
    fft1024_fwd(h, tile, tab, f);
    const float sc = f.sg * (1.0f / 1024.0f);        // 1/1024 and the inverse FFT's quad sign (PRESCALED)
#pragma unroll
    for (int r = 0; r < 16; ++r) { h[r].x *= sc; h[r].y *= sc; }

    const int nblocks = (n + B - 1) / B;
    float2 xn[16];
    RefSlot xs[16];
    float2 xbase = make_float2(1.f, 0.f);
    auto issue_x = [&](int p) {
        const bool live = p < nblocks;
        const int mstart = (live ? p * B : 0) - ext;       // input index of register slot 0
        if (a.rot) xbase = phase_rot(a.pr, (int64_t)mstart + lane + a.peek);  // may be negative: e^{j theta i0}
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mstart + 64 * r + lane;
            xs[r] = ref_slot(m, n, a.peek, circ, live && m < n);
            xn[r] = ref[xs[r].off];
        }
    };
    issue_x(wg);
    for (int p = wg; p < nblocks; p += nwaves) {
        const int n0 = p * B;
        float2 x[16], sv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nn = n0 + 64 * r + lane - ext;
            sv[r] = srv[(nn >= 0 && nn < n) ? nn : 0];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = ref_finish(xn[r], xs[r], a.rot, a.theta32, xbase, a.step[r]);
        __builtin_amdgcn_sched_barrier(0);
        fft1024_fwd(x, tile, tab, f);
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = cmul(x[r], h[r]);
        __builtin_amdgcn_sched_barrier(0);
        issue_x(p + nwaves);
        __builtin_amdgcn_sched_barrier(0);
        fft1024_inv<true>(x, tile, tab, f);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int idx = 64 * r + lane;
            const int nn = n0 + idx - ext;
            if (idx >= ext && nn < n) out[nn] = make_float2(sv[r].x - x[r].x, sv[r].y - x[r].y);
        }
    }
}

// ---- linear-boundary fast paths (LS_Filter_Toeplitz / LS_Filter_Multiple) -----------------------
// Same arithmetic as the two kernels above for circular == 0, with every global access turned into
// a raw buffer load/store: base + num_records live in SGPRs and the hardware range check supplies
// the zeros (padding of U, extension past the block end, the block start of the first FIR block,
// the prefetch past the last piece), so the loop bodies carry no per-lane compares, selects or
// 64-bit address arithmetic.  The <= peek samples whose source index wrapped around the block end
// (np.roll at :139) are patched in by a wave-uniform, rarely taken branch with their own phase.
__device__ __forceinline__ unsigned lsf_clampu(int x) { return x < 0 ? 0u : (unsigned)x; }

template <bool AUTO>
__global__ __launch_bounds__(64 * LSF_WAVES, AUTO ? 1 : 2) void ls_corr_fft_lin_kernel(LsFftArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* tab = reinterpret_cast<float2*>(smem_raw);
    float2* tile = tab + FFTW_TABLE + (threadIdx.x >> 6) * FFTW_TILE;
    fft_load_tables(tab, a.tab);
    __syncthreads();
    const FftLane f = fft_lane_setup();
    const int lane = f.lane;
    const int wave_id = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wg = blockIdx.x * LSF_WAVES + wave_id;
    const int nwaves = gridDim.x * LSF_WAVES;
    const int b = blockIdx.y;
    const float2* __restrict__ ref = a.ref + (int64_t)b * a.ref_stride;
    const float2* __restrict__ srv = a.srv + (int64_t)b * a.srv_stride;
    const int n = (int)a.n;
    const int T = a.T, B = a.piece, ext = T - 1, peek = a.peek;
    const unsigned vo8 = (unsigned)lane * 8u;

    float2 wrr[16], wrs[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) { wrr[m] = make_float2(0.f, 0.f); wrs[m] = make_float2(0.f, 0.f); }

    const int npieces = (n + B - 1) / B;
    float2 en[16];
    float2 ebase = make_float2(1.f, 0.f);
    auto issue_e = [&](int p) {
        const bool live = p < npieces;
        const int m0 = live ? p * B : 0;
        const int rem = n - m0;
        const int cnt = live ? (rem < B ? rem : B) : 0;
        int ce = live ? cnt + ext : 0;                        // slots wanted
        if (n - peek - m0 < ce) ce = n - peek - m0;           // ... whose source m+peek does not wrap
        const __amdgpu_buffer_rsrc_t re = prc_rsrc(ref + m0 + peek, lsf_clampu(ce) * 8u);
        if (a.rot) ebase = phase_rot(a.pr, (int64_t)m0 + lane + peek);
#pragma unroll
        for (int r = 0; r < 16; ++r) en[r] = prc_buf_load_c64(re, vo8, 512u * r);
    };
    issue_e(wg);
    for (int p = wg; p < npieces; p += nwaves) {
        const int m0 = p * B;
        const int rem = n - m0;
        const int cnt = rem < B ? rem : B;
        float2 u[16], v[16], sv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = a.rot ? cmul(en[r], cmul(ebase, a.step[r])) : en[r];
        // slots m in [n-peek, n): source wrapped to ref[0..peek), phase ramp restarted (rare)
        const int wstart = n - peek - m0;                     // first wrapped slot of this piece
        int want = cnt + ext;
        if (n - m0 < want) want = n - m0;                     // linear correlation: nothing beyond the block
        if (peek > 0 && want > wstart) {
            const __amdgpu_buffer_rsrc_t rw = prc_rsrc(ref, lsf_clampu(want - wstart) * 8u);
            const unsigned voff = vo8 - (unsigned)wstart * 8u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float2 w = prc_buf_load_c64(rw, voff + 512u * r, 0u);
                if (a.rot) w = cmul(w, small_rot(a.theta32 * wrap_index(64 * r + lane - wstart, peek)));
                v[r].x += w.x;
                v[r].y += w.y;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) u[r] = (64 * r + lane) < cnt ? v[r] : make_float2(0.f, 0.f);
        __builtin_amdgcn_sched_barrier(0);
        fft1024_fwd(u, tile, tab, f);
        __builtin_amdgcn_sched_barrier(0);
        {   // surveillance slots of this piece (consumed one or two FFTs later)
            const __amdgpu_buffer_rsrc_t rs = prc_rsrc(srv + m0, lsf_clampu(want) * 8u);
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[r] = prc_buf_load_c64(rs, vo8, 512u * r);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (AUTO) {
            fft1024_fwd(v, tile, tab, f);
#pragma unroll
            for (int m = 0; m < 16; ++m) cmac_conj_a(wrr[m], u[m], v[m]);
        }
        __builtin_amdgcn_sched_barrier(0);
        issue_e(p + nwaves);
        __builtin_amdgcn_sched_barrier(0);
        fft1024_fwd(sv, tile, tab, f);
#pragma unroll
        for (int m = 0; m < 16; ++m) cmac_conj_a(wrs[m], u[m], sv[m]);
    }
    if (AUTO) fft1024_inv(wrr, tile, tab, f);
    fft1024_inv(wrs, tile, tab, f);
    float2* __restrict__ part = a.partial + ((int64_t)b * nwaves + wg) * 2 * T;
    const float sc = 1.0f / 1024.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int lag = 64 * r + lane;
        if (lag < T) {
            if (AUTO) part[lag] = make_float2(wrr[r].x * sc, -wrr[r].y * sc);
            part[T + lag] = make_float2(wrs[r].x * sc, -wrs[r].y * sc);
        }
    }
}

__global__ __launch_bounds__(64 * LSF_WAVES, 2) void ls_fir_fft_lin_kernel(LsFftArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* tab = reinterpret_cast<float2*>(smem_raw);
    float2* tile = tab + FFTW_TABLE + (threadIdx.x >> 6) * FFTW_TILE;
    fft_load_tables(tab, a.tab);
    __syncthreads();
    const FftLane f = fft_lane_setup();
    const int lane = f.lane;
    const int wave_id = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wg = blockIdx.x * LSF_WAVES + wave_id;
    const int nwaves = gridDim.x * LSF_WAVES;
    const int b = blockIdx.y;
    const float2* __restrict__ ref = a.ref + (int64_t)b * a.ref_stride;
    const float2* __restrict__ srv = a.srv + (int64_t)b * a.srv_stride;
    float2* __restrict__ out = a.out + (int64_t)b * a.out_stride;
    const double2* __restrict__ taps = a.taps + (int64_t)b * a.T;
    const int n = (int)a.n;
    const int T = a.T, B = a.piece, ext = T - 1, peek = a.peek;
    const unsigned vo8 = (unsigned)lane * 8u;

    float2 h[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int idx = 64 * r + lane;
        const double2 t = taps[idx < T ? idx : 0];
        h[r] = idx < T ? make_float2((float)t.x, (float)t.y) : make_float2(0.f, 0.f);
    }
    fft1024_fwd(h, tile, tab, f);
    const float sc = f.sg * (1.0f / 1024.0f);        // 1/1024 and the inverse FFT's quad sign (PRESCALED)
#pragma unroll
    for (int r = 0; r < 16; ++r) { h[r].x *= sc; h[r].y *= sc; }

    const int nblocks = (n + B - 1) / B;
    // unwrapped reference: slot idx of block p <-> ref[peek + mstart + idx], valid for 0 <= mstart+idx < n-peek:
    // one descriptor for the whole block (base ref+peek), the lane offset carries mstart (negative
    // for the first block: wraps to a huge unsigned offset -> out of range -> 0)
    const __amdgpu_buffer_rsrc_t rx = prc_rsrc(ref + peek, lsf_clampu(n - peek) * 8u);
    float2 xn[16];
    float2 xbase = make_float2(1.f, 0.f);
    auto issue_x = [&](int p) {
        const bool live = p < nblocks;
        const int mstart = (live ? p * B : n) - ext;          // dead prefetch: everything out of range
        if (a.rot) xbase = phase_rot(a.pr, (int64_t)mstart + lane + peek);
        const unsigned voff = vo8 + (unsigned)mstart * 8u;
#pragma unroll
        for (int r = 0; r < 16; ++r) xn[r] = prc_buf_load_c64(rx, voff + 512u * r, 0u);
    };
    issue_x(wg);
    for (int p = wg; p < nblocks; p += nwaves) {
        const int n0 = p * B;
        const int mstart = n0 - ext;
        float2 x[16], sv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = a.rot ? cmul(xn[r], cmul(xbase, a.step[r])) : xn[r];
        const int wstart = n - peek - mstart;                 // first slot whose source wrapped (rare)
        if (peek > 0 && wstart < FFTW_P) {
            int cw = FFTW_P - wstart;
            if (cw > peek) cw = peek;
            const __amdgpu_buffer_rsrc_t rw = prc_rsrc(ref, lsf_clampu(cw) * 8u);
            const unsigned voff = vo8 - (unsigned)wstart * 8u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float2 w = prc_buf_load_c64(rw, voff + 512u * r, 0u);
                if (a.rot) w = cmul(w, small_rot(a.theta32 * wrap_index(64 * r + lane - wstart, peek)));
                x[r].x += w.x;
                x[r].y += w.y;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        fft1024_fwd(x, tile, tab, f);
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = cmul(x[r], h[r]);
        __builtin_amdgcn_sched_barrier(0);
        issue_x(p + nwaves);
        // outputs nn = n0 + idx - ext for idx >= ext: lane offset carries -ext (idx < ext -> out of range)
        const int cnt = (n - n0) < B ? (n - n0) : B;
        const unsigned vout = vo8 - (unsigned)ext * 8u;
        {
            const __amdgpu_buffer_rsrc_t rs = prc_rsrc(srv + n0, lsf_clampu(cnt) * 8u);
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[r] = prc_buf_load_c64(rs, vout + 512u * r, 0u);
        }
        __builtin_amdgcn_sched_barrier(0);
        fft1024_inv<true>(x, tile, tab, f);
        const __amdgpu_buffer_rsrc_t ro = prc_rsrc(out + n0, lsf_clampu(cnt) * 8u);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            prc_v2u d;
            d.x = __float_as_uint(sv[r].x - x[r].x);
            d.y = __float_as_uint(sv[r].y - x[r].y);
            __builtin_amdgcn_raw_buffer_store_b64(d, ro, (int)(vout + 512u * r), 0, 0);
        }
    }
}

// ---- cached-spectrum chain (linear boundary, LS_Filter_Multiple) ---------------------------------
// Everything is expressed with the UNROTATED rho = roll(ref, -peek); the Doppler rotation of bin f
// moves onto the surveillance stream (s~[n] = s[n] e^{-j theta (n+peek)}) and onto the FIR output
// (identities checked in tools/ls_shared_inverse_model.py).  Then X_p = FFT(rho[block p]) is the
// SAME for every bin: the first bin writes it to HBM (8 KB per 1025-T samples), later bins read it:
//   first bin        : 3 FFTs per piece  (X_p, rho piece, s~ piece) -> c_0 and b_0 partials, cache
//   FIR(i)+corr(i+1) : 2 FFTs per block  (inverse of X_p H~_i; forward of the cleaned piece)
// Correlations use the "roles swapped" block form: with the piece placed in slots [ext, ext+cnt) and
// zeros elsewhere,  sum_n s[n] conj(rho[n-k]) = IFFT( FFT(s slots) conj(X_p) )[k]  for k <= ext.
// The <= peek samples whose source wrapped (phase ramp restarted, factor gamma) are corrected
// exactly in the solve kernel (right-hand side) and in ls_edge_fix_kernel (last peek outputs).
__device__ __forceinline__ void cmac_bconj(float2& w, float2 u, float2 x) {   // w += u * conj(x)
    w.x = fmaf(u.x, x.x, w.x);
    w.x = fmaf(u.y, x.y, w.x);
    w.y = fmaf(u.y, x.x, w.y);
    w.y = fmaf(-u.x, x.y, w.y);
}

__global__ __launch_bounds__(64 * LSF_WAVES, 2) void ls_corr_cached_kernel(LsFftArgs a) {
    constexpr bool FIRST = true;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* tab = reinterpret_cast<float2*>(smem_raw);
    float2* tile = tab + FFTW_TABLE + (threadIdx.x >> 6) * FFTW_TILE;
    fft_load_tables(tab, a.tab);
    __syncthreads();
    const FftLane f = fft_lane_setup();
    const int lane = f.lane;
    const int wave_id = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wg = blockIdx.x * LSF_WAVES + wave_id;
    const int nwaves = gridDim.x * LSF_WAVES;
    const int b = blockIdx.y;
    const float2* __restrict__ ref = a.ref + (int64_t)b * a.ref_stride;
    const float2* __restrict__ srv = a.srv + (int64_t)b * a.srv_stride;
    const int n = (int)a.n;
    const int T = a.T, B = a.piece, ext = T - 1, peek = a.peek;
    const int npieces = (n + B - 1) / B;
    float2* __restrict__ cache = a.cache + (int64_t)b * npieces * FFTW_P;
    const unsigned vo8 = (unsigned)lane * 8u;
    const unsigned vslot = vo8 - (unsigned)ext * 8u;          // slot idx -> piece sample idx-ext (idx<ext: out of range)
    const __amdgpu_buffer_rsrc_t rx = prc_rsrc(ref + peek, lsf_clampu(n - peek) * 8u);

    // the autocorrelation accumulator lives in LDS (one 8 KB strip per wave, read-modify-write once per piece):
    // with it in registers the kernel does not fit two waves per SIMD without scratch spills
    float2* Wrr = tab + FFTW_TABLE + LSF_WAVES * FFTW_TILE + wave_id * FFTW_P;
    float2 wrs[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) { Wrr[64 * m + lane] = make_float2(0.f, 0.f); wrs[m] = make_float2(0.f, 0.f); }

    float2 xn[16];                                            // block of the next piece (prefetched)
    auto issue_x = [&](int p) {
        const bool live = p < npieces;
        if (FIRST) {
            const int mstart = (live ? p * B : n) - ext;
            const unsigned voff = vo8 + (unsigned)mstart * 8u;
#pragma unroll
            for (int r = 0; r < 16; ++r) xn[r] = prc_buf_load_c64(rx, voff + 512u * r, 0u);
        } else {
            const __amdgpu_buffer_rsrc_t rc = prc_rsrc(cache + (int64_t)(live ? p : 0) * FFTW_P,
                                                       live ? FFTW_P * 8u : 0u);
#pragma unroll
            for (int m = 0; m < 8; ++m) prc_buf_load_2c64(rc, (unsigned)lane * 16u, 1024u * m, xn[2 * m], xn[2 * m + 1]);
        }
    };
    issue_x(wg);
    for (int p = wg; p < npieces; p += nwaves) {
        const int n0 = p * B;
        const int cnt = (n - n0) < B ? (n - n0) : B;
        const int mstart = n0 - ext;
        float2 x[16], u[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = xn[r];
        // surveillance piece in slots [ext, ext+cnt), rotated by e^{-j theta (n+peek)}
        {
            const __amdgpu_buffer_rsrc_t rs = prc_rsrc(srv + n0, lsf_clampu(cnt) * 8u);
#pragma unroll
            for (int r = 0; r < 16; ++r) u[r] = prc_buf_load_c64(rs, vslot + 512u * r, 0u);
        }
        float2 sbase = make_float2(1.f, 0.f);
        if (a.rot) {
            sbase = phase_rot(a.pr, (int64_t)mstart + lane + peek);
            sbase.y = -sbase.y;
        }
        if (FIRST) {
            // wrapped tail of rho (source index restarts at ref[0]); unrotated, so no phase here
            const int wstart = n - peek - mstart;
            if (peek > 0 && wstart < FFTW_P) {
                int cw = FFTW_P - wstart;
                if (cw > peek) cw = peek;
                const __amdgpu_buffer_rsrc_t rw = prc_rsrc(ref, lsf_clampu(cw) * 8u);
                const unsigned voff = vo8 - (unsigned)wstart * 8u;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float2 w = prc_buf_load_c64(rw, voff + 512u * r, 0u);
                    x[r].x += w.x;
                    x[r].y += w.y;
                }
            }
            // rho piece in slots [ext, ext+cnt): the same samples, masked by the range check
            float2 up[16];
            {
                int cu = cnt;
                if (n - peek - n0 < cu) cu = n - peek - n0;
                const __amdgpu_buffer_rsrc_t ru = prc_rsrc(ref + peek + n0, lsf_clampu(cu) * 8u);
#pragma unroll
                for (int r = 0; r < 16; ++r) up[r] = prc_buf_load_c64(ru, vslot + 512u * r, 0u);
                const int wst = n - peek - n0;                // first wrapped sample of the piece
                if (peek > 0 && wst < cnt) {
                    // a last piece shorter than peek starts inside the wrapped run (wst < 0): the source then
                    // starts at ref[-wst], not at ref[0] -- otherwise the slots below `ext` would pick up the
                    // wrapped samples that belong to the previous piece and count them twice
                    const int w0 = wst > 0 ? wst : 0;
                    const __amdgpu_buffer_rsrc_t rw = prc_rsrc(ref + (w0 - wst), lsf_clampu(cnt - w0) * 8u);
                    const unsigned voff = vslot - (unsigned)w0 * 8u;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float2 w = prc_buf_load_c64(rw, voff + 512u * r, 0u);
                        up[r].x += w.x;
                        up[r].y += w.y;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            fft1024_fwd(x, tile, tab, f);
            if (a.cache) {
                float2* __restrict__ cp = cache + (int64_t)p * FFTW_P;
#pragma unroll
                for (int m = 0; m < 8; ++m)
                    reinterpret_cast<float4*>(cp)[64 * m + lane] = make_float4(x[2 * m].x, x[2 * m].y, x[2 * m + 1].x, x[2 * m + 1].y);
            }
            fft1024_fwd(up, tile, tab, f);
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                float2 t = Wrr[64 * m + lane];
                cmac_bconj(t, up[m], x[m]);
                Wrr[64 * m + lane] = t;
            }
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
        issue_x(p + nwaves);
        __builtin_amdgcn_sched_barrier(0);
        fft1024_fwd(u, tile, tab, f);
#pragma unroll
        for (int m = 0; m < 16; ++m) cmac_bconj(wrs[m], u[m], x[m]);
    }
    fft1024_inv(wrs, tile, tab, f);
    float2 wrr[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) wrr[m] = Wrr[64 * m + lane];
    if (FIRST) fft1024_inv(wrr, tile, tab, f);
    // partial[b][wave][0/1][lag] holds conj(g): the prepare / solve prologues conjugate back
    float2* __restrict__ part = a.partial + ((int64_t)b * nwaves + wg) * 2 * T;
    const float sc = 1.0f / 1024.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int lag = 64 * r + lane;
        if (lag < T) {
            if (FIRST) part[lag] = make_float2(wrr[r].x * sc, -wrr[r].y * sc);
            part[T + lag] = make_float2(wrs[r].x * sc, -wrs[r].y * sc);
        }
    }
}

// FIR of bin i fused with the cross-correlation of bin i+1 (cached-spectrum chain): per block the
// wave reads X_p (cache) and the surveillance piece once, writes the cleaned piece once and keeps it
// in registers as the input of the next bin's correlation -- 20 KB of HBM traffic per 1025-T
// samples per bin instead of 34 KB for the two separate kernels.
// CACHED (the default, prc_ls_desc.method 0 / 3): X_p = FFT(rho block p) is read from the spectrum cache (8 KB
// per block).  !CACHED (method 2, or when the cache does not fit): it is recomputed from the reference (6 KB per
// block + L2-served overlap, one more FFT): fewer HBM bytes but a third transform, which makes the kernel
// VALU-bound -- measured slower than the HBM-bound cached form.
#ifndef LSF_FUSED_OCC
#define LSF_FUSED_OCC 2      // wavefronts per SIMD (A/B builds: with the transforms ablated, 3 tells whether more loads in flight help)
#endif
template <bool CACHED, bool ROT_IN>
__global__ __launch_bounds__(64 * LSF_WAVES, LSF_FUSED_OCC) void ls_fused_cached_kernel(LsFftArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* tab = reinterpret_cast<float2*>(smem_raw);
    float2* tile = tab + FFTW_TABLE + (threadIdx.x >> 6) * FFTW_TILE;
    fft_load_tables(tab, a.tab);
    __syncthreads();
    const FftLane f = fft_lane_setup();
    const int lane = f.lane;
    const int wave_id = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wg = blockIdx.x * LSF_WAVES + wave_id;
    const int nwaves = gridDim.x * LSF_WAVES;
    const int b = blockIdx.y;
    const float2* __restrict__ ref = a.ref + (int64_t)b * a.ref_stride;
    const float2* __restrict__ srv = a.srv + (int64_t)b * a.srv_stride;
    float2* __restrict__ out = a.out + (int64_t)b * a.out_stride;
    const double2* __restrict__ taps = a.taps_t + (int64_t)b * a.T;
    const int n = (int)a.n;
    const int T = a.T, B = a.piece, ext = T - 1, peek = a.peek;
    const int nblocks = (n + B - 1) / B;
    const float2* __restrict__ cache = a.cache + (int64_t)b * nblocks * FFTW_P;
    const unsigned vo8 = (unsigned)lane * 8u;
    const unsigned vslot = vo8 - (unsigned)ext * 8u;

    // H~ = FFT(taps)/1024 of this block: made once by wave 0, shared by the workgroup through LDS (keeping it in
    // registers costs 32 VGPRs per lane for the whole kernel and pushes the compiler into scratch spills)
    float2* Hs = tab + FFTW_TABLE + LSF_WAVES * FFTW_TILE;
    const float sc = 1.0f / 1024.0f;
    if (wave_id == 0) {
        float2 h[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int idx = 64 * r + lane;
            const double2 t = taps[idx < T ? idx : 0];
            h[r] = idx < T ? make_float2((float)t.x, (float)t.y) : make_float2(0.f, 0.f);
        }
        fft1024_fwd(h, tile, tab, f);
#pragma unroll
        for (int r = 0; r < 16; ++r) Hs[64 * r + lane] = make_float2(h[r].x * sc * f.sg, h[r].y * sc * f.sg);   // quad sign of the inverse FFT folded in
    }
    __syncthreads();

    float2 wrs[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) wrs[m] = make_float2(0.f, 0.f);

    const __amdgpu_buffer_rsrc_t rx = prc_rsrc(ref + peek, lsf_clampu(n - peek) * 8u);
    float2 xn[16];
    auto issue_x = [&](int p) {
        const bool live = p < nblocks;
        if (CACHED) {
#ifdef LSF_EXP_NOLOAD       // timing ablation only (wrong results): no global loads
#pragma unroll
            for (int r = 0; r < 16; ++r) xn[r] = make_float2((float)(lane + p), (float)r);
            return;
#endif
            const __amdgpu_buffer_rsrc_t rc = prc_rsrc(cache + (int64_t)(live ? p : 0) * FFTW_P,
                                                       live ? FFTW_P * 8u : 0u);
            // cache layout [register pair m][lane][2]: registers 2m, 2m+1 of a lane are 16 contiguous bytes (1 KB per
            // wavefront instruction; measured: the writer 2 % faster, the reader unchanged -- it is bound by the memory
            // pattern itself, see the ablation in DESIGN.md section 4)
#pragma unroll
            for (int m = 0; m < 8; ++m) prc_buf_load_2c64(rc, (unsigned)lane * 16u, 1024u * m, xn[2 * m], xn[2 * m + 1]);
        } else {
            const int mstart = (live ? p * B : n) - ext;
            const unsigned voff = vo8 + (unsigned)mstart * 8u;
#pragma unroll
            for (int r = 0; r < 16; ++r) xn[r] = prc_buf_load_c64(rx, voff + 512u * r, 0u);
        }
    };
    issue_x(wg);
    for (int p = wg; p < nblocks; p += nwaves) {
        const int n0 = p * B;
        const int cnt = (n - n0) < B ? (n - n0) : B;
        float2 xc[16], y[16], sv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) xc[r] = xn[r];
        if (!CACHED) {
            // wrapped tail of rho (source index restarts at ref[0]); unrotated, so no phase here
            const int wstart = n - peek - (n0 - ext);
            if (peek > 0 && wstart < FFTW_P) {
                int cw = FFTW_P - wstart;
                if (cw > peek) cw = peek;
                const __amdgpu_buffer_rsrc_t rw = prc_rsrc(ref, lsf_clampu(cw) * 8u);
                const unsigned voff = vo8 - (unsigned)wstart * 8u;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float2 w = prc_buf_load_c64(rw, voff + 512u * r, 0u);
                    xc[r].x += w.x;
                    xc[r].y += w.y;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            fft1024_fwd(xc, tile, tab, f);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] = cmul(xc[r], Hs[64 * r + lane]);
        __builtin_amdgcn_sched_barrier(0);
        issue_x(p + nwaves);
        {
#ifdef LSF_EXP_NOLOAD
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[r] = make_float2((float)(lane - n0), (float)(r + cnt));
#else
            const __amdgpu_buffer_rsrc_t rs = prc_rsrc(srv + n0, lsf_clampu(cnt) * 8u);
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[r] = prc_buf_load_c64(rs, vslot + 512u * r, 0u);
#endif
        }
        // one rotation on the way out: from this bin's frame to the frame of whoever reads the stream next
        const bool rot_out = a.rot || a.rot2;
        float2 obase = make_float2(1.f, 0.f), ibase = make_float2(1.f, 0.f);
        if (rot_out) {
            const int64_t idx = (int64_t)n0 - ext + lane + peek;
            const float2 p1 = a.rot ? phase_rot(a.pr, idx) : make_float2(1.f, 0.f);
            float2 p2 = a.rot2 ? phase_rot(a.pr2, idx) : make_float2(1.f, 0.f);
            p2.y = -p2.y;
            obase = cmul(p1, p2);
            if (ROT_IN) ibase = make_float2(p1.x, -p1.y);
        }
        __builtin_amdgcn_sched_barrier(0);
#ifndef LSF_EXP_NOFFT       // timing ablation only (wrong results): no transforms, the memory pattern alone
        fft1024_inv<true>(y, tile, tab, f);
#endif
        // last `peek` outputs of the block: rho samples whose ramp restarted carry gamma instead of 1
        if (a.rot && peek > 0 && n0 + cnt > n - peek) {
            const float2 g1 = a.gamma_m1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = n0 + 64 * r + lane - ext;
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
        const __amdgpu_buffer_rsrc_t ro = prc_rsrc(out + n0, lsf_clampu(cnt) * 8u);
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
            prc_v2u d;
            d.x = __float_as_uint(o.x);
            d.y = __float_as_uint(o.y);
            __builtin_amdgcn_raw_buffer_store_b64(d, ro, (int)(vslot + 512u * r), 0, 0);
            // the stored piece (already in the next bin's frame) stays in registers as that bin's correlation
            // input: slots [ext, ext+cnt) only (the other slots of y are circular-convolution garbage)
            const int idx = 64 * r + lane;
            const bool in = idx >= ext && idx < ext + cnt;
            y[r] = in ? o : make_float2(0.f, 0.f);
        }
        if (a.has_next) {
#ifndef LSF_EXP_NOFFT
            fft1024_fwd(y, tile, tab, f);
#endif
#pragma unroll
            for (int m = 0; m < 16; ++m) cmac_bconj(wrs[m], y[m], xc[m]);
        }
    }
    if (a.has_next) {
        fft1024_inv(wrs, tile, tab, f);
        float2* __restrict__ part = a.partial + ((int64_t)b * nwaves + wg) * 2 * T;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lag = 64 * r + lane;
            if (lag < T) part[T + lag] = make_float2(wrs[r].x * sc, -wrs[r].y * sc);
        }
    }
}

// samples per piece: what a 1024-point transform holds next to T - 1 samples of history.  (Line-aligned pieces -- T - 1 a
// multiple of 16 samples -- measured 2 % faster in the fused kernel, tools/ls_align_probe.py: not worth a second slot origin.)
static inline int64_t ls_piece(int T) { return FFTW_P - (T - 1); }

int64_t ls_cache_elems_per_block(int64_t n, int T) {
    const int64_t B = ls_piece(T);
    return ((n + B - 1) / B) * FFTW_P;
}

bool ls_fft_supported(int T) { return T >= 2 && T - 1 <= 768; }

int ls_fft_waves_per_block(int64_t n, int T) {
    // enough waves to fill the chip with a few blocks in flight, but >= ~8 pieces per wave so the
    // two inverse FFTs per wave stay in the noise
    const int64_t B = ls_piece(T);
    const int64_t pieces = (n + B - 1) / B;
    int64_t groups = pieces / (8 * LSF_WAVES);
    if (groups < 1) groups = 1;
    if (groups > 16) groups = 16;
    return (int)groups * LSF_WAVES;
}

static void fill_common(LsFftArgs& a, int T, double theta) {
    a.piece = (int)ls_piece(T);
    a.theta32 = (float)theta;
    for (int r = 0; r < 16; ++r) {
        const double ang = theta * 64.0 * r;
        a.step[r] = make_float2((float)cos(ang), (float)sin(ang));
    }
}

int ls_launch_corr_fft(LsFftArgs a, double theta, int waves_per_block, int nblocks, bool with_autocorr,
                       hipStream_t stream) {
    fill_common(a, a.T, theta);
    int rc = fftw_device_tables(&a.tab);
    if (rc) return rc;
    dim3 grid((unsigned)(waves_per_block / LSF_WAVES), (unsigned)nblocks);
    const size_t lds = sizeof(float2) * (FFTW_TABLE + LSF_WAVES * FFTW_TILE);
    if (!a.circular) {
        if (with_autocorr)
            hipLaunchKernelGGL(ls_corr_fft_lin_kernel<true>, grid, dim3(64 * LSF_WAVES), lds, stream, a);
        else
            hipLaunchKernelGGL(ls_corr_fft_lin_kernel<false>, grid, dim3(64 * LSF_WAVES), lds, stream, a);
    } else if (with_autocorr)
        hipLaunchKernelGGL(ls_corr_fft_kernel<true>, grid, dim3(64 * LSF_WAVES), lds, stream, a);
    else
        hipLaunchKernelGGL(ls_corr_fft_kernel<false>, grid, dim3(64 * LSF_WAVES), lds, stream, a);
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}

int ls_launch_fir_fft(LsFftArgs a, double theta, int nblocks, hipStream_t stream) {
    fill_common(a, a.T, theta);
    int rc = fftw_device_tables(&a.tab);
    if (rc) return rc;
    const int64_t B = a.piece;
    const int64_t pieces = (a.n + B - 1) / B;
    int64_t groups = (pieces + 16 * LSF_WAVES - 1) / (16 * LSF_WAVES);   // ~16 blocks per wave: the FFT of the taps costs 1/33
    if (groups < 1) groups = 1;
    dim3 grid((unsigned)groups, (unsigned)nblocks);
    const size_t lds = sizeof(float2) * (FFTW_TABLE + LSF_WAVES * FFTW_TILE);
    if (!a.circular)
        hipLaunchKernelGGL(ls_fir_fft_lin_kernel, grid, dim3(64 * LSF_WAVES), lds, stream, a);
    else
        hipLaunchKernelGGL(ls_fir_fft_kernel, grid, dim3(64 * LSF_WAVES), lds, stream, a);
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}

int ls_launch_corr_cached(LsFftArgs a, double theta, int waves_per_block, int nblocks, hipStream_t stream) {
    fill_common(a, a.T, theta);
    int rc = fftw_device_tables(&a.tab);
    if (rc) return rc;
    dim3 grid((unsigned)(waves_per_block / LSF_WAVES), (unsigned)nblocks);
    const size_t lds = sizeof(float2) * (FFTW_TABLE + LSF_WAVES * FFTW_TILE + LSF_WAVES * FFTW_P);
    { int rc_ = prc_lds_optin(reinterpret_cast<const void*>(&ls_corr_cached_kernel), 80 * 1024); if (rc_) return rc_; }
    hipLaunchKernelGGL(ls_corr_cached_kernel, grid, dim3(64 * LSF_WAVES), lds, stream, a);
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}

int ls_launch_fused_cached(LsFftArgs a, double theta, double theta_out, double gamma_angle, int waves_per_block,
                           int nblocks, hipStream_t stream) {
    fill_common(a, a.T, theta);
    for (int r = 0; r < 16; ++r) {
        const double ang_in = theta * 64.0 * r, ang = (theta - theta_out) * 64.0 * r;
        a.step2[r] = make_float2((float)cos(ang_in), (float)sin(ang_in));
        a.step[r] = make_float2((float)cos(ang), (float)sin(ang));
    }
    a.gamma_m1 = make_float2((float)(cos(gamma_angle) - 1.0), (float)sin(gamma_angle));
    int rc = fftw_device_tables(&a.tab);
    if (rc) return rc;
    dim3 grid((unsigned)(waves_per_block / LSF_WAVES), (unsigned)nblocks);
    const size_t lds = sizeof(float2) * (FFTW_TABLE + LSF_WAVES * FFTW_TILE + FFTW_P);
    const dim3 block(64 * LSF_WAVES);
    if (a.cache) {
        if (a.rot_in) hipLaunchKernelGGL((ls_fused_cached_kernel<true, true>), grid, block, lds, stream, a);
        else hipLaunchKernelGGL((ls_fused_cached_kernel<true, false>), grid, block, lds, stream, a);
    } else {
        if (a.rot_in) hipLaunchKernelGGL((ls_fused_cached_kernel<false, true>), grid, block, lds, stream, a);
        else hipLaunchKernelGGL((ls_fused_cached_kernel<false, false>), grid, block, lds, stream, a);
    }
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}
