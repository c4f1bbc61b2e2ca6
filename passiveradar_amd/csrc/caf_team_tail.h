// Shared by the 4096-point CAF segment kernels (caf_fft_team.hip, caf_fft_team_multi.hip).
#pragma once
#include "caf_internal.h"
#include "fft_team.h"

#define CAFT_TAIL_MAX 16      // a last piece of at most this many samples is added directly after the inverse transform

// Direct lag products of the `tail` samples after the last full piece (acc is the unnormalised inverse transform,
// x 4096).  Raw buffer loads over the frame's surveillance samples: an index >= n_valid, a lag outside this block
// or beyond range_bins reads as zero from the hardware range check -- no 64-bit addresses, no selects.
template <bool HAS_WIN>
__device__ __forceinline__ void caft_tail(float2 (&acc)[16], const float2* __restrict__ ref,
                                          const float2* __restrict__ srv, const float* __restrict__ win, int hi_f,
                                          int tail, int L0, int LB, int R, int N, int NV, int t) {
    const __amdgpu_buffer_rsrc_t rs = prc_rsrc(srv, (unsigned)NV * 8u);
    for (int i = 0; i < tail; ++i) {
        const int n1 = hi_f + 1 + i;
        float2 uu = make_float2(0.f, 0.f);
        if (n1 < NV) {
            uu = ref[n1];
            if (HAS_WIN) { const float w = win[n1]; uu.x *= w; uu.y *= w; }
        }
        uu.x *= (float)FT_P;
        uu.y *= (float)FT_P;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int within = 256 * r + t;
            const int lag = L0 + within;
            int idx = n1 + lag;
            if (idx >= N) idx -= N;
            const bool ok = within < LB && lag <= R;
            const float2 sv = prc_buf_load_c64(rs, ok ? (unsigned)idx * 8u : 0xFFFFFFF0u, 0u);
            cmac_conj_a(acc[r], uu, sv);
        }
    }
}

