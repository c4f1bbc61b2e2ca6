// Shared by the 4096-point CAF segment kernels (caf_fft_team.hip, caf_fft_team_multi.hip).
#pragma once
#include "caf_internal.h"
#include "fft_team.h"

struct CafTeamArgs {
    CafSegArgs s;
    const float2* gtab;
    int32_t piece;      // B = 4097 - lagblk samples of ref per transform
    int32_t lagblk;     // lags per block (<= 3073)
    int32_t nlagblk;    // lag blocks covering 0..range_bins
    int32_t segs;       // consecutive slow-time samples per workgroup
    // several reference channels in ONE launch (nothing shared between them but the L2 -- "turns" without the tail of four
    // small launches): channel z reads refs[z] and writes its surfaces y_ref_stride elements further on
    const float2* refs[PRC_CAF_MAX_REFS];
    int64_t y_ref_stride;
    int32_t nref, chunks_x, nchunks;   // channels; workgroup chunks per frame; chunks_x * nframes
    int32_t xcd_contig;                // 1: an XCD takes a contiguous run of chunks; 0: chunks go round the XCDs in launch order
    int32_t pair_half;                 // > 0: frames overlap by half (= this many chunks): the two frames that cover the same
                                       // samples run in consecutive slots of one XCD (PRC_OPT_CAF_PAIR_FRAMES)
    int32_t nframes;
};

// the same segment kernel on the eight-wavefront transform of fft_team8.h (caf_fft_team8.hip); `a` as prepared by
// caf_launch_fft_team_refs
int caf_launch_fft_team8(const CafTeamArgs& a, dim3 grid, bool has_window, hipStream_t stream);

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

