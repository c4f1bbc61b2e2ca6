// Shared by the two translation units of the cached-spectrum chain on the 4096-point team transform
// (ls_fft_team_corr_cached.hip, ls_fft_team_cached.hip): they are compiled with different twiddle storage (Makefile).
#pragma once
#include "ls_internal.h"
#include "fft_team.h"
#include <math.h>

#define LTC_SPEC FT_P                        // float2 of per-thread spectrum kept in LDS behind the transform's area
static constexpr size_t LTC_LDS = sizeof(float2) * (FT_LDS_ELEMS + LTC_SPEC);

__device__ __forceinline__ unsigned ltc_clampu(int x) { return x < 0 ? 0u : (unsigned)x; }

__device__ __forceinline__ void ltc_cmac_bconj(float2& w, float2 u, float2 x) {   // w += u * conj(x)
    w.x = fmaf(u.x, x.x, w.x);
    w.x = fmaf(u.y, x.y, w.x);
    w.y = fmaf(u.y, x.x, w.y);
    w.y = fmaf(-u.x, x.y, w.y);
}

static inline void ltc_fill(LsFftArgs& a, double theta) {
    a.piece = FT_P - (a.T - 1);
    a.theta32 = (float)theta;
    for (int r = 0; r < 16; ++r) {
        const double ang = theta * (double)FT_THREADS * r;
        a.step[r] = make_float2((float)cos(ang), (float)sin(ang));
    }
}
