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
#ifdef FT_PK
    v2f ww = pk_from(w);
    pk_cmac_bconj(ww, pk_from(u), pk_from(x));
    w = pk_to(ww);
#else
    w.x = fmaf(u.x, x.x, w.x);
    w.x = fmaf(u.y, x.y, w.x);
    w.y = fmaf(u.y, x.x, w.y);
    w.y = fmaf(-u.x, x.y, w.y);
#endif
}

// Samples per piece.  The slot origin E (history in slots [0, E), the piece in [E, E + B)) is T - 1 rounded up to 16
// samples, so that every piece starts on a 128-byte line of the streams (tools/ubench/stream4.hip: +4 .. 8 % of HBM rate
// for the same bytes); align = 0 keeps E = T - 1 (A/B runs).  The plan decides ONCE (PRC_OPT_LS_TEAM_ALIGN at
// prc_ls_plan_create) and hands the piece length to both kernels through LsFftArgs.piece, so the cache it sized and
// the pieces the kernels cut can never disagree.
static inline int ltc_piece(int T, int align) {
    const int E = align ? ((T - 1 + 15) & ~15) : T - 1;
    return FT_P - E;
}

// All loads issued so far have landed.  Placed between the first prefetch and a loop that prefetches AND stores: without
// it the loop head is reached with loads pending (first entry) or with loads followed by stores pending (back edge), the
// compiler's wait has to cover both, and it becomes an s_waitcnt vmcnt(0) on every iteration -- which also waits for the
// stores of the previous piece (vmcnt counts loads and stores in order on gfx9).  tools/ubench/stream4.hip: +5 % of HBM rate.
__device__ __forceinline__ void ltc_loads_landed() {
#ifndef LTC_NO_WAIT_FIRST
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), expcnt and lgkmcnt untouched
#endif
}

static inline void ltc_fill(LsFftArgs& a, double theta) {      // a.piece: set by the plan (fill_xa)
    a.theta32 = (float)theta;
    for (int r = 0; r < 16; ++r) {
        const double ang = theta * (double)FT_THREADS * r;
        a.step[r] = make_float2((float)cos(ang), (float)sin(ang));
    }
}
