// Sample-recursive NLMS clutter canceller, one wavefront per independent stream.
//
// Replaces NLMS_filter, clutter_removal.py:189-249:
//   u_k[i] = ref[L+k+peek-i], i = 0..T-1;  e = srv[k+L] - w^H u;  w += mu u conj(e) / (u^H u)
// The recursion is strictly sequential in k; the parallelism is the T taps (spread over the 64
// lanes of ONE wavefront so that no workgroup barrier sits on the critical path) and the
// independent streams (one wavefront each, >= 4 per CU).  This is a latency/VALU-bound wavefront
// dot-product + AXPY loop -- not HBM-bound (24 B/sample) and not MFMA-shaped.
#include "common.h"

#define NLMS_KT 1024   // steps per staged window

struct NlmsArgs {
    const float2* ref;
    const float2* srv;
    const float2* taps_in;   // [nstreams][T] or nullptr
    float2* out;
    float2* taps_out;        // [nstreams][T] or nullptr
    int64_t n, stride, out_stride;
    int32_t L, peek, T;
    float mu;
};

// All-lanes sum by DPP (no LDS crossbar): xor-1 / xor-2 inside quads, half-row mirror, row mirror
// -> every row of 16 lanes holds its row sum; the four row sums are read with v_readlane.
__device__ __forceinline__ float dpp_f(float x, const int which) {
    const int i = __float_as_int(x);
    int r;
    switch (which) {
        case 0: r = __builtin_amdgcn_mov_dpp(i, 0xB1, 0xF, 0xF, true); break;    // quad_perm [1,0,3,2]
        case 1: r = __builtin_amdgcn_mov_dpp(i, 0x4E, 0xF, 0xF, true); break;    // quad_perm [2,3,0,1]
        case 2: r = __builtin_amdgcn_mov_dpp(i, 0x141, 0xF, 0xF, true); break;   // row_half_mirror
        default: r = __builtin_amdgcn_mov_dpp(i, 0x140, 0xF, 0xF, true); break;  // row_mirror
    }
    return __int_as_float(r);
}
__device__ __forceinline__ float wave_allsum(float v) {
    v += dpp_f(v, 0);
    v += dpp_f(v, 1);
    v += dpp_f(v, 2);
    v += dpp_f(v, 3);
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (a + b) + (c + d);
}

template <int TPL>
__global__ __launch_bounds__(64) void nlms_kernel(NlmsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* Rw = reinterpret_cast<float2*>(smem_raw);   // NLMS_KT + 64*TPL : ref window
    float2* D = Rw + NLMS_KT + 64 * TPL;                // NLMS_KT : srv in, error out
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const float2* __restrict__ ref = a.ref + (int64_t)b * a.stride;
    const float2* __restrict__ srv = a.srv + (int64_t)b * a.stride;
    float2* __restrict__ out = a.out + (int64_t)b * a.out_stride;
    const int T = a.T;
    const int64_t nsteps = a.n - T;   // k = 0..nsteps-1 (may be <= 0)

    float2 w[TPL];
#pragma unroll
    for (int t = 0; t < TPL; ++t) {
        const int i = lane + 64 * t;
        w[t] = (a.taps_in && i < T) ? a.taps_in[(int64_t)b * T + i] : make_float2(0.f, 0.f);
    }
    // out[0:L] = 0 and out[n-peek:] = 0 (:231)
    for (int64_t i = lane; i < a.n; i += 64)
        if (i < a.L || i >= a.L + (nsteps > 0 ? nsteps : 0)) out[i] = make_float2(0.f, 0.f);

    const int WIN = 64 * TPL;   // >= T; taps i >= T are held at zero through zero u
    for (int64_t k0 = 0; k0 < nsteps; k0 += NLMS_KT) {
        const int64_t rem = nsteps - k0;
        const int cnt = rem < NLMS_KT ? (int)rem : NLMS_KT;
        __syncthreads();
        // window element x <-> ref[k0 + 1 - (WIN - T) + x];  u_kk[i] = Rw[kk + WIN - 1 - i]
        const int64_t base = k0 + 1 - (WIN - T);
        for (int x = lane; x < NLMS_KT + WIN; x += 64) {
            const int64_t idx = base + x;
            Rw[x] = (idx >= 0 && idx < a.n) ? ref[idx] : make_float2(0.f, 0.f);
        }
        for (int x = lane; x < cnt; x += 64) D[x] = srv[k0 + x + a.L];
        __syncthreads();
        // u^H u is summed exactly at the start of every staged window and then slid:
        //   E(k+1) = E(k) + |ref[T+k+1]|^2 - |ref[k+1]|^2      (window element kk+WIN enters, kk+WIN-T leaves)
        float energy;
        {
            float e0 = 0.f;
#pragma unroll
            for (int t = 0; t < TPL; ++t) {
                const int i = lane + 64 * t;
                const float2 v = Rw[WIN - 1 - i];
                if (i < T) e0 = fmaf(v.x, v.x, fmaf(v.y, v.y, e0));
            }
            energy = wave_allsum(e0);
        }
        // One step: dot (registers), two DPP reductions, AXPY.  The sliding window of the NEXT step
        // is fetched from LDS while this step reduces (ua/ub swap roles, loop unrolled by two), so
        // the only latency left on the critical path is the reduction itself.
        auto fetch = [&](float2 (&u)[TPL], int kk) {
#pragma unroll
            for (int t = 0; t < TPL; ++t) {
                const int i = lane + 64 * t;
                float2 v = Rw[kk + WIN - 1 - i];
                if (t == TPL - 1 && i >= T) v = make_float2(0.f, 0.f);   // only the last group can overhang T
                u[t] = v;
            }
        };
        auto step = [&](const float2 (&u)[TPL], float2 (&unext)[TPL], int kk) {
            float yr = 0.f, yi = 0.f;
#pragma unroll
            for (int t = 0; t < TPL; ++t) {          // conj(w) * u
                yr = fmaf(w[t].x, u[t].x, yr);
                yr = fmaf(w[t].y, u[t].y, yr);
                yi = fmaf(w[t].x, u[t].y, yi);
                yi = fmaf(-w[t].y, u[t].x, yi);
            }
            fetch(unext, kk + 1);                    // independent of this step's result
            yr = wave_allsum(yr);
            yi = wave_allsum(yi);
            const float en = energy;
            {   // slide the energy to the next step (wave-uniform LDS reads, broadcast)
                const float2 vin = Rw[kk + WIN], vout = Rw[kk + WIN - T];
                energy += (vin.x * vin.x + vin.y * vin.y) - (vout.x * vout.x + vout.y * vout.y);
            }
            const float2 d = D[kk];
            const float er = d.x - yr, ei = d.y - yi;
            const float s = a.mu / en;               // coefficient mu * conj(e) / (u^H u)
            const float cr = er * s, ci = -ei * s;
#pragma unroll
            for (int t = 0; t < TPL; ++t) {
                w[t].x = fmaf(cr, u[t].x, w[t].x);
                w[t].x = fmaf(-ci, u[t].y, w[t].x);
                w[t].y = fmaf(cr, u[t].y, w[t].y);
                w[t].y = fmaf(ci, u[t].x, w[t].y);
            }
            if (lane == 0) D[kk] = make_float2(er, ei);
        };
        float2 ua[TPL], ub[TPL];
        fetch(ua, 0);
        int kk = 0;
        for (; kk + 2 <= cnt; kk += 2) {
            step(ua, ub, kk);
            step(ub, ua, kk + 1);
        }
        if (kk < cnt) step(ua, ub, kk);
        __syncthreads();
        for (int x = lane; x < cnt; x += 64) out[a.L + k0 + x] = D[x];
    }
    if (a.taps_out) {
#pragma unroll
        for (int t = 0; t < TPL; ++t) {
            const int i = lane + 64 * t;
            if (i < T) a.taps_out[(int64_t)b * T + i] = w[t];
        }
    }
}

extern "C" int prc_nlms_execute(const void* ref, const void* srv, int64_t n, int64_t stride,
                                int32_t filter_len, int32_t peek, float mu, const void* taps_in,
                                void* out, int64_t out_stride, void* taps_out, int32_t nstreams,
                                void* stream) {
    PRC_REQUIRE(ref && srv && out, PRC_EINVAL, "prc_nlms_execute: null argument");
    PRC_REQUIRE(n > 0 && filter_len > 0 && peek >= 0 && nstreams > 0, PRC_EINVAL,
                "prc_nlms_execute: non-positive size");
    PRC_REQUIRE(stride >= n && out_stride >= n, PRC_ESHAPE, "prc_nlms_execute: stride shorter than n");
    const int T = filter_len + peek;
    const int tpl = (T + 63) / 64;
    PRC_REQUIRE(tpl <= 32, PRC_EUNSUPPORTED,
                "prc_nlms_execute: %d taps exceed the single-wavefront kernel (max 2048)", T);
    NlmsArgs a;
    a.ref = (const float2*)ref;
    a.srv = (const float2*)srv;
    a.taps_in = (const float2*)taps_in;
    a.out = (float2*)out;
    a.taps_out = (float2*)taps_out;
    a.n = n;
    a.stride = stride;
    a.out_stride = out_stride;
    a.L = filter_len;
    a.peek = peek;
    a.T = T;
    a.mu = mu;
#define PRC_NLMS_CASE(G)                                                                        \
    if (tpl <= G) {                                                                             \
        size_t lds = sizeof(float2) * (NLMS_KT + 64 * G + NLMS_KT);                             \
        hipLaunchKernelGGL(nlms_kernel<G>, dim3(nstreams), dim3(64), lds, (hipStream_t)stream, a); \
        PRC_LAUNCH_CHECK();                                                                     \
        return PRC_OK;                                                                          \
    }
    PRC_NLMS_CASE(1)
    PRC_NLMS_CASE(2)
    PRC_NLMS_CASE(3)
    PRC_NLMS_CASE(4)
    PRC_NLMS_CASE(5)
    PRC_NLMS_CASE(6)
    PRC_NLMS_CASE(8)
    PRC_NLMS_CASE(12)
    PRC_NLMS_CASE(17)
    PRC_NLMS_CASE(24)
    PRC_NLMS_CASE(32)
#undef PRC_NLMS_CASE
    return PRC_EUNSUPPORTED;
}
