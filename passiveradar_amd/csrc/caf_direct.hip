// Time-domain cross-ambiguity segment sums (the K2+K3 steps of SURVEY 2.3).
//
// Replaces the hot loop of the reference's fast_xambg, range_doppler_processing.py:81-86:
//   for each lag:  roll(conj(srv), lag) * ref * window  ->  scipy.signal.decimate(.., q, FIR)
// i.e.  y[j, k] = sum_m h[m] p_k[j q + half - m],  p_k[n] = w[n] ref[n] conj(srv[(n + R-k) mod N]).
//
// One workgroup per decimated slow-time sample j (per frame): the FIR window of segment j is
// streamed through LDS in tiles; P[i] = h*w*ref is broadcast-read, S[i + lag] is read with
// consecutive lanes on consecutive lags (conflict-free ds_read_b64), 4 wavefronts split the
// tile and are reduced through LDS at the end.  Arithmetic: fp32 FMA, VALU-bound (no MFMA:
// the Hankel structure gives each srv sample to one lag-product only once per segment).
#include "caf_internal.h"

#define CAFD_TILE 1024
#define CAFD_THREADS 256

template <int NLG>
__global__ __launch_bounds__(CAFD_THREADS) void caf_direct_kernel(CafSegArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* P = reinterpret_cast<float2*>(smem_raw);           // CAFD_TILE
    float2* S = P + CAFD_TILE;                                  // CAFD_TILE + 64*NLG
    float2* red = S + CAFD_TILE + 64 * NLG;                     // 4 * NLG * 64

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int64_t j = blockIdx.x;
    const int b = blockIdx.y;
    const float2* __restrict__ ref = a.ref + (int64_t)b * a.frame_stride;
    const float2* __restrict__ srv = a.srv + (int64_t)b * a.frame_stride;

    const int64_t n_hi = j * a.q + a.half;          // last sample of the FIR window (tap 0)
    const int64_t n_lo = n_hi - (a.ntaps - 1);      // first sample (tap ntaps-1)
    const int64_t lo_c = n_lo < 0 ? 0 : n_lo;
    const int64_t hi_c = n_hi > a.n - 1 ? a.n - 1 : n_hi;
    const int nlags = a.range_bins + 1;

    for (int L0 = 0; L0 < nlags; L0 += 64 * NLG) {
        float2 acc[NLG];
#pragma unroll
        for (int g = 0; g < NLG; ++g) acc[g] = make_float2(0.f, 0.f);

        for (int64_t t0 = lo_c; t0 <= hi_c; t0 += CAFD_TILE) {
            const int64_t rem = hi_c - t0 + 1;
            const int cnt = rem < CAFD_TILE ? (int)rem : CAFD_TILE;
            // stage P = taps * window * ref (zero beyond the segment / beyond n_valid)
            for (int i = tid; i < CAFD_TILE; i += CAFD_THREADS) {
                float2 v = make_float2(0.f, 0.f);
                const int64_t n = t0 + i;
                if (i < cnt && n < a.n_valid) {
                    v = ref[n];
                    float g = 1.f;
                    if (a.window) g = a.window[n];
                    if (a.taps) g *= a.taps[n_hi - n];
                    v.x *= g;
                    v.y *= g;
                }
                P[i] = v;
            }
            // stage S = srv with circular wrap inside the frame (np.roll, :82)
            for (int i = tid; i < CAFD_TILE + 64 * NLG; i += CAFD_THREADS) {
                int64_t idx = t0 + i + L0;
                if (idx >= a.n) idx %= a.n;
                float2 v = make_float2(0.f, 0.f);
                if (idx < a.n_valid) v = srv[idx];
                S[i] = v;
            }
            __syncthreads();
            const int i0 = wave * (CAFD_TILE / 4);
            int i1 = i0 + CAFD_TILE / 4;
            if (i1 > cnt) i1 = cnt;
            const float2* Sl = S + lane;
#pragma unroll 4
            for (int i = i0; i < i1; ++i) {
                const float2 p = P[i];
#pragma unroll
                for (int g = 0; g < NLG; ++g) cmac_conj(acc[g], p, Sl[i + 64 * g]);
            }
            __syncthreads();
        }
        // cross-wave reduction and store
#pragma unroll
        for (int g = 0; g < NLG; ++g) red[(wave * NLG + g) * 64 + lane] = acc[g];
        __syncthreads();
        for (int t = tid; t < 64 * NLG; t += CAFD_THREADS) {
            float2 s0 = red[t];
            const float2 s1 = red[NLG * 64 + t], s2 = red[2 * NLG * 64 + t],
                         s3 = red[3 * NLG * 64 + t];
            s0.x += s1.x + s2.x + s3.x;
            s0.y += s1.y + s2.y + s3.y;
            const int lag = L0 + t;
            if (lag < nlags) {
                const int k = a.range_bins - lag;
                caf_store_y(a, b, j, k, s0);
            }
        }
        __syncthreads();
    }
}

int caf_launch_direct(const CafSegArgs& a, int nframes, hipStream_t stream) {
    const int nlags = a.range_bins + 1;
    dim3 grid((unsigned)a.freq_bins, (unsigned)nframes);
    // lag groups per pass: enough for the whole lag span up to 8 (512 lags), else loop
    int nlg = (nlags + 63) / 64;
    if (nlg > 8) nlg = 8;
#define PRC_CAFD_CASE(G)                                                                    \
    case G: {                                                                               \
        size_t lds = sizeof(float2) * (CAFD_TILE + CAFD_TILE + 64 * G + 4 * G * 64);        \
        hipLaunchKernelGGL(caf_direct_kernel<G>, grid, dim3(CAFD_THREADS), lds, stream, a); \
    } break;
    switch (nlg) {
        PRC_CAFD_CASE(1)
        PRC_CAFD_CASE(2)
        PRC_CAFD_CASE(3)
        PRC_CAFD_CASE(4)
        PRC_CAFD_CASE(5)
        PRC_CAFD_CASE(6)
        PRC_CAFD_CASE(7)
        PRC_CAFD_CASE(8)
    }
#undef PRC_CAFD_CASE
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}
