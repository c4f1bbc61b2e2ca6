// Internal interfaces between the LS translation units.
#pragma once
#include "common.h"

struct LsFftArgs {
    const float2* ref;
    const float2* srv;
    float2* out;             // FIR only
    float2* partial;         // correlation only: [block][wave][2][T]
    const double2* taps;     // FIR only: [block][T] complex128
    const float2* tab;       // FFT twiddle tables (device)
    float2* cache;           // cached-spectrum chain: FFT(rho block) per piece, [block][npieces][1024]
    const double2* taps_t;   // cached-spectrum chain: w~[k] = w[k] e^{-j theta k}, [block][T]
    int64_t ref_stride, srv_stride, out_stride;
    int64_t n;
    int32_t T, peek, circular, rot, piece;
    PhaseRamp pr;
    float theta32;           // 2 pi fc / Fs (phase of the <= peek samples that wrapped to index 0)
    float2 step[16];         // exp(j theta 64 r): per-register phase step of the Doppler rotation
    // fused FIR(bin i) + correlation(bin i+1) kernel: rotation of the NEXT bin and gamma_i - 1
    int32_t has_next, rot2;
    PhaseRamp pr2;
    float2 step2[16];
    float2 gamma_m1;
};

bool ls_fft_supported(int T);
int ls_fft_waves_per_block(int64_t n, int T);
int ls_launch_corr_fft(LsFftArgs a, double theta, int waves_per_block, int nblocks, bool with_autocorr,
                       hipStream_t stream);
int ls_launch_fir_fft(LsFftArgs a, double theta, int nblocks, hipStream_t stream);
// cached-spectrum chain (linear boundary): first bin builds the cache, later bins and every FIR reuse it
int64_t ls_cache_elems_per_block(int64_t n, int T);
int ls_launch_corr_cached(LsFftArgs a, double theta, int waves_per_block, int nblocks, hipStream_t stream);
int ls_launch_fused_cached(LsFftArgs a, double theta, double theta_next, double gamma_angle, int waves_per_block,
                           int nblocks, hipStream_t stream);
