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
    int64_t ref_stride, srv_stride, out_stride;
    int64_t n;
    int32_t T, peek, circular, rot, piece;
    PhaseRamp pr;
    float theta32;           // 2 pi fc / Fs (phase of the <= peek samples that wrapped to index 0)
    float2 step[16];         // exp(j theta 64 r): per-register phase step of the Doppler rotation
};

bool ls_fft_supported(int T);
int ls_fft_waves_per_block(int64_t n, int T);
int ls_launch_corr_fft(LsFftArgs a, double theta, int waves_per_block, int nblocks, bool with_autocorr,
                       hipStream_t stream);
int ls_launch_fir_fft(LsFftArgs a, double theta, int nblocks, hipStream_t stream);
