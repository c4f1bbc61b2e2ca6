// FFT-domain segment correlation + fused Doppler FFT (filled in after the direct path is green).
#include "caf_internal.h"

bool caf_fft_supported(int64_t, int, int, int) { return false; }
bool caf_doppler_fused_supported(int) { return false; }
int caf_launch_fft(const CafSegArgs&, int, hipStream_t) {
    prc_set_error("FFT segment method not built");
    return PRC_EUNSUPPORTED;
}
int caf_launch_doppler_fused(const float2*, float2*, int, int, int, hipStream_t) {
    prc_set_error("fused Doppler FFT not built");
    return PRC_EUNSUPPORTED;
}
