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
    // fused FIR(bin i) + correlation(bin i+1) kernel.  The streams between the bins of a chain are private to
    // it and are kept in the ROTATED frame of the bin that reads them (value = s e^{-j phi_bin(n)}): the kernel
    // subtracts the unrotated FIR output directly and applies ONE rotation e^{j(phi_i - phi_out)} on the way out
    // (phi_out: the next bin's ramp, or 0 for the last bin) instead of one for the FIR and one for the next
    // correlation.  pr2/rot2 describe phi_out; step[] is overwritten with exp(j (theta_i - theta_out) 64 r);
    // rot_in: the input is still the caller's raw stream and bin i is rotated (first bin only), step2[] =
    // exp(j theta_i 64 r) then.
    int32_t has_next, rot2, rot_in;
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
int ls_launch_fused_cached(LsFftArgs a, double theta, double theta_out, double gamma_angle, int waves_per_block,
                           int nblocks, hipStream_t stream);
// 4096-point team kernels (ls_fft_team.hip): 770 .. 3073 taps, per-bin correlate -> Levinson -> FIR
bool ls_team_supported(int T);
int ls_team_teams_per_block(int64_t n, int T);
int ls_launch_corr_team(LsFftArgs a, double theta, int teams_per_block, int nblocks, bool with_autocorr,
                        hipStream_t stream);
int ls_launch_fir_team(LsFftArgs a, double theta, int nblocks, hipStream_t stream);
// cached-spectrum chain on the 4096-point transform (ls_fft_team_cached.hip); cache: [block][npieces][4096]
int ls_team_piece(int T, int align);            // samples per piece of the chain (align: PRC_OPT_LS_TEAM_ALIGN)
int64_t ls_team_cache_elems_per_block(int64_t n, int piece);
int ls_team_chain_teams_per_block(int64_t n, int piece, int max_blocks, int pieces_per_team);
int ls_launch_corr_cached_team(LsFftArgs a, double theta, int teams_per_block, int nblocks, hipStream_t stream);
int ls_launch_fused_cached_team(LsFftArgs a, double theta, double theta_out, double gamma_angle, int teams_per_block,
                                int nblocks, hipStream_t stream);
