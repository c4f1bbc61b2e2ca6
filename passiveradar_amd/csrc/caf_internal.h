// Internal interfaces between the CAF translation units.
#pragma once
#include "common.h"

// Slow-time buffer layouts written by the segment kernels and read by the Doppler stage.
enum { PRC_Y_JK = 0,   // y[frame][j][k]   (k contiguous)  -> written row-wise by the FFT segment kernel
       PRC_Y_KJ = 1 }; // y[frame][k][j]   (j contiguous)  -> rocFFT batched plan

#define PRC_CAF_MAX_REFS 8

struct CafSegArgs {
    const float2* ref;
    const float2* srv;
    const float* window;   // float32[n] or nullptr
    const float* taps;     // float32[ntaps] or nullptr (boxcar)
    const float* taps_rev; // the same taps in reverse order (tap of sample n of segment j at [n - n_lo]): FFT kernels
    float2* y;             // slow-time buffer
    int64_t frame_stride;  // elements between consecutive frames in ref/srv
    int64_t n;             // CPI length (circular-wrap modulus)
    int64_t n_valid;       // samples >= n_valid read as zero
    int64_t q;             // decimation
    int32_t ntaps;         // FIR length (q+1 for the boxcar)
    int32_t half;          // (ntaps-1)/2
    int32_t range_bins;
    int32_t freq_bins;
    int32_t y_layout;
    // PRC_Y_JK only: 0 = plain rows y[frame][j][k]; kt > 0 (a power of two <= 32) = tiles of kt columns,
    // y[frame][k / kt][j][k % kt] -- the layout the column-FFT Doppler kernel reads: a workgroup's tile (kt columns x all
    // F rows) is ONE contiguous block instead of F row segments of 8 kt bytes a row pitch apart (measured on MI355X,
    // 2048 rows x 8 columns: 5.1 TB/s from contiguous tiles against 3.1 TB/s from row segments)
    int32_t y_kt;
    int64_t y_surface;     // elements per surface of the slow-time buffer (F * ceil(cols / kt) * kt when tiled)
};

// element offset of (frame, slow-time sample j, column k) in a PRC_Y_JK buffer
__device__ __forceinline__ int64_t caf_y_off(const CafSegArgs& a, int frame, int64_t j, int k) {
    if (a.y_kt == 0) return ((int64_t)frame * a.freq_bins + j) * (a.range_bins + 1) + k;
    const int sh = 31 - __builtin_clz((unsigned)a.y_kt);
    return (int64_t)frame * a.y_surface + (((int64_t)(k >> sh) * a.freq_bins + j) << sh) + (k & (a.y_kt - 1));
}

__device__ __forceinline__ void caf_store_y(const CafSegArgs& a, int frame, int64_t j, int k,
                                            float2 v) {
    const int64_t cols = a.range_bins + 1;
    if (a.y_layout == PRC_Y_JK)
        a.y[caf_y_off(a, frame, j, k)] = v;
    else
        a.y[(int64_t)frame * a.freq_bins * cols + (int64_t)k * a.freq_bins + j] = v;
}

int caf_launch_direct(const CafSegArgs& a, int nframes, hipStream_t stream);
int caf_launch_fft(const CafSegArgs& a, int nframes, hipStream_t stream);
bool caf_fft_supported(int64_t n, int range_bins, int freq_bins, int ntaps_is_boxcar);
// 4096-point team transforms (caf_fft_team.hip); the *_blocking functions return the cost of one segment in
// transforms of their own size
int caf_launch_fft_team(const CafSegArgs& a, int nframes, hipStream_t stream);
// the same kernel for nref reference channels in one launch (blockIdx.z; nothing shared): channel i reads refs[i] and writes
// its nframes surfaces at a.y + i * y_ref_stride
int caf_launch_fft_team_refs(const CafSegArgs& a, const float2* const* refs, int nref, int64_t y_ref_stride, int nframes,
                             hipStream_t stream);
bool caf_team_supported(int64_t n, int range_bins, int freq_bins, int ntaps_is_boxcar);
double caf_team_blocking(int64_t q1, int range_bins, int* nlb_out, int* lb_out);
double caf_fft_blocking(int64_t q1, int range_bins, int* nlb_out, int* lb_out);
// nref reference channels against one surveillance channel in one launch (segments of <= 2 pieces): illuminator i's
// surfaces go to y + i * y_ref_stride
int caf_launch_fft_team_multi(const CafSegArgs& a, const float2* const* refs, int nref, int64_t y_ref_stride,
                              int nframes, hipStream_t stream);
bool caf_team_multi_supported(int64_t n, int range_bins, int freq_bins, int64_t q1, int nref);
double caf_team_multi_blocking(int64_t q1, int range_bins, int nref, int* nlb_out, int* lb_out);
int caf_launch_transpose_jk_kj(const float2* src, float2* dst, int freq_bins, int cols, int nframes,
                               hipStream_t stream);
