// Doppler stage of fast_xambg (scipy.fftpack.fft(axis=0) + np.fft.fftshift, range_doppler_processing.py:89) as one
// column-FFT kernel over the row-major slow-time buffer; phases and index algebra in doppler_col.h.
#include "caf_internal.h"
#include "doppler_col.h"
#include <math.h>

template <int F>
__global__ __launch_bounds__(DopCfg<F>::THREADS) void doppler_col_kernel(const float2* __restrict__ y,
                                                                         float2* __restrict__ out,
                                                                         const float2* __restrict__ tw, int cols) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dop_smem[];
    float2* lds = reinterpret_cast<float2*>(dop_smem);
    constexpr int Q = DopCfg<F>::Q, KT = DopCfg<F>::KT, F3 = DopCfg<F>::F3;
    const int c = threadIdx.x % KT, p = threadIdx.x / KT;
    const int k = blockIdx.x * KT + c;
    const bool live = k < cols;
    const int64_t base = (int64_t)blockIdx.y * F * cols + k;
    float2 x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
        x[r] = live ? y[base + (int64_t)(r * Q + p) * cols] : make_float2(0.f, 0.f);
    dop_stage1<F>(x, tw, p);
    dop_write1<F>(x, lds, p, c);
    __syncthreads();
    dop_read1<F>(x, lds, p, c);
    dop_stage2<F>(x, tw, p);
    if (F3 > 1) {
        dop_write2<F>(x, lds, p, c);      // the slots this thread just read: no barrier in between
        __syncthreads();
        dop_read2<F>(x, lds, p, c);
        dop_stage3<F>(x);
    }
    if (live) {
#pragma unroll
        for (int m = 0; m < 16; ++m) out[base + (int64_t)dop_out_row<F>(p, m) * cols] = x[m];
    }
}

void dop_make_table(float2* t, int F) {
    const double PI = 3.14159265358979323846;
    for (int m = 0; m < F; ++m) {
        const double a = -2.0 * PI * (double)m / (double)F;
        t[m] = make_float2((float)cos(a), (float)sin(a));
    }
}

bool dop_supported(int F) { return F == 256 || F == 512 || F == 1024 || F == 2048 || F == 4096; }

template <int F>
static int dop_launch_t(const float2* y, float2* out, const float2* tw, int cols, int nframes, hipStream_t stream) {
    constexpr int KT = DopCfg<F>::KT;
    const size_t lds = sizeof(float2) * DopCfg<F>::LDS_ELEMS;
    static bool attr_done[16] = {false};
    int dev = 0;
    PRC_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16 || !attr_done[dev]) {
        PRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&doppler_col_kernel<F>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev >= 0 && dev < 16) attr_done[dev] = true;
    }
    dim3 grid((unsigned)((cols + KT - 1) / KT), (unsigned)nframes);
    hipLaunchKernelGGL((doppler_col_kernel<F>), grid, dim3(DopCfg<F>::THREADS), lds, stream, y, out, tw, cols);
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}

int dop_launch(const float2* y, float2* out, const float2* tw, int F, int cols, int nframes, hipStream_t stream) {
    switch (F) {
        case 256: return dop_launch_t<256>(y, out, tw, cols, nframes, stream);
        case 512: return dop_launch_t<512>(y, out, tw, cols, nframes, stream);
        case 1024: return dop_launch_t<1024>(y, out, tw, cols, nframes, stream);
        case 2048: return dop_launch_t<2048>(y, out, tw, cols, nframes, stream);
        case 4096: return dop_launch_t<4096>(y, out, tw, cols, nframes, stream);
    }
    prc_set_error("dop_launch: unsupported freq_bins %d", F);
    return PRC_EUNSUPPORTED;
}
