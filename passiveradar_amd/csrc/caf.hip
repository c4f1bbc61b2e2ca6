// Cross-ambiguity plan: segment sums (caf_direct.hip / caf_fft.hip) + Doppler FFT + fftshift.
//
// Replaces fast_xambg, range_doppler_processing.py:12-90.  The Doppler stage (:89,
// scipy.fftpack.fft(axis=0) then np.fft.fftshift) is either a rocFFT batched 1-D plan over
// the contiguous slow-time axis followed by one shift+transpose kernel.
#include "caf_internal.h"
#include <rocfft/rocfft.h>
#include <vector>

static std::once_flag g_rocfft_once;
static void rocfft_setup_once() {
    std::call_once(g_rocfft_once, [] { rocfft_setup(); });
}

struct prc_caf_plan {
    prc_caf_desc desc;
    int method;   // resolved prc_caf_method
    int doppler;  // resolved prc_doppler_method
    int64_t q;
    int ntaps;
    int half;
    float* d_taps = nullptr;         // device copy of the long decimation FIR (or null)
    float2* d_y = nullptr;           // slow-time buffer, max_frames * F * (R+1)
    float2* d_y2 = nullptr;          // [j][k]-ordered staging written by the FFT segment kernel
    size_t y_bytes = 0;
    rocfft_plan fft = nullptr;       // one batched plan for max_frames
    rocfft_execution_info info = nullptr;
    void* d_work = nullptr;
    size_t work_bytes = 0;
    std::mutex mtx;
};

// out[b][f'][k] = Y[b][k][(f' - F/2) mod F]  (np.fft.fftshift along axis 0, then the
// (F, R+1) C-order layout of the reference's xambg array, :64,:89)
__global__ __launch_bounds__(256) void shift_transpose_kernel(const float2* __restrict__ yT,
                                                              float2* __restrict__ out, int F,
                                                              int cols) {
    __shared__ float2 tile[32][33];
    const int b = blockIdx.z;
    const int64_t base = (int64_t)b * F * cols;
    const int f0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, f = f0 + tx;
        if (k < cols && f < F) tile[r][tx] = yT[base + (int64_t)k * F + f];
    }
    __syncthreads();
    const int sh = F / 2;
    for (int r = ty; r < 32; r += 8) {
        const int f = f0 + r, k = k0 + tx;
        if (f < F && k < cols) {
            int fo = f + sh;
            if (fo >= F) fo -= F;
            out[base + (int64_t)fo * cols + k] = tile[tx][r];
        }
    }
}

static int build_rocfft(prc_caf_plan* p, int frames) {
    rocfft_setup_once();
    size_t len = (size_t)p->desc.freq_bins;
    size_t batch = (size_t)frames * (size_t)(p->desc.range_bins + 1);
    rocfft_status st = rocfft_plan_create(&p->fft, rocfft_placement_inplace,
                                          rocfft_transform_type_complex_forward,
                                          rocfft_precision_single, 1, &len, batch, nullptr);
    PRC_REQUIRE(st == rocfft_status_success, PRC_EROCFFT, "rocfft_plan_create failed (%d)", (int)st);
    st = rocfft_plan_get_work_buffer_size(p->fft, &p->work_bytes);
    PRC_REQUIRE(st == rocfft_status_success, PRC_EROCFFT, "rocfft work size failed (%d)", (int)st);
    st = rocfft_execution_info_create(&p->info);
    PRC_REQUIRE(st == rocfft_status_success, PRC_EROCFFT, "rocfft info create failed (%d)", (int)st);
    if (p->work_bytes) {
        PRC_HIP(hipMalloc(&p->d_work, p->work_bytes));
        st = rocfft_execution_info_set_work_buffer(p->info, p->d_work, p->work_bytes);
        PRC_REQUIRE(st == rocfft_status_success, PRC_EROCFFT, "rocfft set work buffer failed (%d)", (int)st);
    }
    return PRC_OK;
}

extern "C" int prc_caf_plan_create(prc_caf_plan** plan, const prc_caf_desc* d) {
    PRC_REQUIRE(plan && d, PRC_EINVAL, "prc_caf_plan_create: null argument");
    PRC_REQUIRE(d->n > 0 && d->range_bins >= 0 && d->freq_bins > 0 && d->max_frames > 0,
                PRC_EINVAL, "prc_caf_plan_create: non-positive size");
    PRC_REQUIRE(d->freq_bins <= d->n, PRC_EINVAL,
                "prc_caf_plan_create: freq_bins (%d) exceeds the CPI length (%lld): the reference "
                "divides by q = int(n/freq_bins) = 0", d->freq_bins, (long long)d->n);
    PRC_REQUIRE(d->range_bins < d->n, PRC_EINVAL, "prc_caf_plan_create: range_bins >= n");
    PRC_REQUIRE(d->ntaps == 0 || d->taps_host, PRC_EINVAL, "prc_caf_plan_create: ntaps without taps");
    prc_caf_plan* p = new prc_caf_plan();
    p->desc = *d;
    p->desc.taps_host = nullptr;
    p->q = d->n / d->freq_bins;
    p->ntaps = d->ntaps ? d->ntaps : (int)(p->q + 1);
    p->half = (p->ntaps - 1) / 2;
    const bool boxcar = d->ntaps == 0;
    // resolve methods
    p->method = d->method;
    if (p->method == PRC_CAF_AUTO) {
        p->method = caf_fft_supported(d->n, d->range_bins, d->freq_bins, boxcar) ? PRC_CAF_FFT
                                                                                 : PRC_CAF_DIRECT;
        // wide range spans: a 4096-point team transform costs about 4.4 one-wavefront 1024-point transforms
        // (four wavefronts, one more radix-16 pass); take it when the segment needs fewer of them
        if (p->method == PRC_CAF_FFT && caf_team_supported(d->n, d->range_bins, d->freq_bins, boxcar) &&
            4.4 * caf_team_blocking(p->ntaps, d->range_bins, nullptr, nullptr) <
                caf_fft_blocking(p->ntaps, d->range_bins, nullptr, nullptr))
            p->method = PRC_CAF_FFT4096;
    }
    if (p->method == PRC_CAF_FFT4096 && !caf_team_supported(d->n, d->range_bins, d->freq_bins, boxcar)) {
        prc_set_error("prc_caf_plan_create: 4096-point FFT segment method unsupported for n=%lld R=%d F=%d%s",
                      (long long)d->n, d->range_bins, d->freq_bins, boxcar ? "" : " (long FIR)");
        delete p;
        return PRC_EUNSUPPORTED;
    }
    if (p->method == PRC_CAF_FFT && !caf_fft_supported(d->n, d->range_bins, d->freq_bins, boxcar)) {
        prc_set_error("prc_caf_plan_create: FFT segment method unsupported for n=%lld R=%d F=%d%s",
                      (long long)d->n, d->range_bins, d->freq_bins, boxcar ? "" : " (long FIR)");
        delete p;
        return PRC_EUNSUPPORTED;
    }
    p->doppler = d->doppler == PRC_DOPPLER_AUTO ? PRC_DOPPLER_ROCFFT : d->doppler;
    if (p->doppler != PRC_DOPPLER_ROCFFT) {
        prc_set_error("prc_caf_plan_create: unknown Doppler method %d", d->doppler);
        delete p;
        return PRC_EINVAL;
    }
    int rc = PRC_OK;
    auto fail = [&](int code) { prc_caf_plan_destroy(p); return code; };
    p->y_bytes = sizeof(float2) * (size_t)d->max_frames * d->freq_bins * (d->range_bins + 1);
    if (hipMalloc(&p->d_y, p->y_bytes) != hipSuccess) {
        prc_set_error("prc_caf_plan_create: hipMalloc(%zu) failed: %s", p->y_bytes,
                      hipGetErrorString(hipGetLastError()));
        return fail(PRC_EHIP);
    }
    if ((p->method == PRC_CAF_FFT || p->method == PRC_CAF_FFT4096) && p->doppler == PRC_DOPPLER_ROCFFT) {
        if (hipMalloc(&p->d_y2, p->y_bytes) != hipSuccess) {
            prc_set_error("prc_caf_plan_create: hipMalloc(%zu) failed", p->y_bytes);
            return fail(PRC_EHIP);
        }
    }
    if (!boxcar) {
        if (hipMalloc(&p->d_taps, sizeof(float) * d->ntaps) != hipSuccess ||
            hipMemcpy(p->d_taps, d->taps_host, sizeof(float) * d->ntaps, hipMemcpyHostToDevice) != hipSuccess) {
            prc_set_error("prc_caf_plan_create: taps upload failed");
            return fail(PRC_EHIP);
        }
    }
    if (p->doppler == PRC_DOPPLER_ROCFFT) {
        rc = build_rocfft(p, d->max_frames);
        if (rc != PRC_OK) return fail(rc);
    }
    *plan = p;
    return PRC_OK;
}

extern "C" int prc_caf_plan_destroy(prc_caf_plan* p) {
    if (!p) return PRC_OK;
    if (p->info) rocfft_execution_info_destroy(p->info);
    if (p->fft) rocfft_plan_destroy(p->fft);
    if (p->d_work) (void)hipFree(p->d_work);
    if (p->d_taps) (void)hipFree(p->d_taps);
    if (p->d_y) (void)hipFree(p->d_y);
    if (p->d_y2) (void)hipFree(p->d_y2);
    delete p;
    return PRC_OK;
}

extern "C" int prc_caf_plan_info(const prc_caf_plan* p, int32_t* method, int32_t* doppler,
                                 int64_t* workspace_bytes) {
    PRC_REQUIRE(p, PRC_EINVAL, "prc_caf_plan_info: null plan");
    if (method) *method = p->method;
    if (doppler) *doppler = p->doppler;
    if (workspace_bytes) *workspace_bytes = (int64_t)(p->y_bytes * (p->d_y2 ? 2 : 1) + p->work_bytes);
    return PRC_OK;
}

static int check_exec(prc_caf_plan* p, int nframes, const char* who) {
    PRC_REQUIRE(p, PRC_EINVAL, "%s: null plan", who);
    PRC_REQUIRE(nframes > 0 && nframes <= p->desc.max_frames, PRC_EINVAL,
                "%s: nframes=%d outside [1, max_frames=%d]", who, nframes, p->desc.max_frames);
    return PRC_OK;
}

static int run_segments(prc_caf_plan* p, const void* ref, const void* srv, int64_t frame_stride,
                        int64_t n_valid, const float* window, int nframes, hipStream_t stream) {
    PRC_REQUIRE(ref && srv, PRC_EINVAL, "prc_caf_execute: null input");
    PRC_REQUIRE(n_valid >= 0 && n_valid <= p->desc.n, PRC_ESHAPE,
                "prc_caf_execute: n_valid=%lld exceeds inputLen=%lld", (long long)n_valid,
                (long long)p->desc.n);
    CafSegArgs a;
    a.ref = (const float2*)ref;
    a.srv = (const float2*)srv;
    a.window = window;
    a.taps = p->d_taps;
    a.y = p->d_y;
    a.frame_stride = frame_stride;
    a.n = p->desc.n;
    a.n_valid = n_valid;
    a.q = p->q;
    a.ntaps = p->ntaps;
    a.half = p->half;
    a.range_bins = p->desc.range_bins;
    a.freq_bins = p->desc.freq_bins;
    a.y_layout = PRC_Y_KJ;
    if (p->method == PRC_CAF_FFT || p->method == PRC_CAF_FFT4096) {
        // the FFT kernels write whole rows y[j][0..R] (coalesced); rocFFT wants j contiguous
        a.y = p->d_y2;
        a.y_layout = PRC_Y_JK;
        int rc = p->method == PRC_CAF_FFT ? caf_launch_fft(a, nframes, stream) : caf_launch_fft_team(a, nframes, stream);
        if (rc) return rc;
        return caf_launch_transpose_jk_kj(p->d_y2, p->d_y, a.freq_bins, a.range_bins + 1, nframes, stream);
    }
    return caf_launch_direct(a, nframes, stream);
}

static int run_doppler(prc_caf_plan* p, void* out, int nframes, hipStream_t stream) {
    PRC_REQUIRE(out, PRC_EINVAL, "prc_caf_execute: null output");
    const int F = p->desc.freq_bins, cols = p->desc.range_bins + 1;
    // rocFFT: the plan is batched for max_frames; transforming the unused tail is harmless
    // (it lives in the plan's own buffer) and keeps one plan per shape.
    rocfft_status st = rocfft_execution_info_set_stream(p->info, stream);
    PRC_REQUIRE(st == rocfft_status_success, PRC_EROCFFT, "rocfft set stream failed (%d)", (int)st);
    void* bufs[1] = {p->d_y};
    st = rocfft_execute(p->fft, bufs, nullptr, p->info);
    PRC_REQUIRE(st == rocfft_status_success, PRC_EROCFFT, "rocfft_execute failed (%d)", (int)st);
    dim3 grid((F + 31) / 32, (cols + 31) / 32, nframes);
    hipLaunchKernelGGL(shift_transpose_kernel, grid, dim3(256), 0, stream, p->d_y, (float2*)out, F, cols);
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}

extern "C" int prc_caf_execute_segments(prc_caf_plan* p, const void* ref, const void* srv,
                                        int64_t frame_stride, int64_t n_valid, const float* window,
                                        int32_t nframes, void* stream) {
    int rc = check_exec(p, nframes, "prc_caf_execute_segments");
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(p->mtx);
    return run_segments(p, ref, srv, frame_stride, n_valid, window, nframes, (hipStream_t)stream);
}

extern "C" int prc_caf_execute_doppler(prc_caf_plan* p, void* out, int32_t nframes, void* stream) {
    int rc = check_exec(p, nframes, "prc_caf_execute_doppler");
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(p->mtx);
    return run_doppler(p, out, nframes, (hipStream_t)stream);
}

extern "C" int prc_caf_execute(prc_caf_plan* p, const void* ref, const void* srv, int64_t frame_stride,
                               int64_t n_valid, const float* window, void* out, int32_t nframes,
                               void* stream) {
    int rc = check_exec(p, nframes, "prc_caf_execute");
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(p->mtx);
    rc = run_segments(p, ref, srv, frame_stride, n_valid, window, nframes, (hipStream_t)stream);
    if (rc) return rc;
    return run_doppler(p, out, nframes, (hipStream_t)stream);
}
