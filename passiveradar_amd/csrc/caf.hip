// Cross-ambiguity plan: segment sums (caf_direct.hip / caf_fft.hip) + Doppler FFT + fftshift.
//
// Replaces fast_xambg, range_doppler_processing.py:12-90.  The Doppler stage (:89,
// scipy.fftpack.fft(axis=0) then np.fft.fftshift) is ONE column-FFT kernel over the row-major slow-time
// buffer for power-of-two freq_bins from 256 to 4096 (doppler_col.h), and for any other size a rocFFT batched
// 1-D plan over the contiguous slow-time axis between a transpose and a shift+transpose kernel.
//
// Segment sums and Doppler transforms alternate over groups of surfaces sized to the Infinity Cache, so the
// slow-time buffer is read back from the cache that took its writes rather than from HBM.
#include "caf_internal.h"
#include "doppler_col.h"
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
    float* d_taps_rev = nullptr;     // ... and in reverse order (the FFT segment kernel's weight stream)
    float2* d_y = nullptr;           // [k][j]-ordered slow-time buffer of the rocFFT path, max_frames * F * (R+1)
    float2* d_y2 = nullptr;          // [j][k]-ordered slow-time buffer written row-wise by the segment kernels
    float2* d_dop_tw = nullptr;      // W_F^m table of the column-FFT Doppler kernel
    int y_kt = 0;                    // column-FFT Doppler path: the slow-time buffer is tiled by this many columns (0: plain rows)
    int64_t y_surf = 0;              // elements per surface of the slow-time buffer
    int group = 1;                   // surfaces per segment/Doppler round of prc_caf_execute
    int multi = PRC_CAF_MULTI_TURNS; // resolved prc_caf_multi_mode of prc_caf_execute_multi
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

// surfaces per segment/Doppler round.  Cache-sized groups (so that the slow-time buffer would be read back from the
// Infinity Cache) lost to the tail effect of the smaller launches at every size measured: the
// default is the whole batch; PRC_OPT_CAF_GROUP_MB, read here at plan creation, sizes the groups for A/B runs.
static int pick_group(const prc_caf_desc* d) {
    const int64_t mb = prc_opt(PRC_OPT_CAF_GROUP_MB);
    if (mb <= 0) return d->max_frames;
    const double surf = 8.0 * (double)d->freq_bins * (double)(d->range_bins + 1);
    int g = (int)((double)mb * 1048576.0 / surf);
    if (g < 1) g = 1;
    if (g > d->max_frames) g = d->max_frames;
    return g;
}

extern "C" int prc_caf_plan_create(prc_caf_plan** plan, const prc_caf_desc* host_desc) {
    PRC_REQUIRE(plan && host_desc, PRC_EINVAL, "prc_caf_plan_create: null argument");
    static_assert(sizeof(prc_caf_desc) == PRC_CAF_DESC_SIZE_600, "prc_caf_desc grew: keep PRC_CAF_DESC_SIZE_600, default the new fields to 0");
    prc_caf_desc mine;
    if (int rc = prc_take_desc(&mine, host_desc, PRC_CAF_DESC_SIZE_600, "prc_caf_plan_create", "prc_caf_desc")) return rc;
    const prc_caf_desc* d = &mine;
    PRC_REQUIRE(d->n > 0 && d->range_bins >= 0 && d->freq_bins > 0 && d->max_frames > 0,
                PRC_EINVAL, "prc_caf_plan_create: non-positive size");
    PRC_REQUIRE(d->freq_bins <= d->n, PRC_EINVAL,
                "prc_caf_plan_create: freq_bins (%d) exceeds the CPI length (%lld): the reference "
                "divides by q = int(n/freq_bins) = 0", d->freq_bins, (long long)d->n);
    PRC_REQUIRE(d->range_bins < d->n, PRC_EINVAL, "prc_caf_plan_create: range_bins >= n");
    PRC_REQUIRE(d->ntaps == 0 || d->taps_host, PRC_EINVAL, "prc_caf_plan_create: ntaps without taps");
    // every argument check comes before the plan exists (a failed check after `new` leaked it)
    PRC_REQUIRE(d->multi >= PRC_CAF_MULTI_AUTO && d->multi <= PRC_CAF_MULTI_PAIRS, PRC_EINVAL,
                "prc_caf_plan_create: unknown multi mode %d", d->multi);
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
    p->doppler = d->doppler;
    if (p->doppler == PRC_DOPPLER_AUTO)
        p->doppler = dop_supported(d->freq_bins, d->range_bins + 1) ? PRC_DOPPLER_COLUMN : PRC_DOPPLER_ROCFFT;
    if (p->doppler == PRC_DOPPLER_COLUMN && !dop_supported(d->freq_bins, d->range_bins + 1)) {
        prc_set_error("prc_caf_plan_create: the column-FFT Doppler kernel takes freq_bins 256, 512, 1024, 2048, 4096 "
                      "and surfaces below 2^32 / 3 bytes (got %d x %d)", d->freq_bins, d->range_bins + 1);
        delete p;
        return PRC_EUNSUPPORTED;
    }
    p->multi = d->multi != PRC_CAF_MULTI_AUTO ? d->multi : (int)prc_opt(PRC_OPT_CAF_MULTI_MODE);
    if (p->multi == PRC_CAF_MULTI_AUTO) {
        // measured on MI355X, four illuminators (tools/caf_bench.py --nref 4): config-3 span (1025
        // lags, pieces of 3072 + a remainder) shared 110 us per frame against 120 for turns; config-5 span (2049 lags, two
        // pieces of 2048) shared 324-329 against 313 -- sharing pays where a segment's first piece is longer than 2048
        // samples (fewer, longer transforms per illuminator next to the two shared ones)
        p->multi = PRC_CAF_MULTI_TURNS;
        if (p->method == PRC_CAF_FFT4096 && p->doppler == PRC_DOPPLER_COLUMN &&
            caf_team_multi_supported(d->n, d->range_bins, d->freq_bins, p->ntaps, 2)) {
            int nlb = 1, lb = d->range_bins + 1;
            caf_team_multi_blocking(p->ntaps, d->range_bins, 2, &nlb, &lb);
            if (4097 - lb > 2048) p->multi = PRC_CAF_MULTI_SHARED;
        }
    }
    if (p->doppler != PRC_DOPPLER_ROCFFT && p->doppler != PRC_DOPPLER_COLUMN) {
        prc_set_error("prc_caf_plan_create: unknown Doppler method %d", d->doppler);
        delete p;
        return PRC_EINVAL;
    }
    int rc = PRC_OK;
    auto fail = [&](int code) { prc_caf_plan_destroy(p); return code; };
    // the column-FFT Doppler kernel reads tiles of KT columns x F rows: the segment kernels write them contiguously
    p->y_kt = p->doppler == PRC_DOPPLER_COLUMN ? dop_tile_cols(d->freq_bins) : 0;
    {
        const int64_t cols = d->range_bins + 1;
        p->y_surf = p->y_kt ? (int64_t)d->freq_bins * ((cols + p->y_kt - 1) / p->y_kt) * p->y_kt : (int64_t)d->freq_bins * cols;
    }
    p->y_bytes = sizeof(float2) * (size_t)d->max_frames * (size_t)p->y_surf;
    p->group = pick_group(d);
    const bool rowwise = p->method == PRC_CAF_FFT || p->method == PRC_CAF_FFT4096 || p->doppler == PRC_DOPPLER_COLUMN;
    if (p->doppler == PRC_DOPPLER_ROCFFT && hipMalloc(&p->d_y, p->y_bytes) != hipSuccess) {
        prc_set_error("prc_caf_plan_create: hipMalloc(%zu) failed: %s", p->y_bytes,
                      hipGetErrorString(hipGetLastError()));
        return fail(PRC_EHIP);
    }
    if (rowwise && hipMalloc(&p->d_y2, p->y_bytes) != hipSuccess) {
        prc_set_error("prc_caf_plan_create: hipMalloc(%zu) failed", p->y_bytes);
        return fail(PRC_EHIP);
    }
    if (!boxcar) {
        std::vector<float> rev((size_t)d->ntaps);
        for (int i = 0; i < d->ntaps; ++i) rev[i] = d->taps_host[d->ntaps - 1 - i];
        if (hipMalloc(&p->d_taps, sizeof(float) * d->ntaps) != hipSuccess ||
            hipMemcpy(p->d_taps, d->taps_host, sizeof(float) * d->ntaps, hipMemcpyHostToDevice) != hipSuccess ||
            hipMalloc(&p->d_taps_rev, sizeof(float) * d->ntaps) != hipSuccess ||
            hipMemcpy(p->d_taps_rev, rev.data(), sizeof(float) * d->ntaps, hipMemcpyHostToDevice) != hipSuccess) {
            prc_set_error("prc_caf_plan_create: taps upload failed");
            return fail(PRC_EHIP);
        }
    }
    if (p->doppler == PRC_DOPPLER_ROCFFT) {
        rc = build_rocfft(p, d->max_frames);
        if (rc != PRC_OK) return fail(rc);
    } else {
        std::vector<float2> tw((size_t)d->freq_bins);
        dop_make_table(tw.data(), d->freq_bins);
        if (hipMalloc(&p->d_dop_tw, sizeof(float2) * tw.size()) != hipSuccess ||
            hipMemcpy(p->d_dop_tw, tw.data(), sizeof(float2) * tw.size(), hipMemcpyHostToDevice) != hipSuccess) {
            prc_set_error("prc_caf_plan_create: Doppler twiddle upload failed");
            return fail(PRC_EHIP);
        }
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
    if (p->d_taps_rev) (void)hipFree(p->d_taps_rev);
    if (p->d_y) (void)hipFree(p->d_y);
    if (p->d_y2) (void)hipFree(p->d_y2);
    if (p->d_dop_tw) (void)hipFree(p->d_dop_tw);
    delete p;
    return PRC_OK;
}

extern "C" int prc_caf_plan_multi_mode(const prc_caf_plan* p, int32_t* multi) {
    PRC_REQUIRE(p && multi, PRC_EINVAL, "prc_caf_plan_multi_mode: null argument");
    *multi = p->multi;
    return PRC_OK;
}

extern "C" int prc_caf_plan_info(const prc_caf_plan* p, int32_t* method, int32_t* doppler,
                                 int64_t* workspace_bytes) {
    PRC_REQUIRE(p, PRC_EINVAL, "prc_caf_plan_info: null plan");
    if (method) *method = p->method;
    if (doppler) *doppler = p->doppler;
    if (workspace_bytes)
        *workspace_bytes = (int64_t)(p->y_bytes * ((p->d_y ? 1 : 0) + (p->d_y2 ? 1 : 0)) + p->work_bytes);
    return PRC_OK;
}

static int check_exec(prc_caf_plan* p, int nframes, const char* who) {
    PRC_REQUIRE(p, PRC_EINVAL, "%s: null plan", who);
    PRC_REQUIRE(nframes > 0 && nframes <= p->desc.max_frames, PRC_EINVAL,
                "%s: nframes=%d outside [1, max_frames=%d]", who, nframes, p->desc.max_frames);
    return PRC_OK;
}

static int64_t surf_elems(const prc_caf_plan* p) {
    return (int64_t)p->desc.freq_bins * (p->desc.range_bins + 1);
}

// segment sums of frames [f0, f0 + nf) of one reference channel into the plan's slow-time buffer, surfaces s0...
static int run_segments(prc_caf_plan* p, const void* ref, const void* srv, int64_t frame_stride,
                        int64_t n_valid, const float* window, int f0, int nf, int s0, hipStream_t stream) {
    PRC_REQUIRE(ref && srv, PRC_EINVAL, "prc_caf_execute: null input");
    PRC_REQUIRE(n_valid >= 0 && n_valid <= p->desc.n, PRC_ESHAPE,
                "prc_caf_execute: n_valid=%lld exceeds inputLen=%lld", (long long)n_valid,
                (long long)p->desc.n);
    CafSegArgs a;
    a.ref = (const float2*)ref + (int64_t)f0 * frame_stride;
    a.srv = (const float2*)srv + (int64_t)f0 * frame_stride;
    a.window = window;
    a.taps = p->d_taps;
    a.taps_rev = p->d_taps_rev;
    a.frame_stride = frame_stride;
    a.n = p->desc.n;
    a.n_valid = n_valid;
    a.q = p->q;
    a.ntaps = p->ntaps;
    a.half = p->half;
    a.range_bins = p->desc.range_bins;
    a.freq_bins = p->desc.freq_bins;
    a.y_kt = p->y_kt;
    a.y_surface = p->y_surf;
    const int64_t off = (int64_t)s0 * p->y_surf;
    if (p->method == PRC_CAF_FFT || p->method == PRC_CAF_FFT4096) {
        // the FFT kernels write whole rows y[j][0..R] (coalesced)
        a.y = p->d_y2 + off;
        a.y_layout = PRC_Y_JK;
        int rc = p->method == PRC_CAF_FFT ? caf_launch_fft(a, nf, stream) : caf_launch_fft_team(a, nf, stream);
        if (rc) return rc;
        if (p->doppler == PRC_DOPPLER_ROCFFT)       // rocFFT wants j contiguous
            return caf_launch_transpose_jk_kj(p->d_y2 + off, p->d_y + off, a.freq_bins, a.range_bins + 1, nf, stream);
        return PRC_OK;
    }
    a.y = (p->doppler == PRC_DOPPLER_COLUMN ? p->d_y2 : p->d_y) + off;
    a.y_layout = p->doppler == PRC_DOPPLER_COLUMN ? PRC_Y_JK : PRC_Y_KJ;
    return caf_launch_direct(a, nf, stream);
}

// Doppler transform + fftshift of surfaces [s0, s0 + ns) of the slow-time buffer into out (surface s0 first)
static int run_doppler(prc_caf_plan* p, void* out, int s0, int ns, hipStream_t stream) {
    PRC_REQUIRE(out, PRC_EINVAL, "prc_caf_execute: null output");
    const int F = p->desc.freq_bins, cols = p->desc.range_bins + 1;
    if (p->doppler == PRC_DOPPLER_COLUMN)
        return dop_launch(p->d_y2 + (int64_t)s0 * p->y_surf, p->y_surf, (float2*)out, p->d_dop_tw, F, cols, ns, stream);
    // rocFFT: the plan is batched for max_frames and always transforms the whole buffer (one plan per shape); the
    // surfaces outside [s0, s0 + ns) are transformed in place too, so this path takes whole batches only
    PRC_REQUIRE(s0 == 0, PRC_EINVAL, "prc_caf_execute: the rocFFT Doppler path transforms whole batches");
    rocfft_status st = rocfft_execution_info_set_stream(p->info, stream);
    PRC_REQUIRE(st == rocfft_status_success, PRC_EROCFFT, "rocfft set stream failed (%d)", (int)st);
    void* bufs[1] = {p->d_y};
    st = rocfft_execute(p->fft, bufs, nullptr, p->info);
    PRC_REQUIRE(st == rocfft_status_success, PRC_EROCFFT, "rocfft_execute failed (%d)", (int)st);
    dim3 grid((F + 31) / 32, (cols + 31) / 32, ns);
    hipLaunchKernelGGL(shift_transpose_kernel, grid, dim3(256), 0, stream, p->d_y, (float2*)out, F, cols);
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}

extern "C" int prc_caf_execute_segments(prc_caf_plan* p, const void* ref, const void* srv,
                                        int64_t frame_stride, int64_t n_valid, const float* window,
                                        int32_t nframes, void* stream) {
    PRC_RANGE("prc_caf_execute_segments");
    int rc = check_exec(p, nframes, "prc_caf_execute_segments");
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(p->mtx);
    return run_segments(p, ref, srv, frame_stride, n_valid, window, 0, nframes, 0, (hipStream_t)stream);
}

extern "C" int prc_caf_execute_doppler(prc_caf_plan* p, void* out, int32_t nframes, void* stream) {
    PRC_RANGE("prc_caf_execute_doppler");
    int rc = check_exec(p, nframes, "prc_caf_execute_doppler");
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(p->mtx);
    return run_doppler(p, out, 0, nframes, (hipStream_t)stream);
}

extern "C" int prc_caf_execute(prc_caf_plan* p, const void* ref, const void* srv, int64_t frame_stride,
                               int64_t n_valid, const float* window, void* out, int32_t nframes,
                               void* stream) {
    PRC_RANGE("prc_caf_execute");
    int rc = check_exec(p, nframes, "prc_caf_execute");
    if (rc) return rc;
    PRC_REQUIRE(out, PRC_EINVAL, "prc_caf_execute: null output");
    std::lock_guard<std::mutex> lk(p->mtx);
    const int g = p->doppler == PRC_DOPPLER_COLUMN ? p->group : nframes;
    for (int f0 = 0; f0 < nframes; f0 += g) {
        const int nf = nframes - f0 < g ? nframes - f0 : g;
        rc = run_segments(p, ref, srv, frame_stride, n_valid, window, f0, nf, f0, (hipStream_t)stream);
        if (rc) return rc;
        rc = run_doppler(p, (float2*)out + (int64_t)f0 * surf_elems(p), f0, nf, (hipStream_t)stream);
        if (rc) return rc;
    }
    return PRC_OK;
}

// fast_xambg for nref reference channels against ONE surveillance channel (range_doppler_processing.py:12-90 once
// per pair): surface (i, b) of the workspace is illuminator i, frame b.
extern "C" int prc_caf_execute_multi(prc_caf_plan* p, const void* const* refs_host, int32_t nref, const void* srv,
                                     int64_t frame_stride, int64_t n_valid, const float* window,
                                     void* const* outs_host, int32_t nframes, void* stream) {
    PRC_RANGE("prc_caf_execute_multi");
    PRC_REQUIRE(p && refs_host && outs_host && srv, PRC_EINVAL, "prc_caf_execute_multi: null argument");
    PRC_REQUIRE(nref >= 1 && nref <= PRC_CAF_MAX_REFS, PRC_EINVAL, "prc_caf_execute_multi: nref=%d outside [1, %d]",
                nref, PRC_CAF_MAX_REFS);
    PRC_REQUIRE(nframes > 0 && (int64_t)nframes * nref <= p->desc.max_frames, PRC_EINVAL,
                "prc_caf_execute_multi: nframes * nref = %lld surfaces exceed the plan's max_frames = %d",
                (long long)nframes * nref, p->desc.max_frames);
    PRC_REQUIRE(n_valid >= 0 && n_valid <= p->desc.n, PRC_ESHAPE,
                "prc_caf_execute_multi: n_valid=%lld exceeds inputLen=%lld", (long long)n_valid, (long long)p->desc.n);
    for (int i = 0; i < nref; ++i)
        PRC_REQUIRE(refs_host[i] && outs_host[i], PRC_EINVAL, "prc_caf_execute_multi: null channel %d", i);
    std::lock_guard<std::mutex> lk(p->mtx);
    hipStream_t st = (hipStream_t)stream;
    const int64_t se = surf_elems(p);
    // SHARED: all illuminators in one launch; PAIRS: two per launch (a last odd one alone goes through the same kernel)
    const int per = p->multi == PRC_CAF_MULTI_SHARED ? nref : (p->multi == PRC_CAF_MULTI_PAIRS ? 2 : 1);
    const bool shared = per > 1 && p->method == PRC_CAF_FFT4096 && p->doppler == PRC_DOPPLER_COLUMN && nref > 1 &&
                        caf_team_multi_supported(p->desc.n, p->desc.range_bins, p->desc.freq_bins, p->ntaps, per);
    if (!shared && nref > 1 && p->method == PRC_CAF_FFT4096 && p->doppler == PRC_DOPPLER_COLUMN) {
        // "turns" in ONE launch per stage: the single-reference kernel with the illuminator as a third grid dimension, then the
        // Doppler kernel over every channel's surfaces -- nothing shared, but no tail of nref small launches (config 5, 16
        // frames: 10.7 rounds of workgroups per segment launch and 8.03 per Doppler launch, each rounded up, become 42.7 and 32.1)
        int g = p->group / nref;
        if (g < 1) g = 1;
        for (int f0 = 0; f0 < nframes; f0 += g) {
            const int nf = nframes - f0 < g ? nframes - f0 : g;
            CafSegArgs a;
            a.ref = nullptr;
            a.srv = (const float2*)srv + (int64_t)f0 * frame_stride;
            a.window = window;
            a.taps = nullptr;
            a.taps_rev = nullptr;
            a.y = p->d_y2;
            a.frame_stride = frame_stride;
            a.n = p->desc.n;
            a.n_valid = n_valid;
            a.q = p->q;
            a.ntaps = p->ntaps;
            a.half = p->half;
            a.range_bins = p->desc.range_bins;
            a.freq_bins = p->desc.freq_bins;
            a.y_layout = PRC_Y_JK;
            a.y_kt = p->y_kt;
            a.y_surface = p->y_surf;
            const float2* refs[PRC_CAF_MAX_REFS];
            float2* outs[PRC_CAF_MAX_REFS];
            for (int i = 0; i < nref; ++i) {
                refs[i] = (const float2*)refs_host[i] + (int64_t)f0 * frame_stride;
                outs[i] = (float2*)outs_host[i] + (int64_t)f0 * se;
            }
            int rc = caf_launch_fft_team_refs(a, refs, nref, (int64_t)nf * p->y_surf, nf, st);
            if (rc) return rc;
            rc = dop_launch_multi(p->d_y2, p->y_surf, (int64_t)nf * p->y_surf, outs, nref, p->d_dop_tw, p->desc.freq_bins,
                                  p->desc.range_bins + 1, nf, st);
            if (rc) return rc;
        }
        return PRC_OK;
    }
    if (!shared) {
        // one pass per illuminator (any method): same results, nothing shared
        for (int i = 0; i < nref; ++i) {
            const int g = p->doppler == PRC_DOPPLER_COLUMN ? p->group : nframes;
            for (int f0 = 0; f0 < nframes; f0 += g) {
                const int nf = nframes - f0 < g ? nframes - f0 : g;
                int rc = run_segments(p, refs_host[i], srv, frame_stride, n_valid, window, f0, nf, f0, st);
                if (rc) return rc;
                rc = run_doppler(p, (float2*)outs_host[i] + (int64_t)f0 * se, f0, nf, st);
                if (rc) return rc;
            }
        }
        return PRC_OK;
    }
    // frames in groups of g: surface (i, b) of a group at [i * nf + b] of the workspace; the surveillance pieces of a
    // segment are transformed once per launch for the `per` illuminators it carries
    int g = p->group / nref;
    if (g < 1) g = 1;
    for (int f0 = 0; f0 < nframes; f0 += g) {
        const int nf = nframes - f0 < g ? nframes - f0 : g;
        CafSegArgs a;
        a.ref = nullptr;
        a.srv = (const float2*)srv + (int64_t)f0 * frame_stride;
        a.window = window;
        a.taps = nullptr;
        a.taps_rev = nullptr;
        a.frame_stride = frame_stride;
        a.n = p->desc.n;
        a.n_valid = n_valid;
        a.q = p->q;
        a.ntaps = p->ntaps;
        a.half = p->half;
        a.range_bins = p->desc.range_bins;
        a.freq_bins = p->desc.freq_bins;
        a.y_layout = PRC_Y_JK;
        a.y_kt = p->y_kt;
        a.y_surface = p->y_surf;
        for (int i0 = 0; i0 < nref; i0 += per) {
            const int k = nref - i0 < per ? nref - i0 : per;
            const float2* refs[PRC_CAF_MAX_REFS];
            for (int i = 0; i < k; ++i) refs[i] = (const float2*)refs_host[i0 + i] + (int64_t)f0 * frame_stride;
            a.y = p->d_y2 + (int64_t)i0 * nf * p->y_surf;
            int rc = caf_launch_fft_team_multi(a, refs, k, (int64_t)nf * p->y_surf, nf, st);
            if (rc) return rc;
        }
        float2* outs[PRC_CAF_MAX_REFS];
        for (int i = 0; i < nref; ++i) outs[i] = (float2*)outs_host[i] + (int64_t)f0 * se;
        int rc = dop_launch_multi(p->d_y2, p->y_surf, (int64_t)nf * p->y_surf, outs, nref, p->d_dop_tw, p->desc.freq_bins,
                                  p->desc.range_bins + 1, nf, st);
        if (rc) return rc;
    }
    return PRC_OK;
}
