// Channel offset estimation on device (SURVEY 8f "next" #2): signal_utils.py:73-78,
//   B1 = decimate(s1, nd); B2 = pad(decimate(s2, nd), nl); xc = |correlate(B1, B2, 'valid')|; (argmax - nl) * nd
// called once at start-up with nd = 1, nl = 5e6 on 10..20 CPIs of raw data (main.py:47-54, :81-83).
//
// scipy.signal.decimate(x, q) defaults to an order-8 Chebyshev-I low-pass (0.05 dB, 0.8/q) run forwards and
// backwards (sosfiltfilt): odd extension of 27 samples each side, each pass started in the steady state of its
// first input sample.  A recursion is serial; its impulse response is not: |pole|max^settle < 1e-9, so one pass
// equals a circular convolution of [first sample x settle | extended signal | zeros] with h, i.e. a spectral
// multiply by H(e^jw) evaluated from the poles/zeros in double.  Second pass: hold the last forward output for
// `settle` samples and multiply by conj(H).  Same trimming and [::q] as the reference; the long correlation is
// three rocFFT transforms.
#include "common.h"
#include <rocfft/rocfft.h>

namespace {

std::once_flag g_once;
void setup_once() { std::call_once(g_once, [] { rocfft_setup(); }); }

struct Fft1d {
    rocfft_plan fwd = nullptr, inv = nullptr;
    rocfft_execution_info info = nullptr;
    void* work = nullptr;
    ~Fft1d() {
        if (info) rocfft_execution_info_destroy(info);
        if (fwd) rocfft_plan_destroy(fwd);
        if (inv) rocfft_plan_destroy(inv);
        if (work) (void)hipFree(work);
    }
    int create(size_t len, hipStream_t stream) {
        setup_once();
        rocfft_status st = rocfft_plan_create(&fwd, rocfft_placement_inplace, rocfft_transform_type_complex_forward,
                                              rocfft_precision_single, 1, &len, 1, nullptr);
        PRC_REQUIRE(st == rocfft_status_success, PRC_EROCFFT, "rocfft_plan_create(%zu) failed (%d)", len, (int)st);
        st = rocfft_plan_create(&inv, rocfft_placement_inplace, rocfft_transform_type_complex_inverse,
                                rocfft_precision_single, 1, &len, 1, nullptr);
        PRC_REQUIRE(st == rocfft_status_success, PRC_EROCFFT, "rocfft_plan_create(%zu) failed (%d)", len, (int)st);
        size_t wf = 0, wi = 0;
        rocfft_plan_get_work_buffer_size(fwd, &wf);
        rocfft_plan_get_work_buffer_size(inv, &wi);
        const size_t wb = wf > wi ? wf : wi;
        st = rocfft_execution_info_create(&info);
        PRC_REQUIRE(st == rocfft_status_success, PRC_EROCFFT, "rocfft info create failed (%d)", (int)st);
        if (wb) {
            PRC_HIP(hipMalloc(&work, wb));
            st = rocfft_execution_info_set_work_buffer(info, work, wb);
            PRC_REQUIRE(st == rocfft_status_success, PRC_EROCFFT, "rocfft set work buffer failed (%d)", (int)st);
        }
        st = rocfft_execution_info_set_stream(info, stream);
        PRC_REQUIRE(st == rocfft_status_success, PRC_EROCFFT, "rocfft set stream failed (%d)", (int)st);
        return PRC_OK;
    }
    int run(bool forward, void* buf) {
        void* bufs[1] = {buf};
        rocfft_status st = rocfft_execute(forward ? fwd : inv, bufs, nullptr, info);
        PRC_REQUIRE(st == rocfft_status_success, PRC_EROCFFT, "rocfft_execute failed (%d)", (int)st);
        return PRC_OK;
    }
};

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};

struct IirDev {
    int nz, np;
    double2 z[16], p[16];
    double gain;
};

size_t next_pow2(size_t v) {
    size_t m = 1;
    while (m < v) m <<= 1;
    return m;
}

}  // namespace

// buf[0, L) = xe[0]; buf[L + i] = xe[i] (odd extension of x by P each side); zero to M
__global__ void iir_extend_kernel(const float2* __restrict__ x, int64_t n, int P, int64_t L, int64_t M,
                                  float2* __restrict__ buf) {
    const int64_t nx = n + 2 * (int64_t)P;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < M; t += (int64_t)gridDim.x * blockDim.x) {
        float2 v = make_float2(0.f, 0.f);
        if (t < L + nx) {
            int64_t i = t < L ? 0 : t - L;
            if (i < P) {
                const float2 a = x[0], b = x[P - i];
                v = make_float2(2.f * a.x - b.x, 2.f * a.y - b.y);
            } else if (i < P + n) {
                v = x[i - P];
            } else {
                const float2 a = x[n - 1], b = x[n - 2 - (i - P - n)];
                v = make_float2(2.f * a.x - b.x, 2.f * a.y - b.y);
            }
        }
        buf[t] = v;
    }
}

// spectrum *= (CONJ ? conj(H) : H)(e^{j 2 pi k / M}) * scale
template <bool CONJ>
__global__ void iir_response_kernel(float2* __restrict__ buf, int64_t M, IirDev f, double scale) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < M; k += (int64_t)gridDim.x * blockDim.x) {
        double s, c;
        sincospi(2.0 * (double)k / (double)M, &s, &c);
        double2 num = make_double2(f.gain * scale, 0.0), den = make_double2(1.0, 0.0);
        for (int i = 0; i < f.nz; ++i) num = zmul(num, make_double2(c - f.z[i].x, s - f.z[i].y));
        for (int i = 0; i < f.np; ++i) den = zmul(den, make_double2(c - f.p[i].x, s - f.p[i].y));
        double2 h = zdiv(num, den);
        if (CONJ) h.y = -h.y;
        const float2 v = buf[k];
        buf[k] = make_float2((float)(v.x * h.x - v.y * h.y), (float)(v.x * h.y + v.y * h.x));
    }
}

// backward pass starts from the steady state of the last forward output
__global__ void iir_hold_kernel(float2* __restrict__ buf, int64_t end, int64_t L) {
    const float2 v = buf[end - 1];
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < L; t += (int64_t)gridDim.x * blockDim.x)
        buf[end + t] = v;
}

__global__ void iir_pick_kernel(const float2* __restrict__ buf, int64_t first, int q, int64_t m,
                                float2* __restrict__ y) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x)
        y[j] = buf[first + j * q];
}

// b = b * conj(a) * scale
__global__ void xspec_kernel(const float2* __restrict__ a, float2* __restrict__ b, int64_t M, float scale) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < M; k += (int64_t)gridDim.x * blockDim.x) {
        const float2 u = a[k], v = b[k];
        b[k] = make_float2((v.x * u.x + v.y * u.y) * scale, (v.y * u.x - v.x * u.y) * scale);
    }
}

#define AM_THREADS 256
// xc[i] = |c[K - i]|, per-block (max, first index)
__global__ __launch_bounds__(AM_THREADS) void xc_abs_argmax_kernel(const float2* __restrict__ c, int64_t K,
                                                                   float* __restrict__ xc, float* __restrict__ pv,
                                                                   int64_t* __restrict__ pi) {
    __shared__ float sv[AM_THREADS];
    __shared__ int64_t si[AM_THREADS];
    float bv = -1.f;
    int64_t bi = 0;
    for (int64_t i = (int64_t)blockIdx.x * AM_THREADS + threadIdx.x; i <= K; i += (int64_t)gridDim.x * AM_THREADS) {
        const float2 v = c[K - i];
        const float a = hypotf(v.x, v.y);
        if (xc) xc[i] = a;
        if (a > bv) { bv = a; bi = i; }          // i ascends per thread: first occurrence kept
    }
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = AM_THREADS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const float ov = sv[threadIdx.x + o];
            const int64_t oi = si[threadIdx.x + o];
            if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) {
                sv[threadIdx.x] = ov;
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { pv[blockIdx.x] = sv[0]; pi[blockIdx.x] = si[0]; }
}

__global__ __launch_bounds__(AM_THREADS) void argmax_final_kernel(const float* __restrict__ pv,
                                                                  const int64_t* __restrict__ pi, int np,
                                                                  int64_t* __restrict__ out) {
    __shared__ float sv[AM_THREADS];
    __shared__ int64_t si[AM_THREADS];
    float bv = -2.f;
    int64_t bi = 0;
    for (int t = threadIdx.x; t < np; t += AM_THREADS) {
        const float v = pv[t];
        const int64_t i = pi[t];
        if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = AM_THREADS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const float ov = sv[threadIdx.x + o];
            const int64_t oi = si[threadIdx.x + o];
            if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) {
                sv[threadIdx.x] = ov;
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = si[0];
}

static int check_iir(const prc_iir_desc* d, IirDev* f) {
    PRC_REQUIRE(d != nullptr, PRC_EINVAL, "iir descriptor is NULL");
    PRC_REQUIRE(d->q >= 1, PRC_EINVAL, "q must be a positive integer (got %d)", d->q);
    PRC_REQUIRE(d->padlen >= 0 && d->settle >= 1, PRC_EINVAL, "padlen >= 0 and settle >= 1 required");
    PRC_REQUIRE(d->nzeros >= 0 && d->nzeros <= 16 && d->npoles >= 0 && d->npoles <= 16, PRC_EINVAL,
                "at most 16 zeros and 16 poles (got %d, %d)", d->nzeros, d->npoles);
    PRC_REQUIRE((d->nzeros == 0 || d->zeros_host) && (d->npoles == 0 || d->poles_host), PRC_EINVAL,
                "zeros_host / poles_host is NULL");
    f->nz = d->nzeros;
    f->np = d->npoles;
    f->gain = d->gain;
    for (int i = 0; i < d->nzeros; ++i) f->z[i] = make_double2(d->zeros_host[2 * i], d->zeros_host[2 * i + 1]);
    for (int i = 0; i < d->npoles; ++i) {
        f->p[i] = make_double2(d->poles_host[2 * i], d->poles_host[2 * i + 1]);
        PRC_REQUIRE(f->p[i].x * f->p[i].x + f->p[i].y * f->p[i].y < 1.0, PRC_EINVAL, "pole %d is not inside the unit circle", i);
    }
    return PRC_OK;
}

static inline int grid_for(int64_t n) {
    const int64_t g = ceil_div64(n, 256);
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// filtfilt + [::q] of x (n) into y (ceil(n/q)); tmp holds M complex samples
static int run_filtfilt(const float2* x, int64_t n, const prc_iir_desc* d, const IirDev& f, float2* tmp, int64_t M,
                        Fft1d& fft, float2* y, hipStream_t st) {
    const int64_t L = d->settle, P = d->padlen, nx = n + 2 * P;
    iir_extend_kernel<<<grid_for(M), 256, 0, st>>>(x, n, (int)P, L, M, tmp);
    PRC_LAUNCH_CHECK();
    int rc = fft.run(true, tmp);
    if (rc) return rc;
    iir_response_kernel<false><<<grid_for(M), 256, 0, st>>>(tmp, M, f, 1.0 / (double)M);
    PRC_LAUNCH_CHECK();
    if ((rc = fft.run(false, tmp))) return rc;
    iir_hold_kernel<<<grid_for(L), 256, 0, st>>>(tmp, L + nx, L);
    PRC_LAUNCH_CHECK();
    if ((rc = fft.run(true, tmp))) return rc;
    iir_response_kernel<true><<<grid_for(M), 256, 0, st>>>(tmp, M, f, 1.0 / (double)M);
    PRC_LAUNCH_CHECK();
    if ((rc = fft.run(false, tmp))) return rc;
    const int64_t m = ceil_div64(n, d->q);
    iir_pick_kernel<<<grid_for(m), 256, 0, st>>>(tmp, L + P, d->q, m, y);
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}

static int64_t filt_len(int64_t n, const prc_iir_desc* d) {
    return (int64_t)next_pow2((size_t)(n + 2 * (int64_t)d->padlen + 2 * (int64_t)d->settle));
}

extern "C" int prc_decimate_iir(const void* x, int64_t n, const prc_iir_desc* host_iir, void* y, void* stream) {
    PRC_RANGE("prc_decimate_iir");
    PRC_REQUIRE(host_iir != nullptr, PRC_EINVAL, "iir descriptor is NULL");
    static_assert(sizeof(prc_iir_desc) == PRC_IIR_DESC_SIZE_600, "prc_iir_desc grew: keep PRC_IIR_DESC_SIZE_600, default the new fields to 0");
    prc_iir_desc mine;
    int rc = prc_take_desc(&mine, host_iir, PRC_IIR_DESC_SIZE_600, "prc_decimate_iir", "prc_iir_desc");
    if (rc) return rc;
    const prc_iir_desc* iir = &mine;
    IirDev f;
    rc = check_iir(iir, &f);
    if (rc) return rc;
    PRC_REQUIRE(x && y, PRC_EINVAL, "x / y is NULL");
    PRC_REQUIRE(n > iir->padlen, PRC_ESHAPE,
                "The length of the input vector x must be greater than padlen, which is %d.", iir->padlen);
    hipStream_t st = (hipStream_t)stream;
    const int64_t M = filt_len(n, iir);
    DevBuf tmp;
    PRC_HIP(hipMalloc(&tmp.p, (size_t)M * sizeof(float2)));
    Fft1d fft;
    if ((rc = fft.create((size_t)M, st))) return rc;
    rc = run_filtfilt((const float2*)x, n, iir, f, (float2*)tmp.p, M, fft, (float2*)y, st);
    if (rc) return rc;
    PRC_HIP(hipStreamSynchronize(st));        // tmp and the plans die here
    return PRC_OK;
}

extern "C" int prc_channel_offset(const void* s1, int64_t n1, const void* s2, int64_t n2, const prc_iir_desc* host_iir,
                                  int64_t nl, float* xc_out, int64_t* n_xc, int64_t* argmax_out, void* stream) {
    PRC_RANGE("prc_channel_offset");
    PRC_REQUIRE(host_iir != nullptr, PRC_EINVAL, "iir descriptor is NULL");
    static_assert(sizeof(prc_iir_desc) == PRC_IIR_DESC_SIZE_600, "prc_iir_desc grew: keep PRC_IIR_DESC_SIZE_600, default the new fields to 0");
    prc_iir_desc mine;
    int rc = prc_take_desc(&mine, host_iir, PRC_IIR_DESC_SIZE_600, "prc_channel_offset", "prc_iir_desc");
    if (rc) return rc;
    const prc_iir_desc* iir = &mine;
    IirDev f;
    rc = check_iir(iir, &f);
    if (rc) return rc;
    PRC_REQUIRE(s1 && s2 && argmax_out, PRC_EINVAL, "s1 / s2 / argmax_out is NULL");
    PRC_REQUIRE(nl >= 0, PRC_EINVAL, "nl must be >= 0");
    PRC_REQUIRE(n1 > iir->padlen && n2 > iir->padlen, PRC_ESHAPE,
                "The length of the input vector x must be greater than padlen, which is %d.", iir->padlen);
    const int64_t m1 = ceil_div64(n1, iir->q), m2 = ceil_div64(n2, iir->q);
    const int64_t K = m2 + 2 * nl - m1;
    PRC_REQUIRE(K >= 0, PRC_ESHAPE, "decimate(s2) padded by nl (%lld) is shorter than decimate(s1) (%lld)",
                (long long)(m2 + 2 * nl), (long long)m1);
    if (n_xc) *n_xc = K + 1;
    hipStream_t st = (hipStream_t)stream;
    const int64_t nmax = n1 > n2 ? n1 : n2;
    const int64_t M1 = filt_len(nmax, iir);
    const int64_t Mc = (int64_t)next_pow2((size_t)(m2 + 2 * nl));
    const int np = grid_for(K + 1) > 1024 ? 1024 : grid_for(K + 1);
    DevBuf tmp, A, B, pv, pi, res;
    PRC_HIP(hipMalloc(&tmp.p, (size_t)M1 * sizeof(float2)));
    PRC_HIP(hipMalloc(&A.p, (size_t)Mc * sizeof(float2)));
    PRC_HIP(hipMalloc(&B.p, (size_t)Mc * sizeof(float2)));
    PRC_HIP(hipMalloc(&pv.p, (size_t)np * sizeof(float)));
    PRC_HIP(hipMalloc(&pi.p, (size_t)np * sizeof(int64_t)));
    PRC_HIP(hipMalloc(&res.p, sizeof(int64_t)));
    PRC_HIP(hipMemsetAsync(A.p, 0, (size_t)Mc * sizeof(float2), st));
    PRC_HIP(hipMemsetAsync(B.p, 0, (size_t)Mc * sizeof(float2), st));
    {
        Fft1d fft;
        if ((rc = fft.create((size_t)M1, st))) return rc;
        if ((rc = run_filtfilt((const float2*)s1, n1, iir, f, (float2*)tmp.p, M1, fft, (float2*)A.p, st))) return rc;
        if ((rc = run_filtfilt((const float2*)s2, n2, iir, f, (float2*)tmp.p, M1, fft, (float2*)B.p + nl, st))) return rc;
        PRC_HIP(hipStreamSynchronize(st));
    }
    Fft1d fc;
    if ((rc = fc.create((size_t)Mc, st))) return rc;
    if ((rc = fc.run(true, A.p))) return rc;
    if ((rc = fc.run(true, B.p))) return rc;
    xspec_kernel<<<grid_for(Mc), 256, 0, st>>>((const float2*)A.p, (float2*)B.p, Mc, 1.0f / (float)Mc);
    PRC_LAUNCH_CHECK();
    if ((rc = fc.run(false, B.p))) return rc;
    xc_abs_argmax_kernel<<<np, AM_THREADS, 0, st>>>((const float2*)B.p, K, xc_out, (float*)pv.p, (int64_t*)pi.p);
    PRC_LAUNCH_CHECK();
    argmax_final_kernel<<<1, AM_THREADS, 0, st>>>((const float*)pv.p, (const int64_t*)pi.p, np, (int64_t*)res.p);
    PRC_LAUNCH_CHECK();
    PRC_HIP(hipMemcpyAsync(argmax_out, res.p, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    PRC_HIP(hipStreamSynchronize(st));
    return PRC_OK;
}
