// CFAR_2D on device (SURVEY 8f "next" #3): target_detection.py:683-703 as called per frame at
// range_doppler_plot.py:56-57 on |xambg|.
//   Tfilt = ones(fw,fw)/(fw^2-gw^2), Tfilt[e1:e2, e1:e2] = 0, e1 = (fw-gw)//2, e2 = fw-e1+1
//   CR = (X / mean|X|) / (convolve2d(X, Tfilt, 'same', boundary='wrap') + 1e-10)
// 'same' centring of scipy.signal.convolve2d: out[i,j] = sum_{a,b} T[a,b] X[(i + (fw-1)/2 - a) mod H,
// (j + (fw-1)/2 - b) mod W]  (asymmetric for even fw -- reproduced, not "fixed").
// One pass for the mean (block partials), one pass for the box sums from an LDS tile with wrap halo.
#include "common.h"
#include <vector>

#define CF_TX 32
#define CF_TY 8

// The map comes in as magnitudes (float: CFAR_2D's own argument) or as the complex range-Doppler map itself (float2:
// range_doppler_plot.py:56-57 calls CFAR_2D(np.abs(xambg), ...) -- |X| is then taken on the tile load, one read of the
// complex map instead of an abs kernel's read + write and two reads of its result; np.abs of complex64 is hypotf).
__device__ __forceinline__ float cfar_mag(float v) { return v; }
__device__ __forceinline__ float cfar_mag(float2 v) { return hypotf(v.x, v.y); }

template <typename TIn>
__global__ void cfar_abs_partial_kernel(const TIn* __restrict__ X, int64_t n, float* __restrict__ partial) {
    __shared__ float red[256];
    const int f = blockIdx.y;
    const TIn* x = X + (int64_t)f * n;
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += fabsf(cfar_mag(x[i]));
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(int64_t)f * gridDim.x + blockIdx.x] = red[0];
}

template <typename TIn>
__global__ __launch_bounds__(CF_TX * CF_TY) void cfar_kernel(const TIn* __restrict__ X, int H, int W, int fw,
                                                             int e1, int e2, float inv_cells,
                                                             const float* __restrict__ partial, int npartial,
                                                             float thresh, int use_thresh, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* tile = reinterpret_cast<float*>(smem_raw);        // (CF_TY + fw - 1) x (CF_TX + fw - 1)
    const int f = blockIdx.z;
    const TIn* x = X + (int64_t)f * H * W;
    const int c = (fw - 1) / 2;
    const int th = CF_TY + fw - 1, tw = CF_TX + fw - 1;
    const int i0 = blockIdx.y * CF_TY, j0 = blockIdx.x * CF_TX;
    // tile element (r, s) <-> X[(i0 + c - (fw-1) + r) mod H, (j0 + c - (fw-1) + s) mod W]
    const int ib = i0 + c - (fw - 1), jb = j0 + c - (fw - 1);
    for (int t = threadIdx.x; t < th * tw; t += CF_TX * CF_TY) {
        const int r = t / tw, s = t - r * tw;
        int ii = (ib + r) % H, jj = (jb + s) % W;
        if (ii < 0) ii += H;
        if (jj < 0) jj += W;
        tile[t] = cfar_mag(x[(int64_t)ii * W + jj]);
    }
    float tot = 0.f;
    for (int k = 0; k < npartial; ++k) tot += partial[(int64_t)f * npartial + k];
    const float mean_abs = tot / (float)((int64_t)H * W);
    __syncthreads();
    const int ty = threadIdx.x / CF_TX, tx = threadIdx.x % CF_TX;
    const int i = i0 + ty, j = j0 + tx;
    if (i >= H || j >= W) return;
    // X index for tap (a, b): row i + c - a  -> tile row (fw-1) + ty - a
    float acc = 0.f;
    for (int a = 0; a < fw; ++a) {
        const float* row = tile + ((fw - 1) + ty - a) * tw + (fw - 1) + tx;
        const bool guard_row = a >= e1 && a < e2;
        for (int b = 0; b < fw; ++b) {
            if (guard_row && b >= e1 && b < e2) continue;
            acc += row[-b];
        }
    }
    const float xv = tile[((fw - 1) + ty - c) * tw + (fw - 1) + tx - c];
    const float cr = (xv / mean_abs) / (acc * inv_cells + 1e-10f);
    out[(int64_t)f * H * W + (int64_t)i * W + j] = use_thresh ? (cr > thresh ? 1.f : 0.f) : cr;
}


// The same sums, separably (round 4).  The annulus is rows of full width outside the guard rows and two side pieces
// inside them: per tile row and column the kernel first forms  Ho = sum over the taps outside [e1, e2)  and  Hf = Ho +
// the taps inside, then an output adds fw of those down its column (Hf, or Ho on a guard row).  2 fw LDS reads per
// output instead of fw^2 - (gw+1)^2 (36 against 299 at CFAR_2D(18, 4)), no subtraction of a large box from a larger
// one (a strong cell under test never meets the noise it is compared with in one sum).  Tile 16 x 64 outputs per
// workgroup; the wrap-around indices are advanced, not divided.
#define CS_TX 64
#ifndef CS_TY
#define CS_TY 16
#endif
// FW > 0: the box and guard widths are compile-time constants (the instantiation for the reference's own call,
// CFAR_2D(18, 4) at range_doppler_plot.py:57): the 2 fw-term sums unroll into straight-line LDS reads at immediate
// offsets.  With run-time bounds every term costs a scalar add, compare and branch next to its LDS read and add -- the SQ
// counters had SQ_INSTS_SALU at 0.74 of SQ_INSTS_VALU for this kernel, and a CU has ONE scalar unit for all its wavefronts
// (profiles/r05_salu.md).  FW = 0: run-time widths (any other call).
template <typename TIn, int FW, int GW>
__global__ __launch_bounds__(256) void cfar_sep_kernel(const TIn* __restrict__ X, int H, int W, int fw_rt, int e1_rt, int e2_rt,
                                                       float inv_cells, const float* __restrict__ partial, int npartial,
                                                       float thresh, int use_thresh, float* __restrict__ out) {
    constexpr int E1C = FW > 0 ? ((FW - GW) / 2 < 0 ? 0 : (FW - GW) / 2) : 0;
    constexpr int E2C = FW > 0 ? (FW - E1C + 1 > FW ? FW : FW - E1C + 1) : 0;
    const int fw = FW > 0 ? FW : fw_rt, e1 = FW > 0 ? E1C : e1_rt, e2 = FW > 0 ? E2C : e2_rt;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int th = CS_TY + fw - 1, tw = CS_TX + fw - 1, twp = tw | 1;
    float* tile = reinterpret_cast<float*>(smem_raw);        // th x twp
    float* Ho = tile + th * twp;                             // th x CS_TX
    float* Hf = Ho + th * CS_TX;
    const int f = blockIdx.z;
    const TIn* x = X + (int64_t)f * H * W;
    const int c = (fw - 1) / 2;
    const int i0 = blockIdx.y * CS_TY, j0 = blockIdx.x * CS_TX;
    const int tx = threadIdx.x & 63, tq = threadIdx.x >> 6;
    // tile element (r, s) <-> X[(i0 + c - (fw-1) + r) mod H, (j0 + c - (fw-1) + s) mod W]
    int ii = (i0 + c - (fw - 1) + tq) % H;
    if (ii < 0) ii += H;
    const int rstep = 4 % H;
    for (int r = tq; r < th; r += 4) {
        int jj = (j0 + c - (fw - 1) + tx) % W;
        if (jj < 0) jj += W;
        const int sstep = 64 % W;
        for (int s_ = tx; s_ < tw; s_ += 64) {
            tile[r * twp + s_] = cfar_mag(x[(int64_t)ii * W + jj]);
            jj += sstep;
            if (jj >= W) jj -= W;
        }
        ii += rstep;
        if (ii >= H) ii -= H;
    }
    float tot = 0.f;
    for (int k = 0; k < npartial; ++k) tot += partial[(int64_t)f * npartial + k];
    const float mean_abs = tot / (float)((int64_t)H * W);
    __syncthreads();
    for (int r = tq; r < th; r += 4) {
        const float* row = tile + r * twp + (fw - 1) + tx;   // tap b reads row[-b]
        float so = 0.f, sm = 0.f;
        if (FW > 0) {
#pragma unroll
            for (int b = 0; b < E1C; ++b) so += row[-b];
#pragma unroll
            for (int b = E1C; b < E2C; ++b) sm += row[-b];
#pragma unroll
            for (int b = E2C; b < FW; ++b) so += row[-b];
        } else {
            for (int b = 0; b < e1; ++b) so += row[-b];
            for (int b = e1; b < e2; ++b) sm += row[-b];
            for (int b = e2; b < fw; ++b) so += row[-b];
        }
        Ho[r * CS_TX + tx] = so;
        Hf[r * CS_TX + tx] = so + sm;
    }
    __syncthreads();
    const int j = j0 + tx;
    if (j >= W) return;
    for (int ty = tq; ty < CS_TY; ty += 4) {
        const int i = i0 + ty;
        if (i >= H) break;
        float acc = 0.f;
        const int base = ((fw - 1) + ty) * CS_TX + tx;       // tap a reads [base - a CS_TX]
        if (FW > 0) {
#pragma unroll
            for (int a = 0; a < E1C; ++a) acc += Hf[base - a * CS_TX];
#pragma unroll
            for (int a = E1C; a < E2C; ++a) acc += Ho[base - a * CS_TX];
#pragma unroll
            for (int a = E2C; a < FW; ++a) acc += Hf[base - a * CS_TX];
        } else {
            for (int a = 0; a < e1; ++a) acc += Hf[base - a * CS_TX];
            for (int a = e1; a < e2; ++a) acc += Ho[base - a * CS_TX];
            for (int a = e2; a < fw; ++a) acc += Hf[base - a * CS_TX];
        }
        const float xv = tile[((fw - 1) + ty - c) * twp + (fw - 1) + tx - c];
        const float cr = (xv / mean_abs) / (acc * inv_cells + 1e-10f);
        out[(int64_t)f * H * W + (int64_t)i * W + j] = use_thresh ? (cr > thresh ? 1.f : 0.f) : cr;
    }
}

template <typename TIn>
static int cfar_run(const TIn* X, int32_t H, int32_t W, int32_t fw, int32_t gw, int32_t use_thresh,
                    float thresh, float* out, int32_t nframes, void* stream_) {
    PRC_REQUIRE(X && out, PRC_EINVAL, "prc_cfar2d: null argument");
    PRC_REQUIRE(H > 0 && W > 0 && fw > 0 && gw >= 0 && nframes > 0, PRC_EINVAL, "prc_cfar2d: bad size");
    PRC_REQUIRE(fw * fw != gw * gw, PRC_EINVAL, "prc_cfar2d: fw^2 == gw^2 divides by zero (as in the reference)");
    hipStream_t stream = (hipStream_t)stream_;
    const int np = 64;
    // scratch for the |X| partial sums: per host thread AND per (device, stream), grown on demand and kept.  Two
    // calls from one thread on different streams never share a buffer (the second call's partial-sum kernel could
    // otherwise overwrite what the first call's cfar_kernel has not read yet); calls on one stream are ordered by
    // the stream.  (Stream-ordered pool allocation -- hipMallocAsync / hipFreeAsync -- intermittently handed the
    // block to a later call while this one's kernels were still queued: whole maps came back scaled by a wrong mean.)
    struct Scratch { float* p = nullptr; size_t cap = 0; int dev = -1; hipStream_t stream = nullptr; };
    static thread_local std::vector<Scratch> pool;
    int dev = 0;
    PRC_HIP(hipGetDevice(&dev));
    const size_t need = sizeof(float) * (size_t)np * (size_t)nframes;
    Scratch* hit = nullptr;
    for (Scratch& c : pool)
        if (c.dev == dev && c.stream == stream) { hit = &c; break; }
    if (!hit) {
        // a host cycling through many streams: once THIS device holds 16 blocks, drop this device's blocks (after
        // synchronising it -- hipDeviceSynchronize covers the current device only, so blocks on other devices, whose
        // kernels may still be queued, are left alone)
        size_t mine = 0;
        for (const Scratch& c : pool) mine += c.dev == dev;
        if (mine >= 16) {
            PRC_HIP(hipDeviceSynchronize());
            std::vector<Scratch> keep;
            for (Scratch& c : pool) {
                if (c.dev == dev) (void)hipFree(c.p);
                else keep.push_back(c);
            }
            pool.swap(keep);
        }
        pool.push_back(Scratch());
        hit = &pool.back();
        hit->dev = dev;
        hit->stream = stream;
    }
    Scratch& sc = *hit;
    if (sc.cap < need) {
        if (sc.p) {
            PRC_HIP(hipStreamSynchronize(stream));    // only this stream ever used the block
            (void)hipFree(sc.p);
            sc.p = nullptr;
            sc.cap = 0;
        }
        PRC_HIP(hipMalloc((void**)&sc.p, need));
        sc.cap = need;
    }
    float* d_partial = sc.p;
    hipLaunchKernelGGL(cfar_abs_partial_kernel<TIn>, dim3(np, nframes), dim3(256), 0, stream, X, (int64_t)H * W, d_partial);
    int e1 = (fw - gw) / 2, e2 = fw - e1 + 1;
    if (e1 < 0) e1 = 0;
    if (e2 > fw) e2 = fw;
    const float inv_cells = 1.0f / (float)(fw * fw - gw * gw);
    const size_t th = (size_t)CS_TY + fw - 1, twp = ((size_t)CS_TX + fw - 1) | 1;
    const size_t lds_sep = sizeof(float) * (th * twp + 2 * th * CS_TX);
    if (lds_sep <= 64 * 1024 && prc_opt(PRC_OPT_CFAR_METHOD) != 1) {
        dim3 grid((W + CS_TX - 1) / CS_TX, (H + CS_TY - 1) / CS_TY, nframes);
        if (fw == 18 && gw == 4)
            hipLaunchKernelGGL((cfar_sep_kernel<TIn, 18, 4>), grid, dim3(256), lds_sep, stream, X, H, W, fw, e1, e2, inv_cells,
                               d_partial, np, thresh, use_thresh, out);
        else
            hipLaunchKernelGGL((cfar_sep_kernel<TIn, 0, 0>), grid, dim3(256), lds_sep, stream, X, H, W, fw, e1, e2, inv_cells,
                               d_partial, np, thresh, use_thresh, out);
    } else {
        const size_t lds = sizeof(float) * (size_t)(CF_TY + fw - 1) * (CF_TX + fw - 1);
        PRC_REQUIRE(lds <= 64 * 1024, PRC_EUNSUPPORTED, "prc_cfar2d: kernel width %d too large", fw);
        dim3 grid((W + CF_TX - 1) / CF_TX, (H + CF_TY - 1) / CF_TY, nframes);
        hipLaunchKernelGGL(cfar_kernel<TIn>, grid, dim3(CF_TX * CF_TY), lds, stream, X, H, W, fw, e1, e2, inv_cells, d_partial, np,
                           thresh, use_thresh, out);
    }
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) { prc_set_error("prc_cfar2d: launch failed: %s", hipGetErrorString(le)); return PRC_EHIP; }
    return PRC_OK;
}

extern "C" int prc_cfar2d(const float* X, int32_t H, int32_t W, int32_t fw, int32_t gw, int32_t use_thresh,
                          float thresh, float* out, int32_t nframes, void* stream) {
    PRC_RANGE("prc_cfar2d");
    return cfar_run<float>(X, H, W, fw, gw, use_thresh, thresh, out, nframes, stream);
}

extern "C" int prc_cfar2d_c64(const void* X, int32_t H, int32_t W, int32_t fw, int32_t gw, int32_t use_thresh,
                              float thresh, float* out, int32_t nframes, void* stream) {
    PRC_RANGE("prc_cfar2d_c64");
    return cfar_run<float2>((const float2*)X, H, W, fw, gw, use_thresh, thresh, out, nframes, stream);
}
