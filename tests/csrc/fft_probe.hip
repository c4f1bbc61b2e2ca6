// TEST ONLY: exposes the 4096-point team transform of passiveradar_amd/csrc/fft_team.h on its own, so that the
// GPU tests can check it against numpy.fft directly (layouts of tools/fft4096_model.py).  Not part of libprcore.
#include "../../passiveradar_amd/csrc/fft_team.h"
#include <math.h>

static void make_tables(float2* t) {
    const double PI = 3.14159265358979323846;
    for (int k1 = 0; k1 < 16; ++k1)
        for (int n2 = 0; n2 < 16; ++n2) {
            const double a = -2.0 * PI * (double)(k1 * n2) / 256.0;
            t[k1 * 16 + n2] = make_float2((float)cos(a), (float)sin(a));
        }
    for (int m = 0; m < FT_P; ++m) {
        const double a = -2.0 * PI * (double)m / (double)FT_P;
        t[FT_TW1 + m] = make_float2((float)cos(a), (float)sin(a));
    }
}

// mode 0: forward, output in the frequency layout [t][r] (row-major 256 x 16)
// mode 1: forward then inverse (time layout out, x 4096)
// mode 2: several transforms back to back in the orders the kernels use (alternation / barrier schedule): returns
//         inverse(conj(FFT(u)) * FFT(v)) like the CAF kernel, twice, to catch hazards between consecutive transforms
__global__ __launch_bounds__(FT_THREADS, 2) void probe_kernel(const float2* x, const float2* y, float2* out,
                                                              const float2* gtab, int mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);
    const FtLane f = ft_setup(lds, gtab);
    const int t = f.t;
    const float2* xb = x + (size_t)blockIdx.x * FT_P;
    const float2* yb = y + (size_t)blockIdx.x * FT_P;
    float2* ob = out + (size_t)blockIdx.x * FT_P;
    float2 u[16], v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { u[r] = xb[256 * r + t]; v[r] = yb[256 * r + t]; }
    if (mode == 0) {
        ft4096_fwd<0>(u, f);
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[16 * t + r] = u[r];
    } else if (mode == 1) {
        ft4096_fwd<0>(u, f);
        ft4096_inv<1>(u, f);
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[256 * r + t] = u[r];
    } else {
        float2 acc[16];
        for (int rep = 0; rep < 2; ++rep) {
#pragma unroll
            for (int m = 0; m < 16; ++m) acc[m] = make_float2(0.f, 0.f);
            for (int piece = 0; piece < 3; ++piece) {
                float2 a[16], b[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) { a[r] = u[r]; b[r] = v[r]; }
                ft4096_fwd<0>(a, f);
                ft4096_fwd<1>(b, f);
#pragma unroll
                for (int m = 0; m < 16; ++m) cmac_conj_a(acc[m], a[m], b[m]);
            }
            ft4096_inv<1>(acc, f);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[256 * r + t] = acc[r];
    }
}

extern "C" int fft_probe(const void* x_host, const void* y_host, void* out_host, int nblocks, int mode) {
    float2 *dx = nullptr, *dy = nullptr, *dout = nullptr, *dtab = nullptr;
    const size_t bytes = sizeof(float2) * FT_P * (size_t)nblocks;
    float2* tab = new float2[FT_GTAB];
    make_tables(tab);
    if (hipMalloc(&dx, bytes) || hipMalloc(&dy, bytes) || hipMalloc(&dout, bytes) ||
        hipMalloc(&dtab, sizeof(float2) * FT_GTAB)) return -1;
    hipMemcpy(dx, x_host, bytes, hipMemcpyHostToDevice);
    hipMemcpy(dy, y_host, bytes, hipMemcpyHostToDevice);
    hipMemcpy(dtab, tab, sizeof(float2) * FT_GTAB, hipMemcpyHostToDevice);
    delete[] tab;
    const size_t lds = sizeof(float2) * FT_LDS_ELEMS;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) return -2;
    hipLaunchKernelGGL(probe_kernel, dim3(nblocks), dim3(FT_THREADS), lds, 0, dx, dy, dout, dtab, mode);
    if (hipDeviceSynchronize() != hipSuccess) return -3;
    hipMemcpy(out_host, dout, bytes, hipMemcpyDeviceToHost);
    hipFree(dx); hipFree(dy); hipFree(dout); hipFree(dtab);
    return 0;
}
