// TEST ONLY: exposes the eight-wavefront 4096-point transform of passiveradar_amd/csrc/fft_team8.h on its own, so that
// the GPU tests can check it against numpy.fft directly (layouts of tools/fft4096x8_model.py).  Not part of libprcore.
#include "../../passiveradar_amd/csrc/fft_team8.h"
#include <math.h>

// mode 0: forward, output in the frequency layout [t][r] (row-major 512 x 8)
// mode 1: forward then inverse (time layout out, x 4096)
// mode 2: transforms back to back in the order the CAF kernel uses ((fwd, fwd, accumulate) x 3, inverse -- twice)
// mode 3: forward of a piece whose upper half is zero through the NZ = 4 form (must equal mode 0 on the same input)
// mode 4: inverse straight after an inverse (the schedule that needs the barrier at the head of f8_inv)
__global__ __launch_bounds__(F8_THREADS, 1) void probe8_kernel(const float2* x, const float2* y, float2* out,
                                                               const float2* gtab, int mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);
    const F8Lane f = f8_setup(lds, gtab);
    const int t = f.t;
    const float2* xb = x + (size_t)blockIdx.x * F8_P;
    const float2* yb = y + (size_t)blockIdx.x * F8_P;
    float2* ob = out + (size_t)blockIdx.x * F8_P;
    float2 u[8], v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { u[r] = xb[512 * r + t]; v[r] = yb[512 * r + t]; }
    if (mode == 0 || mode == 3) {
        if (mode == 3) f8_fwd<4>(u, f); else f8_fwd<8>(u, f);
#pragma unroll
        for (int r = 0; r < 8; ++r) ob[8 * t + r] = u[r];
    } else if (mode == 1) {
        f8_fwd(u, f);
        f8_inv(u, f);
#pragma unroll
        for (int r = 0; r < 8; ++r) ob[512 * r + t] = u[r];
    } else if (mode == 2) {
        float2 acc[8];
        for (int rep = 0; rep < 2; ++rep) {
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m] = make_float2(0.f, 0.f);
            for (int piece = 0; piece < 3; ++piece) {
                float2 a[8], b[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) { a[r] = u[r]; b[r] = v[r]; }
                f8_fwd(a, f);
                f8_fwd(b, f);
#pragma unroll
                for (int m = 0; m < 8; ++m) cmac_conj_a(acc[m], a[m], b[m]);
            }
            f8_inv(acc, f);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) ob[512 * r + t] = acc[r];
    } else {
        f8_fwd(u, f);
        f8_fwd(v, f);
        f8_inv(u, f);
        f8_inv(v, f);
#pragma unroll
        for (int r = 0; r < 8; ++r) ob[512 * r + t] = make_float2(u[r].x + 2.f * v[r].x, u[r].y + 2.f * v[r].y);
    }
}

extern "C" int fft_probe8(const void* x_host, const void* y_host, void* out_host, int nblocks, int mode) {
    float2 *dx = nullptr, *dy = nullptr, *dout = nullptr, *dtab = nullptr;
    const size_t bytes = sizeof(float2) * F8_P * (size_t)nblocks;
    float2* tab = new float2[F8_P];
    const double PI = 3.14159265358979323846;
    for (int m = 0; m < F8_P; ++m) {
        const double a = -2.0 * PI * (double)m / (double)F8_P;
        tab[m] = make_float2((float)cos(a), (float)sin(a));
    }
    if (hipMalloc(&dx, bytes) || hipMalloc(&dy, bytes) || hipMalloc(&dout, bytes) || hipMalloc(&dtab, sizeof(float2) * F8_P)) return -1;
    hipMemcpy(dx, x_host, bytes, hipMemcpyHostToDevice);
    hipMemcpy(dy, y_host, bytes, hipMemcpyHostToDevice);
    hipMemcpy(dtab, tab, sizeof(float2) * F8_P, hipMemcpyHostToDevice);
    delete[] tab;
    const size_t lds = sizeof(float2) * F8_LDS_ELEMS;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&probe8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) return -2;
    hipLaunchKernelGGL(probe8_kernel, dim3(nblocks), dim3(F8_THREADS), lds, 0, dx, dy, dout, dtab, mode);
    if (hipDeviceSynchronize() != hipSuccess) return -3;
    hipMemcpy(out_host, dout, bytes, hipMemcpyDeviceToHost);
    hipFree(dx); hipFree(dy); hipFree(dout); hipFree(dtab);
    return 0;
}
