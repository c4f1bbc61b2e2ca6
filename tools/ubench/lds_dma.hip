// Semantics probe for buffer_load_dwordx4 ... lds on gfx950 (global -> LDS without VGPRs).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
template <int MODE>
__global__ void k(const float2* src, float2* dst, int n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2* st = reinterpret_cast<float2*>(smem);
    for (int i = threadIdx.x; i < 1024; i += 64) st[i] = make_float2(-1.f, -1.f);
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n * 8, 0x00020000);
    const int lane = threadIdx.x;
    if (MODE == 0) {          // one LDS base, instruction offsets on both sides
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(st), 16, lane * 16, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(st), 16, lane * 16, 0, 1024, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(st), 16, lane * 16, 0, 2048, 0);
    } else {                  // LDS pointer advanced, global offset in voffset
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(st), 16, lane * 16, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(st + 128), 16, lane * 16 + 1024, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(st + 256), 16, lane * 16 + 2048, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) dst[i] = st[i];
}
// four waves, private landing zones, no workgroup barrier between the DMA and the read (the fused LS kernel's use)
__global__ void k4(const float2* src, float2* dst, int n, int zone_off) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    float2* st = reinterpret_cast<float2*>(smem) + zone_off + wave * 768;
    for (int it = 0; it < 3; ++it) {
        const float2* s0 = src + (wave * 3 + it) * 100;
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)s0, 0, n * 8, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(st), 16, lane * 16, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(st), 16, lane * 16, 0, 1024, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(st), 16, lane * 16, 0, 2048, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        float2 v[6];
        for (int q = 0; q < 6; ++q) v[q] = st[64 * q + lane];
        for (int q = 0; q < 6; ++q) dst[((wave * 3 + it) * 6 + q) * 64 + lane] = v[q];
        __builtin_amdgcn_wave_barrier();
    }
}
int main() {
    const int N = 512;
    float2 h[N], o[N];
    for (int i = 0; i < N; ++i) h[i] = make_float2((float)i, 0.5f);
    float2 *d, *e;
    hipMalloc(&d, sizeof(h)); hipMalloc(&e, sizeof(o));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) for (int n : {512, 301}) {
        if (mode == 0) k<0><<<1, 64, 16384>>>(d, e, n); else k<1><<<1, 64, 16384>>>(d, e, n);
        hipMemcpy(o, e, sizeof(o), hipMemcpyDeviceToHost);
        int good = 0, firstbad = -1;
        for (int i = 0; i < 384; ++i) { const float want = i < n ? (float)i : 0.f; if (o[i].x == want) ++good; else if (firstbad < 0) firstbad = i; }
        printf("mode %d n %d: %d/384 as expected, first mismatch %d (got %.1f), o[300]=%.1f o[301]=%.1f o[383]=%.1f o[384]=%.1f\n", mode, n, good, firstbad,
               firstbad >= 0 ? o[firstbad].x : 0.f, o[300].x, o[301].x, o[383].x, o[384].x);
    }
    {
        const int NS = 4096;
        float2* hs = new float2[NS]; float2* ho = new float2[4 * 3 * 384];
        for (int i = 0; i < NS; ++i) hs[i] = make_float2((float)i, 1.f);
        float2 *ds, *dd; hipMalloc(&ds, NS * 8); hipMalloc(&dd, 4 * 3 * 384 * 8);
        hipMemcpy(ds, hs, NS * 8, hipMemcpyHostToDevice);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k4), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        for (int zone : {0, 6000, 8800}) {
            k4<<<1, 256, 80 * 1024>>>(ds, dd, 301, zone);
            hipMemcpy(ho, dd, 4 * 3 * 384 * 8, hipMemcpyDeviceToHost);
            int good = 0;
            for (int w = 0; w < 4; ++w) for (int it = 0; it < 3; ++it) for (int i = 0; i < 384; ++i) {
                const float want = i < 301 ? (float)((w * 3 + it) * 100 + i) : 0.f;
                good += ho[(w * 3 + it) * 384 + i].x == want;
            }
            printf("4 waves, landing zones at float2 offset %d (byte %d): %d/%d as expected\n", zone, zone * 8, good, 4 * 3 * 384);
        }
    }
    return 0;
}
