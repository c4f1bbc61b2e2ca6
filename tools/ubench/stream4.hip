// Cache-policy bits on the access pattern of the fused LS kernel (4096-point form): per piece a workgroup of 256 threads reads a
// 32 KB spectrum block (16 bytes per lane), a 3831-sample piece of a complex64 stream (8 bytes per lane) and writes a
// 3831-sample piece, through raw buffer instructions whose aux field carries sc0 (1) / nt (2) / sc1 (16).
// hipcc -O3 --offload-arch=gfx950 stream4.hip -o stream4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, unsigned bytes) {
    const unsigned long long p = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
    void* q = (void*)(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, (short)0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

template <int LDC, int LDS_, int ST>
__global__ __launch_bounds__(256, 2) void k(const char* __restrict__ cache, const v2u* __restrict__ srv, v2u* __restrict__ out,
                                            int pieces, int B) {
    extern __shared__ float pad[];
    const int t = threadIdx.x;
    const int team = blockIdx.x, nteams = gridDim.x, chunk = blockIdx.y;
    const int64_t C = (int64_t)pieces * B;
    const char* cch = cache + (int64_t)chunk * pieces * 32768;
    const v2u* s = srv + chunk * C;
    v2u* o = out + chunk * C;
    v4u x[8];
    v2u sv[16];
    auto issue = [&](int p) {
        const bool live = p < pieces;
        const __amdgpu_buffer_rsrc_t rc = rsrc(cch + (int64_t)(live ? p : 0) * 32768, live ? 32768u : 0u);
#pragma unroll
        for (int m = 0; m < 8; ++m) x[m] = __builtin_amdgcn_raw_buffer_load_b128(rc, t * 16, 4096 * m, LDC);
        const __amdgpu_buffer_rsrc_t rs = rsrc(s + (int64_t)(live ? p : 0) * B, live ? (unsigned)B * 8u : 0u);
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[r] = __builtin_amdgcn_raw_buffer_load_b64(rs, t * 8 + 2048 * r, 0, LDS_);
    };
    issue(team);
    unsigned acc = 0;
    for (int p = team; p < pieces; p += nteams) {
        v2u res[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) res[r] = v2u{sv[r].x ^ x[r & 7].x, sv[r].y ^ x[r & 7].w};
#pragma unroll
        for (int m = 0; m < 8; ++m) acc += x[m].y + x[m].z;
        issue(p + nteams);
        const __amdgpu_buffer_rsrc_t ro = rsrc(o + (int64_t)p * B, (unsigned)B * 8u);
#pragma unroll
        for (int r = 0; r < 16; ++r) __builtin_amdgcn_raw_buffer_store_b64(res[r], ro, t * 8 + 2048 * r, 0, ST);
    }
    if (acc == 0x12345678u) pad[0] = 1.f;
}

template <int LDC, int LDS_, int ST>
static void run(void* c, void* s, void* o, int nchunks, int pieces, int B) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (double)nchunks * pieces * 32768 + 2.0 * (double)nchunks * pieces * B * 8;
    hipFuncSetAttribute((const void*)k<LDC, LDS_, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        float ms = 0;
        hipEventRecord(e0);
        k<LDC, LDS_, ST><<<dim3(9, nchunks), 256, 69 * 1024>>>((const char*)c, (const v2u*)s, (v2u*)o, pieces, B);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("aux: cache loads %2d, stream loads %2d, stores %2d : %.3f ms, %.2f TB/s\n", LDC, LDS_, ST, best, bytes / best / 1e9);
}

int main() {
    const int nchunks = 256, pieces = 314, B = 3831;
    const size_t cache_b = (size_t)nchunks * pieces * 32768, str_b = (size_t)nchunks * pieces * B * 8;
    void *c, *s, *o;
    hipMalloc(&c, cache_b); hipMalloc(&s, str_b); hipMalloc(&o, str_b);
    hipMemset(c, 1, cache_b); hipMemset(s, 2, str_b);
    run<0, 0, 0>(c, s, o, nchunks, pieces, B);
    run<2, 2, 0>(c, s, o, nchunks, pieces, B);
    run<0, 0, 2>(c, s, o, nchunks, pieces, B);
    run<2, 2, 2>(c, s, o, nchunks, pieces, B);
    run<16, 16, 0>(c, s, o, nchunks, pieces, B);
    run<17, 17, 0>(c, s, o, nchunks, pieces, B);
    run<18, 18, 18>(c, s, o, nchunks, pieces, B);
    run<0, 0, 17>(c, s, o, nchunks, pieces, B);
    run<0, 0, 19>(c, s, o, nchunks, pieces, B);
    run<2, 0, 0>(c, s, o, nchunks, pieces, B);
    run<0, 2, 0>(c, s, o, nchunks, pieces, B);
    run<0, 0, 0>(c, s, o, nchunks, pieces, B);
    return 0;
}
