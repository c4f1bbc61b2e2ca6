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

// MODE 0: team i takes pieces i, i + nteams, ... (the kernel's order); 1: team i takes a contiguous run of pieces;
// 2: as 0 with the chunk index fastest in the grid (x = chunk, y = team)
template <int LDC, int LDS_, int ST, bool WAIT_FIRST = false>
__global__ __launch_bounds__(256, 2) void k(const char* __restrict__ cache, const v2u* __restrict__ srv, v2u* __restrict__ out,
                                            int pieces, int B, int mode = 0, int stride = 0) {
    extern __shared__ float pad[];
    const int t = threadIdx.x;
    const int team = mode == 2 ? blockIdx.y : blockIdx.x, nteams = mode == 2 ? gridDim.y : gridDim.x;
    const int chunk = mode == 2 ? blockIdx.x : blockIdx.y;
    const int64_t C = stride ? stride : (int64_t)pieces * B;
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
    const int per = (pieces + nteams - 1) / nteams;
    const int p0 = mode == 1 ? team * per : team, pstep = mode == 1 ? 1 : nteams;
    const int pend = mode == 1 ? (p0 + per < pieces ? p0 + per : pieces) : pieces;
    issue(p0 < pend ? p0 : pieces);
    // mode bit 4: the first loads complete before the loop is entered.  Without it the loop header is reached with loads
    // pending (from here) or with loads FOLLOWED BY STORES pending (back edge); the compiler's wait must cover both, i.e.
    // s_waitcnt vmcnt(0) at the top of every iteration -- which also waits for the previous piece's stores
    if (WAIT_FIRST) __builtin_amdgcn_s_waitcnt(0x0F70);
    unsigned acc = 0;
    for (int p = p0; p < pend; p += pstep) {
        v2u res[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) res[r] = v2u{sv[r].x ^ x[r & 7].x, sv[r].y ^ x[r & 7].w};
#pragma unroll
        for (int m = 0; m < 8; ++m) acc += x[m].y + x[m].z;
        issue(p + pstep < pend ? p + pstep : pieces);
        const __amdgpu_buffer_rsrc_t ro = rsrc(o + (int64_t)p * B, (unsigned)B * 8u);
#pragma unroll
        for (int r = 0; r < 16; ++r) __builtin_amdgcn_raw_buffer_store_b64(res[r], ro, t * 8 + 2048 * r, 0, ST);
    }
    if (acc == 0x12345678u) pad[0] = 1.f;
}

static void run(void* c, void* s, void* o, int nchunks, int pieces, int B, int teams, int mode, int stride, const char* what, int wait_first = 0) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (double)nchunks * pieces * 32768 + 2.0 * (double)nchunks * pieces * B * 8;
    hipFuncSetAttribute((const void*)k<0, 0, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipFuncSetAttribute((const void*)k<0, 0, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        float ms = 0;
        hipEventRecord(e0);
        const dim3 grid = mode == 2 ? dim3(nchunks, teams) : dim3(teams, nchunks);
        if (wait_first) k<0, 0, 0, true><<<grid, 256, 69 * 1024>>>((const char*)c, (const v2u*)s, (v2u*)o, pieces, B, mode, stride);
        else k<0, 0, 0, false><<<grid, 256, 69 * 1024>>>((const char*)c, (const v2u*)s, (v2u*)o, pieces, B, mode, stride);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("%-58s B=%d teams=%3d: %.3f ms, %.2f TB/s\n", what, B, teams, best, bytes / best / 1e9);
}

int main() {
    const int nchunks = 256, pieces = 314;
    const size_t cache_b = (size_t)nchunks * pieces * 32768, str_b = (size_t)nchunks * pieces * 3840 * 8 + 4096;
    void *c, *s, *o;
    hipMalloc(&c, cache_b); hipMalloc(&s, str_b); hipMalloc(&o, str_b);
    hipMemset(c, 1, cache_b); hipMemset(s, 2, str_b);
    run(c, s, o, nchunks, pieces, 3831, 9, 0, 0, "strided pieces (round 3's order)");
    run(c, s, o, nchunks, pieces, 3840, 9, 0, 0, "  line-aligned pieces");
    run(c, s, o, nchunks, pieces, 3831, 9, 0, 0, "  first loads landed before the loop", 1);
    run(c, s, o, nchunks, pieces, 3831, 9, 1, 0, "  + a contiguous run of pieces per team", 1);
    run(c, s, o, nchunks, pieces, 3840, 9, 1, 0, "  + line-aligned (the kernel)", 1);
    for (int teams : {18, 36, 78, 157, 314}) run(c, s, o, nchunks, pieces, 3840, teams, 1, 0, "  the kernel's order with more teams per chunk", 1);
    for (int teams : {36, 78, 157}) run(c, s, o, nchunks, pieces, 3840, teams, 0, 0, "  strided, more teams per chunk", 1);
    run(c, s, o, nchunks, pieces, 3831, 314, 0, 0, "one piece per workgroup");
    run(c, s, o, nchunks, pieces, 3840, 314, 0, 0, "one piece per workgroup, line-aligned");
    return 0;
}
