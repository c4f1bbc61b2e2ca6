// Doppler stage of fast_xambg (scipy.fftpack.fft(axis=0) + np.fft.fftshift, range_doppler_processing.py:89) as one
// column-FFT kernel over the row-major slow-time buffer; phases and index algebra in doppler_col.h.
#include "caf_internal.h"
#include "doppler_col.h"
#include <math.h>

// One tile per workgroup.  (A persistent, register-double-buffered form -- the next tile's loads in flight under the
// current tile's transform -- measured SLOWER on MI355X at every size, 0.447 vs 0.363 ms per 16 config-5 surfaces: the
// hardware's own queue of waiting workgroups overlaps tiles better than 32 more VGPRs per thread do.)
// Workgroups reach the XCDs round-robin in launch order and every XCD has its own L2: workgroup b takes tile
// (b % 8) * per_xcd + b / 8, so an XCD works through one contiguous run of tiles -- neighbouring tiles split 128-byte
// lines of y and out between them.
struct DopOuts {                       // per-channel output pointers of a multi-channel launch (one entry for a plain launch)
    float2* out[8];
    int64_t y_ch_stride;               // elements between the channels' surface blocks in y
    int per_ch;                        // tiles * frames of one channel
};

template <int F>
__global__ __launch_bounds__(DopCfg<F>::THREADS) void doppler_col_kernel(const float2* __restrict__ y, int64_t y_surface,
                                                                         DopOuts outs,
                                                                         const float2* __restrict__ tw, int cols,
                                                                         int tiles, int total) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dop_smem[];
    float2* lds = reinterpret_cast<float2*>(dop_smem);
    constexpr int Q = DopCfg<F>::Q, KT = DopCfg<F>::KT, F3 = DopCfg<F>::F3, E = DopCfg<F>::E;
    const int per_xcd = (total + 7) >> 3;
#ifdef DOP_ROUND_ROBIN          // A/B: tiles in launch order (neighbouring tiles on different XCDs)
    const int work = (int)blockIdx.x;
    (void)per_xcd;
#else
    const int work = (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3);
#endif
    if (work >= total) return;                                  // uniform
    const int ch = work / outs.per_ch, wch = work - ch * outs.per_ch;
    const int frame = wch / tiles, tile = wch - frame * tiles;
    float2* __restrict__ out = outs.out[ch];
    y += (int64_t)ch * outs.y_ch_stride;
    const int c = threadIdx.x % KT, p = threadIdx.x / KT;
    const int k = tile * KT + c;
    // Raw buffer accesses: bases are uniform, a thread's part of the offset is ONE 32-bit VGPR per side, and the register's
    // part (row r Q of the loads, the fftshift-ed output row of register m of the stores) is a uniform soffset -- no 64-bit
    // per-lane addresses.  Loads: the tile is a contiguous block of F rows of KT samples (columns of the last tile beyond
    // the surface hold whatever the buffer held: every column is transformed on its own and those are never stored).
    // Stores: columns beyond the surface get an out-of-range offset and are dropped by the hardware range check.
    const unsigned row_bytes = (unsigned)cols * 8u;
    const unsigned ld_thread = (unsigned)(p * KT + c) * 8u;                          // row p, column c of the tile
    const unsigned st_thread = (unsigned)((p / F3) + 16 * (p % F3) * E) * row_bytes;  // row k1 + 16 g E
    const unsigned limit = 0xFFFFFFF0u - 2u * (unsigned)F * row_bytes;
    const unsigned cb = k < cols ? (unsigned)k * 8u : 0xFFFFFFF8u - st_thread;
    const __amdgpu_buffer_rsrc_t ry = prc_rsrc(y + (int64_t)frame * y_surface + (int64_t)tile * F * KT, (unsigned)(F * KT) * 8u);
    const __amdgpu_buffer_rsrc_t ro = prc_rsrc(out + (int64_t)frame * F * cols, limit);
    float2 x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = prc_buf_load_c64(ry, ld_thread, (unsigned)(r * Q * KT) * 8u);
    const DopTw t = dop_load_twiddles<F>(tw, p);
    dop_stage1<F>(x, t);
    if (!DopCfg<F>::SPLIT) {
        dop_write1<F>(x, lds, p, c);
        __syncthreads();
        dop_read1<F>(x, lds, p, c);
        dop_stage2<F>(x, t);
        if (F3 > 1) {
            dop_write2<F>(x, lds, p, c);      // the slots this thread just read: no barrier in between
            __syncthreads();
            dop_read2<F>(x, lds, p, c);
            dop_stage3<F>(x);
        }
    } else {
        // the same exchanges one component at a time through float slots (half the LDS: two workgroups per CU).  A
        // read phase is closed by a barrier before the slots are written again; the X2 write of the real parts follows
        // the X1 read of the imaginary parts directly (own slots, as in the one-round form)
        float* ldf = reinterpret_cast<float*>(dop_smem);
        float2 z[16];
        dop_write1c<F, 0>(x, ldf, p, c);
        __syncthreads();
        dop_read1c<F, 0>(z, ldf, p, c);
        __syncthreads();
        dop_write1c<F, 1>(x, ldf, p, c);
        __syncthreads();
        dop_read1c<F, 1>(z, ldf, p, c);
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = z[r];
        dop_stage2<F>(x, t);
        dop_write2c<F, 0>(x, ldf, p, c);
        __syncthreads();
        dop_read2c<F, 0>(z, ldf, p, c);
        __syncthreads();
        dop_write2c<F, 1>(x, ldf, p, c);
        __syncthreads();
        dop_read2c<F, 1>(z, ldf, p, c);
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = z[r];
        dop_stage3<F>(x);
    }
#pragma unroll
    for (int m = 0; m < 16; ++m)
        prc_buf_store_c64(ro, st_thread + cb, (unsigned)dop_out_row_reg<F>(m) * row_bytes, x[m]);
}

void dop_make_table(float2* t, int F) {
    const double PI = 3.14159265358979323846;
    for (int m = 0; m < F; ++m) {
        const double a = -2.0 * PI * (double)m / (double)F;
        t[m] = make_float2((float)cos(a), (float)sin(a));
    }
}

// the kernel addresses a whole surface through ONE 32-bit buffer descriptor: in-range offsets (< F * row_bytes) must stay
// below its limit 0xFFFFFFF0 - 2 F row_bytes, i.e. 3 F cols 8 bytes < 2^32 (beyond that loads would read zeros and stores
// would be dropped silently; AUTO takes the rocFFT path there)
bool dop_supported(int F, int cols) {
    if (!(F == 256 || F == 512 || F == 1024 || F == 2048 || F == 4096)) return false;
    return 3.0 * (double)F * (double)cols * 8.0 < 4294967280.0;
}

int dop_tile_cols(int F) {
    switch (F) {
        case 256: return DopCfg<256>::KT;
        case 512: return DopCfg<512>::KT;
        case 1024: return DopCfg<1024>::KT;
        case 2048: return DopCfg<2048>::KT;
        case 4096: return DopCfg<4096>::KT;
    }
    return 0;
}

template <int F>
static int dop_launch_t(const float2* y, int64_t y_surface, int64_t y_ch_stride, float2* const* outs, int nch, const float2* tw,
                        int cols, int nframes, hipStream_t stream) {
    constexpr int KT = DopCfg<F>::KT;
    const size_t lds = DopCfg<F>::LDS_BYTES;
    static bool attr_done[16] = {false};
    int dev = 0;
    PRC_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16 || !attr_done[dev]) {
        PRC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&doppler_col_kernel<F>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev >= 0 && dev < 16) attr_done[dev] = true;
    }
    const int tiles = (cols + KT - 1) / KT;
    DopOuts o;
    for (int i = 0; i < 8; ++i) o.out[i] = i < nch ? outs[i] : nullptr;
    o.y_ch_stride = y_ch_stride;
    o.per_ch = tiles * nframes;
    const int total = o.per_ch * nch;
    dim3 grid((unsigned)(((total + 7) / 8) * 8));
    hipLaunchKernelGGL((doppler_col_kernel<F>), grid, dim3(DopCfg<F>::THREADS), lds, stream, y, y_surface, o, tw, cols,
                       tiles, total);
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}

// nch channels (illuminators) in one launch: channel i's nframes surfaces start at y + i * y_ch_stride and go to outs[i]
int dop_launch_multi(const float2* y, int64_t y_surface, int64_t y_ch_stride, float2* const* outs, int nch, const float2* tw,
                     int F, int cols, int nframes, hipStream_t stream) {
    PRC_REQUIRE(nch >= 1 && nch <= 8, PRC_EINVAL, "dop_launch_multi: %d channels", nch);
    switch (F) {
        case 256: return dop_launch_t<256>(y, y_surface, y_ch_stride, outs, nch, tw, cols, nframes, stream);
        case 512: return dop_launch_t<512>(y, y_surface, y_ch_stride, outs, nch, tw, cols, nframes, stream);
        case 1024: return dop_launch_t<1024>(y, y_surface, y_ch_stride, outs, nch, tw, cols, nframes, stream);
        case 2048: return dop_launch_t<2048>(y, y_surface, y_ch_stride, outs, nch, tw, cols, nframes, stream);
        case 4096: return dop_launch_t<4096>(y, y_surface, y_ch_stride, outs, nch, tw, cols, nframes, stream);
    }
    prc_set_error("dop_launch: unsupported freq_bins %d", F);
    return PRC_EUNSUPPORTED;
}

int dop_launch(const float2* y, int64_t y_surface, float2* out, const float2* tw, int F, int cols, int nframes, hipStream_t stream) {
    float2* one[1] = {out};
    return dop_launch_multi(y, y_surface, 0, one, 1, tw, F, cols, nframes, stream);
}
