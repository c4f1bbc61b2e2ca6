/*
 * prcore.h -- C ABI of libprcore.so, the MI355X (gfx950) range-Doppler core.
 *
 * The reference (Max-Manning/passiveRadar) is pure Python and has no FFI layer; its
 * boundary for this path is the Python function level.  Each entry
 * point below replaces one reference function (or the SciPy/NumPy sequence inside it) and
 * is what a reference-side ctypes stub binds (INTEGRATION.md shows the stubs).  Citations
 * are file:line in the reference checkout.
 *
 * Conventions
 *   - extern "C", plain ints / pointers / POD structs; no C++ types, no exceptions cross.
 *   - complex64 samples are interleaved float pairs (re, im): `const void*` to 8-byte items.
 *   - all data pointers are DEVICE pointers unless the name ends in `_host`.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Work is enqueued
 *     asynchronously; nothing synchronises unless stated.
 *   - caller owns inputs and outputs; the library owns plans and their workspaces.
 *   - HOST arrays (names ending in `_host`) are read before the call returns and never afterwards: temporaries are fine.
 *   - every function returns PRC_OK (0) or a negative prc_status; prc_last_error() gives a
 *     thread-local message.  The Python layer maps PRC_ESHAPE to ValueError to keep the
 *     reference's error convention (range_doppler_processing.py:46-49, clutter_removal.py:28-29).
 *   - a plan may be used by one thread at a time (internally serialised by a mutex);
 *     different plans may be used concurrently from different threads (dask-style callers).
 *   - descriptors: every descriptor struct starts with `uint32_t struct_size`, which the host sets to sizeof(the
 *     struct) of the header IT was compiled against (ctypes: ctypes.sizeof), and `uint32_t magic` = PRC_DESC_MAGIC
 *     (PRC_DESC_INIT(d) zeroes a descriptor and sets both).  Descriptors only ever grow at the end.  The library reads
 *     min(struct_size, its own sizeof) bytes and takes fields the host did not know as 0 (= AUTO / default).  A wrong
 *     magic (a host built against a pre-600 header has the first field of the old layout there), a struct_size below
 *     the version-600 layout or not a multiple of 4 is PRC_EINVAL with a message naming the sizes -- a host built
 *     against another layout can never make the library read past its struct.
 */
#ifndef PRCORE_H
#define PRCORE_H

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRC_DESC_MAGIC 0x36435250u     /* "PRC6" */
/* zero a descriptor and fill in its header; then set the fields */
#define PRC_DESC_INIT(d) do { memset(&(d), 0, sizeof(d)); (d).struct_size = (uint32_t)sizeof(d); (d).magic = PRC_DESC_MAGIC; } while (0)

#define PRC_VERSION 600   /* 600: every descriptor (prc_caf_desc, prc_ls_desc, prc_frontend_desc, prc_iir_desc) starts with `struct_size`,
                             `magic` (layout break: rebuild hosts; from here on descriptors only grow at the end and an older host keeps
                             working); PRC_OPT_CAF_TEAM8; host arrays passed to an entry point are read before it returns;
                             500: prc_frequency_shift_phases, prc_frontend_execute2, prc_cfar2d_c64, prc_mem_info, PRC_OPT_MARKERS; 401: prc_comm_loopback, PRC_OPT_FE_METHOD, PRC_OPT_CFAR_METHOD; 400: prc_caf_desc.multi, prc_set_option / prc_get_option (no environment variables are read),
                             prc_comm_count; 310: prc_ls_desc.method = 4, NLMS up to 8192 taps */

typedef enum prc_status {
    PRC_OK = 0,
    PRC_EINVAL = -1,       /* bad argument (null pointer, non-positive size, ...)           */
    PRC_ESHAPE = -2,       /* length mismatch -> ValueError on the Python side             */
    PRC_EHIP = -3,         /* a HIP runtime call failed (no GPU, OOM, launch failure)      */
    PRC_EROCFFT = -4,      /* a rocFFT call failed                                          */
    PRC_EUNSUPPORTED = -5  /* valid request outside what this build implements              */
} prc_status;

/* ---- library / device ------------------------------------------------------------- */
int prc_version(void);
const char* prc_last_error(void);
int prc_device_count(int* count);
int prc_set_device(int device);
int prc_get_device(int* device);       /* the calling thread's current HIP device */
/* Device-memory helpers for hosts that do not bring their own allocator (the NumPy-facing
 * drop-in functions use these; torch-based callers pass tensor.data_ptr() instead). */
int prc_mem_info(size_t* free_bytes, size_t* total_bytes);   /* hipMemGetInfo of the current device */
int prc_malloc(void** dptr, size_t bytes);
int prc_free(void* dptr);
int prc_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream);
int prc_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream);
int prc_memset(void* dptr, int value, size_t bytes, void* stream);
int prc_stream_sync(void* stream);

/* Tuning options.  The library reads NO environment variables: kernel-selection knobs are fields of the plan
 * descriptors (method, doppler, multi) or these process-wide options, which a plan copies ONCE, at creation (a later
 * prc_set_option never changes an existing plan); prc_nlms_execute, which has no plan, reads PRC_OPT_NLMS_WAVES per call. */
typedef enum prc_option {
    PRC_OPT_CAF_MULTI_MODE = 0,   /* what prc_caf_desc.multi = PRC_CAF_MULTI_AUTO resolves to; default AUTO = the library's
                                     measured choice                                                                      */
    PRC_OPT_CAF_GROUP_MB = 1,     /* > 0: surfaces per segment/Doppler round of prc_caf_execute sized to this many MiB of
                                     slow-time buffer; default 0 = the whole batch in one round                           */
    PRC_OPT_LS_TEAM_PIECES = 2,   /* pieces per team of the 4096-point LS chain (default 32)                              */
    PRC_OPT_LS_TEAM_ALIGN = 3,    /* 1 (default): pieces of the 4096-point LS chain start on 128-byte lines; 0: E = T - 1  */
    PRC_OPT_NLMS_WAVES = 4,       /* 0 (default): wavefronts per NLMS stream from the tap count; 2 / 4: split a stream
                                     that would fit one wavefront over 2 / 4 (latency experiments)                        */
    PRC_OPT_LS_CACHE_LIMIT_MB = 5,/* > 0: an LS plan does not allocate a spectrum cache larger than this many MiB and runs
                                     the recomputing / per-bin kernels instead; default 0 = no limit                      */
    PRC_OPT_NLMS_WG_WAVES = 6,    /* 0 (default): NLMS wavefronts (= independent streams) per workgroup / CU from the stream count
                                     and the measured step times; 4 / 8 / 12 / 16: forced (16 = four per SIMD exists for the
                                     config-3 filter length only; A/B runs)                                                */
    PRC_OPT_CAF_XCD_CONTIG = 7,   /* workgroup order of the 4096-point segment kernel, read per launch (A/B runs): 0 (default) =
                                     segments go round the XCDs in launch order, 1 = every XCD takes a contiguous run of them
                                     (measured slower on MI355X)                                                            */
    PRC_OPT_CAF_PAIR_FRAMES = 8,  /* 4096-point segment kernel, frames overlapping by half, read per launch (A/B runs): 1 (default) =
                                     the two frames that cover the same samples run in consecutive slots of one XCD (config 5: -2 %
                                     at one channel, -3 % at four), 0 = frame after frame                                   */
    PRC_OPT_FE_METHOD = 9,        /* front-end kernel, read per launch: 0 (default) = the group form (`up` outputs per thread, taps
                                     through the scalar unit) where it applies (up <= 16, window within LDS), else one output per
                                     thread; 1 = one output per thread; 2 = the group form or PRC_EUNSUPPORTED                */
    PRC_OPT_CFAR_METHOD = 10,     /* prc_cfar2d, read per call: 0 (default) = separable sums (rows, then columns) where the tile fits
                                     LDS, 1 = every tap of the box per output                                               */
    PRC_OPT_MARKERS = 11,         /* 1: every prc_*_execute (and the other device entry points) opens a roctx range named after
                                     itself -- rocprofv3 --marker-trace shows the library's calls around its kernels.  The roctx
                                     library (librocprofiler-sdk-roctx.so.1, else libroctx64.so.4) is bound at run time, the first
                                     time the option is set; PRC_EUNSUPPORTED if neither loads.  Default 0: nothing is loaded    */
    PRC_OPT_FE_BALANCE = 12,      /* front-end plans, read at creation: 0 (default) = the group kernel's wavefronts take equal runs of
                                     tap rows at full width; N > 0 = runs of about equal cost (2 w + N per two rows), each run
                                     multiplying only the w output columns its rows reach (37 % fewer multiply-adds at 13:119;
                                     measured no faster at any N, 7 % slower at N = 8: A/B runs)                             */
    PRC_OPT_CAF_TEAM8 = 13,       /* 4096-point CAF segment kernel, read per launch: 0 = teams of four wavefronts, 16 points per
                                     thread (three wavefronts per SIMD); 1 = teams of EIGHT wavefronts, 8 points per thread (six per
                                     SIMD, a third LDS exchange per transform); default: the measured choice  */
    PRC_OPT_FE_FOLD = 14,         /* front-end group kernel, read per launch: 1 (default) = FOLDED tap rows where they apply (odd decimation:
                                     rows rho and rho + M of the banded tap table never meet the same output, so they share one row of
                                     full width: 184 rows instead of 294 at 13:119, no multiply-adds on zero taps); 0 = the unfolded rows  */
    PRC_OPT_COUNT_ = 15
} prc_option;
int prc_set_option(int32_t option, int64_t value);     /* PRC_EINVAL for an unknown option or a value out of range */
int prc_get_option(int32_t option, int64_t* value);

/* ---- cross-ambiguity surface: fast_xambg (range_doppler_processing.py:12-90) -------- */
typedef enum prc_caf_method {
    PRC_CAF_AUTO = 0,
    PRC_CAF_DIRECT = 1,    /* time-domain lagged products in LDS tiles (any decimation FIR) */
    PRC_CAF_FFT = 2,       /* per-segment FFT correlation held in LDS/registers (boxcar FIR):
                              1024-point transforms, one wavefront per segment                 */
    PRC_CAF_FFT4096 = 3    /* the same with 4096-point transforms by a team of four wavefronts:
                              what AUTO takes for wide range spans (1025 lags and more)       */
} prc_caf_method;

typedef enum prc_doppler_method {
    PRC_DOPPLER_AUTO = 0,   /* COLUMN when freq_bins is 256, 512, 1024, 2048 or 4096, else ROCFFT           */
    PRC_DOPPLER_ROCFFT = 1, /* transpose + batched 1-D rocFFT over the slow-time axis (any freq_bins) +
                               one fftshift/transpose kernel (:89)                           */
    PRC_DOPPLER_COLUMN = 2  /* one column-FFT kernel over the row-major slow-time buffer, fftshift folded
                               into the store: the surface is read once and written once    */
} prc_doppler_method;

/* How prc_caf_execute_multi runs several reference channels against one surveillance channel */
typedef enum prc_caf_multi_mode {
    PRC_CAF_MULTI_AUTO = 0,   /* the library's choice (PRC_OPT_CAF_MULTI_MODE overrides what AUTO means)               */
    PRC_CAF_MULTI_TURNS = 1,  /* one single-reference pass per illuminator (any method)                                */
    PRC_CAF_MULTI_SHARED = 2, /* 4096-point method: the surveillance pieces of a segment are transformed once for ALL
                                 illuminators (two wavefronts per SIMD)                                                */
    PRC_CAF_MULTI_PAIRS = 3   /* 4096-point method: once per PAIR of illuminators (three wavefronts per SIMD)           */
} prc_caf_multi_mode;

typedef struct prc_caf_desc {
    uint32_t struct_size;  /* sizeof(prc_caf_desc) as the host compiled it (see Conventions)   */
    uint32_t magic;        /* PRC_DESC_MAGIC                                                 */
    int64_t n;             /* samples per CPI after zero padding = inputLen (:52-55)         */
    int32_t range_bins;    /* rangeBins; the surface has range_bins+1 columns (:64)          */
    int32_t freq_bins;     /* freqBins; decimation q = (int)(n / freq_bins) (:61)            */
    int32_t max_frames;    /* workspace is sized for this many frames per execute           */
    int32_t method;        /* prc_caf_method                                                 */
    int32_t doppler;       /* prc_doppler_method                                             */
    int32_t ntaps;         /* 0: boxcar ones(q+1) (shortFilt=True, :72); else length of taps */
    const float* taps_host;/* HOST pointer, copied at plan creation (shortFilt=False, :76)   */
    int32_t multi;         /* prc_caf_multi_mode of prc_caf_execute_multi (0 = AUTO)         */
} prc_caf_desc;
#define PRC_CAF_DESC_SIZE_600 56u      /* sizeof(prc_caf_desc) at PRC_VERSION 600: the smallest struct_size accepted */

typedef struct prc_caf_plan prc_caf_plan;

int prc_caf_plan_create(prc_caf_plan** plan, const prc_caf_desc* desc);
int prc_caf_plan_destroy(prc_caf_plan* plan);
/* Which kernels the plan resolved AUTO to (prc_caf_method / prc_doppler_method values). */
int prc_caf_plan_info(const prc_caf_plan* plan, int32_t* method, int32_t* doppler,
                      int64_t* workspace_bytes);
/* The prc_caf_multi_mode the plan resolved AUTO to (TURNS / SHARED / PAIRS). */
int prc_caf_plan_multi_mode(const prc_caf_plan* plan, int32_t* multi);

/* out[f][k] for each frame: complex64 [nframes][freq_bins][range_bins+1], C order; frame b
 * reads ref/srv at element offset b*frame_stride (frame_stride = n for dense batches, = n/2
 * for the 50 %-overlapped frames of main.py:178-194).  Samples with index >= n_valid inside
 * a frame are taken as zero (the zero-pad branch :52-55); pass n_valid = n otherwise.
 * window: float32[n] device pointer or NULL (:83-84).  Column k <-> delay range_bins-k,
 * rows fftshift-ed, circular wrap of srv within the frame (:82) -- exactly as the reference. */
int prc_caf_execute(prc_caf_plan* plan, const void* ref, const void* srv, int64_t frame_stride,
                    int64_t n_valid, const float* window, void* out, int32_t nframes,
                    void* stream);
/* fast_xambg for nref (<= 8) reference channels against ONE surveillance channel in one call -- a multi-illuminator
 * frame (range_doppler_processing.py:12-90 once per pair; :81-86 is the per-pair unit): outs_host[i] receives what
 * prc_caf_execute(plan, refs_host[i], srv, ...) would write.  refs_host / outs_host are HOST arrays of nref DEVICE
 * pointers.  nframes * nref surfaces must fit the plan's max_frames.  prc_caf_desc.multi selects how (prc_caf_multi_mode):
 * the illuminators take turns through the single-reference kernels, or -- 4096-point method, segments of at most two
 * pieces (wide range spans: BASELINE configs 3 and 5) -- the surveillance pieces of a segment are transformed once for all
 * illuminators or once per pair (same results to the order of two sums).  A mode the plan's shape does not support falls
 * back to TURNS. */
int prc_caf_execute_multi(prc_caf_plan* plan, const void* const* refs_host, int32_t nref, const void* srv,
                          int64_t frame_stride, int64_t n_valid, const float* window, void* const* outs_host,
                          int32_t nframes, void* stream);
/* Stages timed separately by bench.py (same arguments; `segment` writes the plan's internal
 * slow-time buffer, `doppler` turns it into out). */
int prc_caf_execute_segments(prc_caf_plan* plan, const void* ref, const void* srv,
                             int64_t frame_stride, int64_t n_valid, const float* window,
                             int32_t nframes, void* stream);
int prc_caf_execute_doppler(prc_caf_plan* plan, void* out, int32_t nframes, void* stream);

/* ---- block least-squares clutter cancellers (clutter_removal.py) ------------------- */
typedef struct prc_ls_desc {
    uint32_t struct_size;  /* sizeof(prc_ls_desc) as the host compiled it (see Conventions)     */
    uint32_t magic;        /* PRC_DESC_MAGIC                                                 */
    int64_t n;             /* samples per block (chunk)                                     */
    int32_t filter_len;    /* filterLen                                                     */
    int32_t peek;          /* non-causal taps (default 10 in the reference)                 */
    int32_t circular;      /* 0: LS_Filter_Toeplitz semantics (:109-160: circular peek shift,
                              linear correlations, linear FIR); 1: LS_Filter semantics (:6-56:
                              circular data matrix => circular correlations and FIR)        */
    int32_t max_blocks;    /* workspace is sized for this many independent blocks           */
    int32_t method;        /* 0 auto (= 3 when it fits; = 4 from 120 taps on blocks of >= 65536 samples), 1 time-domain kernels, 2 FFT kernels that
                              recompute the reference spectra per Doppler bin, 3 FFT kernels with
                              the reference spectra cached in HBM between Doppler bins, 4 the chain
                              of method 3 on 4096-point transforms where it applies (linear form,
                              <= 769 taps, n >= 8192: fewer cache bytes per sample; else as 3).  The FFT
                              kernels take up to 769 taps on 1024-point transforms (one wavefront
                              each) and up to 3073 taps on 4096-point transforms (four wavefronts).
                              Any filter_len + peek < n runs (the reference, clutter_removal.py:109-160,
                              accepts any length): above 3073 taps correlations and FIR run on the time-domain
                              kernels (the FIR in tap tiles beyond ~9000 taps); the Levinson recursion keeps
                              its three complex128 T-vectors in LDS up to 3413 taps, the autocorrelation in a
                              global workspace up to 5120, all three there beyond (seconds at 10^4 taps)       */
} prc_ls_desc;
#define PRC_LS_DESC_SIZE_600 40u

typedef struct prc_ls_plan prc_ls_plan;

int prc_ls_plan_create(prc_ls_plan** plan, const prc_ls_desc* desc);
int prc_ls_plan_destroy(prc_ls_plan* plan);

/* LS_Filter_Multiple (:162-187) on nblocks independent blocks: for each Doppler bin in
 * order, cancel the (frequency-shifted, float32 phase -- signal_utils.py:24-27) reference from
 * the previous bin's output.  nbins = 1, bins = {0} is LS_Filter_Toeplitz; with
 * desc.circular = 1 and reg it is LS_Filter.  block b reads ref/srv at b*stride elements and
 * writes complex64 out at b*out_stride.  taps_out: optional complex128 [nblocks][T] (device)
 * taps of the LAST bin (return_filter=True), T = filter_len + peek.
 * doppler_bins_host: HOST array of nbins doubles (Hz). reg is added to the Gram diagonal. */
int prc_ls_execute(prc_ls_plan* plan, const void* ref, const void* srv, int64_t stride,
                   void* out, int64_t out_stride, int32_t nblocks, double sample_rate,
                   const double* doppler_bins_host, int32_t nbins, double reg,
                   void* taps_out, void* stream);

/* Per-kernel timing for bench.py's roofline: when enabled, HIP events are recorded on the
 * caller's stream around every kernel of the next prc_ls_execute.  prc_ls_get_profile waits
 * for them and returns ms[0..2] = correlation / Levinson / FIR kernel time of that execute
 * (summed over its Doppler bins) and the number of launches of each kind. */
int prc_ls_set_profiling(prc_ls_plan* plan, int32_t enable);
int prc_ls_get_profile(prc_ls_plan* plan, double* ms, int32_t* launches_per_kind);

/* ---- NLMS_filter (clutter_removal.py:189-249) ---------------------------------------- */
/* Any filter_len + peek < n runs (the reference, clutter_removal.py:189-249, takes any length): a stream's taps live in
 * registers, 32 per lane -- one wavefront up to 2048 taps, a workgroup of two up to 4096, of four up to 8192; beyond that a
 * plain kernel (one workgroup per stream, taps in a global workspace allocated for the call: microseconds per step, and
 * the call synchronises `stream`).
 * nstreams independent sample-recursive filters, one wavefront (or one such workgroup) each.  taps_in: optional
 * complex64 [nstreams][T] initial taps (initialTaps), NULL = zeros.  taps_out: optional
 * complex64 [nstreams][T].  out: complex64, zero outside [filter_len, n-peek). */
int prc_nlms_execute(const void* ref, const void* srv, int64_t n, int64_t stride,
                     int32_t filter_len, int32_t peek, float mu, const void* taps_in,
                     void* out, int64_t out_stride, void* taps_out, int32_t nstreams,
                     void* stream);

/* ---- helpers that sit on the path (signal_utils.py) ---------------------------------- */
/* xcorr (:29-32): z[i] = sum_n s1[n] conj(s2[n-(i-nlead)]), i = 0..nlag+nlead; complex64 out. */
int prc_xcorr(const void* s1, const void* s2, int64_t n, int32_t nlead, int32_t nlag,
              void* out, void* stream);
/* frequency_shift (:24-27): y[n] = x[n] exp(j(fl32 phase ramp + phase_offset)). */
int prc_frequency_shift(const void* x, void* y, int64_t n, double fc, double fs,
                        double phase_offset, void* stream);

/* ---- front end (SURVEY 8f "next" #1): main.py:105-166 per block ------------------------------ */
typedef enum prc_raw_dtype {
    PRC_RAW_I8 = 0, PRC_RAW_U8 = 1, PRC_RAW_I16 = 2, PRC_RAW_F32 = 3, /* interleaved I,Q scalars */
    PRC_RAW_C64 = 4                                                   /* already complex64        */
} prc_raw_dtype;

typedef struct prc_frontend_desc {
    uint32_t struct_size;  /* sizeof(prc_frontend_desc) as the host compiled it (see Conventions)      */
    uint32_t magic;        /* PRC_DESC_MAGIC                                                     */
    int64_t n_in;          /* complex samples per block (= raw scalars / 2)                     */
    int32_t raw_dtype;     /* prc_raw_dtype of the input                                         */
    int32_t up, down;      /* rational resampling factor, already reduced by their gcd           */
    int32_t ntaps;         /* length of taps_host                                                */
    int32_t n_pre_remove;  /* resample_poly's output alignment: (half_len + n_pre_pad) / down    */
    int32_t max_blocks;
    const float* taps_host;/* HOST: firwin(20 max+1, 1/max, ('kaiser',5.0)) * up with n_pre_pad
                              leading zeros (scipy.signal.resample_poly's h), copied at creation  */
} prc_frontend_desc;
#define PRC_FRONTEND_DESC_SIZE_600 48u

typedef struct prc_frontend_plan prc_frontend_plan;
int prc_frontend_plan_create(prc_frontend_plan** plan, const prc_frontend_desc* desc);
int prc_frontend_plan_destroy(prc_frontend_plan* plan);
int prc_frontend_out_len(const prc_frontend_plan* plan, int64_t* n_out);   /* ceil(n_in*up/down) */
/* deinterleave_IQ (signal_utils.py:19-22) -> frequency_shift(fc, fs, block phase) (:24-27 with the
 * array phase of main.py:125-149; mix = 0 skips it) -> resample(up, down) (:15-17, padtype 'line'),
 * fused; block b reads raw at b*raw_stride (elements of the raw type) and writes complex64 at
 * b*out_stride.  phases_host: HOST array of nblocks block phase offsets (radians) or NULL; read while the call is being
 * made (the values travel in the kernel arguments), it may be freed or reused as soon as the call returns. */
int prc_frontend_execute(prc_frontend_plan* plan, const void* raw, int64_t raw_stride, int32_t mix,
                         double fc, double fs, const double* phases_host, void* out, int64_t out_stride,
                         int32_t nblocks, void* stream);
/* The same for BOTH channels of every block in one launch (main.py:133-149 tunes the reference and the surveillance
 * recording with the same frequency and the same block phases): raw_a / raw_b and out_a / out_b share the strides and
 * the phases; one rotation factor per input sample serves the two channels.  Results are bit-identical to two
 * prc_frontend_execute calls.  Ratios the group kernel does not carry run the channels one after the other. */
int prc_frontend_execute2(prc_frontend_plan* plan, const void* raw_a, const void* raw_b, int64_t raw_stride,
                          int32_t mix, double fc, double fs, const double* phases_host, void* out_a, void* out_b,
                          int64_t out_stride, int32_t nblocks, void* stream);
/* deinterleave_IQ alone: n_complex pairs of raw scalars -> complex64 */
int prc_deinterleave(const void* raw, int32_t raw_dtype, int64_t n_complex, void* out, void* stream);
/* frequency_shift with an array (per-block) phase offset: complex64 in, complex128 out, float32 ramp
 * + double block phase, as NumPy promotes it (signal_utils.py:24-27, main.py:133-149) */
int prc_frequency_shift_block(const void* x, void* y, int64_t n, double fc, double fs,
                              double block_phase, void* stream);
/* frequency_shift with one phase PER SAMPLE (signal_utils.py:24-27 broadcasts any array): phases = DEVICE double[n]
 * (phases_f32 = 0: float32 ramp + double phase, complex128 out, NumPy's promotion for float64 / integer arrays) or
 * DEVICE float[n] (phases_f32 = 1: everything in float32, complex64 out, its promotion for float32 arrays) */
int prc_frequency_shift_phases(const void* x, void* y, int64_t n, double fc, double fs,
                               const void* phases, int32_t phases_f32, void* stream);

/* ---- CFAR_2D (SURVEY 8f "next" #3): target_detection.py:683-703 ------------------------------ */
/* X: float32 [nframes][H][W] (|xambg|, H = Doppler rows, W = range columns); out float32 same shape:
 * the CFAR ratio, or 0/1 where ratio > thresh when use_thresh != 0.  Wrap-around box filter of width fw
 * with the (gw+1)^2 guard hole and the 1/(fw^2-gw^2) gain of the reference, normalised by mean|X|. */
int prc_cfar2d(const float* X, int32_t H, int32_t W, int32_t fw, int32_t gw, int32_t use_thresh,
               float thresh, float* out, int32_t nframes, void* stream);
/* CFAR_2D(np.abs(X), fw, gw[, thresh]) as range_doppler_plot.py:56-57 calls it: X = the complex64 range-Doppler maps
 * [nframes][H][W] themselves, |X| (hypotf, as np.abs) taken while the tiles are loaded -- one read of the complex map */
int prc_cfar2d_c64(const void* X, int32_t H, int32_t W, int32_t fw, int32_t gw, int32_t use_thresh,
                   float thresh, float* out, int32_t nframes, void* stream);

/* ---- channel offset estimation (SURVEY 8f "next" #2): signal_utils.py:73-78 ------------------- */
/* The zero-phase IIR decimator of scipy.signal.decimate(x, q) (ftype 'iir', zero_phase=True:
 * sosfiltfilt of a Chebyshev-I low-pass, odd extension of `padlen` samples each side, steady-state
 * initial conditions) as two spectral passes H then conj(H) on device.  The host designs the filter
 * (zeros/poles/gain of the digital low-pass) and says how many samples its impulse response needs to
 * settle (`settle`, >= log(1e-9)/log(max|pole|)). */
typedef struct prc_iir_desc {
    uint32_t struct_size;  /* sizeof(prc_iir_desc) as the host compiled it (see Conventions)           */
    uint32_t magic;        /* PRC_DESC_MAGIC                                                      */
    int32_t q;             /* keep every q-th filtered sample                                    */
    int32_t padlen;        /* sosfiltfilt's odd-extension length, 3*(2*nsections+1 - ...) = 27    */
    int32_t settle;        /* constant-extension length standing in for the steady-state zi      */
    int32_t nzeros, npoles;/* <= 16 each                                                          */
    const double* zeros_host;  /* HOST (re, im) pairs                                             */
    const double* poles_host;  /* HOST (re, im) pairs                                             */
    double gain;
} prc_iir_desc;
#define PRC_IIR_DESC_SIZE_600 56u
/* y[j] = filtfilt(x)[j*q], j < ceil(n/q); x, y complex64 DEVICE.  PRC_ESHAPE if n <= padlen. */
int prc_decimate_iir(const void* x, int64_t n, const prc_iir_desc* iir, void* y, void* stream);
/* find_channel_offset(s1, s2, nd, nl), signal_utils.py:73-78: B1 = decimate(s1, nd), B2 = decimate(s2, nd)
 * zero-padded by nl each side, xc = |correlate(B1, B2, 'valid')| (m2 + 2 nl - m1 + 1 values,
 * xc[i] = |sum_l B1[l] conj(B2[l + m2 - m1 + nl - i])|), via rocFFT.  xc_out: DEVICE float32[n_xc] or
 * NULL; *argmax_out (HOST) = first index of the maximum, so the reference's return value is
 * (*argmax_out - nl) * nd.  Synchronises `stream` before returning. */
int prc_channel_offset(const void* s1, int64_t n1, const void* s2, int64_t n2, const prc_iir_desc* iir,
                       int64_t nl, float* xc_out, int64_t* n_xc, int64_t* argmax_out, void* stream);

/* ---- frame gather over xGMI: main.py:213-224 (da.store / to_zarr) for sharded frames ---- */
/* CPI frames shard contiguously over the GPUs of a node (one process per GPU); the only exchange of
 * the path is the gather of every rank's [frames][freq_bins][range_bins+1] complex64 block to one
 * root, as a group of RCCL point-to-point transfers (ncclSend on the peers, one ncclRecv per peer on
 * the root), each on its own xGMI link.  RCCL is bound at run time (dlopen librccl.so.1); without it
 * these entry points return PRC_EUNSUPPORTED and everything else still works.
 *   rank 0:      prc_comm_unique_id(id)  -> ship the 128 bytes to every rank (any side channel)
 *   every rank:  prc_set_device(local_gpu); prc_comm_create(&comm, id, rank, world)   (collective)
 *   every pass:  prc_gather_frames(comm, my_frames, frames_per_rank, F*(R+1), all_frames, 0, stream) */
#define PRC_COMM_ID_BYTES 128
typedef struct prc_comm prc_comm;
int prc_comm_unique_id(void* id_host);                       /* HOST buffer of PRC_COMM_ID_BYTES      */
int prc_comm_create(prc_comm** comm, const void* id_host, int32_t rank, int32_t world);
int prc_comm_rccl_version(int32_t* version);                  /* ncclGetVersion of the RCCL that was bound      */
/* what RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank): the number of ranks it connected and
 * this process's rank among them -- a bench line can show that N ranks really met */
int prc_comm_count(const prc_comm* comm, int32_t* nranks, int32_t* rank);
/* Link check of the point-to-point path on THIS rank alone: nfloats float32 from `send` to `recv` (both DEVICE, not
 * overlapping) as one grouped ncclSend + ncclRecv with this rank as its own peer -- the same RCCL calls, datatype and
 * stream discipline as prc_gather_frames, runnable on a communicator of any size including one (a one-GPU box), so a host
 * can check the RCCL binding before the first gather.  Enqueued on `stream`; not a collective. */
int prc_comm_loopback(prc_comm* comm, const void* send, void* recv, int64_t nfloats, void* stream);
int prc_comm_destroy(prc_comm* comm);
/* send: this rank's frames_per_rank_host[rank] frames of frame_elems complex64 (DEVICE).  recv (root
 * only, DEVICE): sum(frames_per_rank_host) frames, rank r's block at frame offset sum_{q<r}.  Blocks may
 * be ragged or empty.  Enqueued on `stream`; nothing synchronises.  The root's own block is a device
 * copy (skipped when send already points at its slot in recv). */
int prc_gather_frames(prc_comm* comm, const void* send, const int64_t* frames_per_rank_host,
                      int64_t frame_elems, void* recv, int32_t root, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PRCORE_H */
