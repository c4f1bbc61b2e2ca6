/* A plain C host of libprcore.so (no Python, no torch): the reference-side binding of INTEGRATION.md section B
 * written out.  Builds with    gcc -std=c99 -I include examples/c_host.c -L passiveradar_amd -lprcore -lm
 * Prints the peak cell and a checksum of one cross-ambiguity surface of a synthetic echo (delay 7 samples,
 * Doppler +5 cycles per CPI), which tests/test_gpu_parity.py compares with the Python drop-in. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "prcore.h"

#define CHK(call)                                                                  \
    do {                                                                           \
        int rc_ = (call);                                                          \
        if (rc_ != PRC_OK) {                                                       \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, prc_last_error());       \
            return 1;                                                              \
        }                                                                          \
    } while (0)

int main(void) {
    const long n = 8192;
    const int R = 20, F = 32;
    float* ref = (float*)malloc(sizeof(float) * 2 * n);
    float* srv = (float*)malloc(sizeof(float) * 2 * n);
    float* out = (float*)malloc(sizeof(float) * 2 * F * (R + 1));
    unsigned s = 12345u;
    for (long i = 0; i < 2 * n; ++i) {               /* LCG, uniform in [-1, 1) */
        s = s * 1664525u + 1013904223u;
        ref[i] = (float)((double)(s >> 8) / 8388608.0 - 1.0);
    }
    for (long i = 0; i < n; ++i) {                   /* srv[i] = ref[i-7] e^{+j 2 pi 5 i / n} (circular) */
        const long k = (i - 7 + n) % n;
        const double ph = 2.0 * 3.14159265358979323846 * 5.0 * (double)i / (double)n;
        const double c = cos(ph), sn = sin(ph);
        srv[2 * i] = (float)(ref[2 * k] * c - ref[2 * k + 1] * sn);
        srv[2 * i + 1] = (float)(ref[2 * k] * sn + ref[2 * k + 1] * c);
    }
    int ndev = 0;
    CHK(prc_device_count(&ndev));
    if (ndev < 1) { fprintf(stderr, "no ROCm device\n"); return 2; }
    prc_caf_desc d;
    PRC_DESC_INIT(d);                                /* zeroes, then struct_size = sizeof(d), magic */
    d.n = n; d.range_bins = R; d.freq_bins = F; d.max_frames = 1; d.method = 0; d.doppler = 0; d.ntaps = 0; d.taps_host = NULL;
    d.multi = PRC_CAF_MULTI_AUTO;
    prc_caf_plan* plan = NULL;
    CHK(prc_caf_plan_create(&plan, &d));
    void *dref = NULL, *dsrv = NULL, *dout = NULL;
    CHK(prc_malloc(&dref, sizeof(float) * 2 * n));
    CHK(prc_malloc(&dsrv, sizeof(float) * 2 * n));
    CHK(prc_malloc(&dout, sizeof(float) * 2 * F * (R + 1)));
    CHK(prc_memcpy_h2d(dref, ref, sizeof(float) * 2 * n, NULL));
    CHK(prc_memcpy_h2d(dsrv, srv, sizeof(float) * 2 * n, NULL));
    CHK(prc_caf_execute(plan, dref, dsrv, n, n, NULL, dout, 1, NULL));
    CHK(prc_memcpy_d2h(out, dout, sizeof(float) * 2 * F * (R + 1), NULL));
    CHK(prc_stream_sync(NULL));
    double sum = 0.0, best = -1.0;
    int br = -1, bc = -1;
    for (int r = 0; r < F; ++r)
        for (int c = 0; c <= R; ++c) {
            const double re = out[2 * (r * (R + 1) + c)], im = out[2 * (r * (R + 1) + c) + 1];
            const double p = re * re + im * im;
            sum += p;
            if (p > best) { best = p; br = r; bc = c; }
        }
    printf("peak row %d col %d power %.6e checksum %.6e\n", br, bc, best, sum);
    /* The one collective of the path from C: gather of the per-rank map blocks to rank 0 (here a world of one; with
     * several processes rank 0 ships `id` to the others, every rank calls prc_comm_create, and frames_per_rank
     * lists every rank's block).  Skipped when RCCL is not installed (PRC_EUNSUPPORTED). */
    {
        unsigned char id[PRC_COMM_ID_BYTES];
        int rc = prc_comm_unique_id(id);
        if (rc == PRC_OK) {
            prc_comm* comm = NULL;
            void* dall = NULL;
            const int64_t frames_per_rank[1] = {1};
            CHK(prc_comm_create(&comm, id, 0, 1));
            CHK(prc_malloc(&dall, sizeof(float) * 2 * F * (R + 1)));
            CHK(prc_gather_frames(comm, dout, frames_per_rank, (int64_t)F * (R + 1), dall, 0, NULL));
            CHK(prc_memcpy_d2h(ref, dall, sizeof(float) * 2 * F * (R + 1), NULL));   /* ref is free to reuse (2n floats) */
            CHK(prc_stream_sync(NULL));
            double diff = 0.0;
            for (int i = 0; i < 2 * F * (R + 1); ++i) diff += fabs((double)ref[i] - (double)out[i]);
            printf("gathered 1 frame through prc_gather_frames, difference %.1e\n", diff);
            CHK(prc_comm_destroy(comm));
            prc_free(dall);
        } else if (rc == PRC_EUNSUPPORTED) {
            printf("gather skipped: %s\n", prc_last_error());
        } else {
            fprintf(stderr, "prc_comm_unique_id -> %d: %s\n", rc, prc_last_error());
            return 1;
        }
    }
    CHK(prc_caf_plan_destroy(plan));
    prc_free(dref); prc_free(dsrv); prc_free(dout);
    free(ref); free(srv); free(out);
    return 0;
}
