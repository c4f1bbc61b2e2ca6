// TEST ONLY (CPU): runs the phases of the Doppler column-FFT kernel (passiveradar_amd/csrc/doppler_col.h) thread by
// thread on the host, with the kernel's own index algebra, twiddle table and LDS slots -- so the build container,
// which has no GPU, can check them against numpy.fft (tests/test_host_logic.py).  A barrier is the end of a loop
// over the workgroup's threads.
#include <hip/hip_runtime.h>
#include <vector>
#include "../../passiveradar_amd/csrc/doppler_col.h"

void dop_make_table_host(float2* t, int F) {
    const double PI = 3.14159265358979323846;
    for (int m = 0; m < F; ++m) {
        const double a = -2.0 * PI * (double)m / (double)F;
        t[m] = make_float2((float)cos(a), (float)sin(a));
    }
}

template <int F>
static int emul(const float2* y, float2* out, int cols, int nframes) {
    constexpr int Q = DopCfg<F>::Q, KT = DopCfg<F>::KT, F3 = DopCfg<F>::F3, NT = DopCfg<F>::THREADS;
    std::vector<float2> tw(F), lds(DopCfg<F>::LDS_ELEMS);
    dop_make_table_host(tw.data(), F);
    std::vector<float2> regs((size_t)NT * 16);
    for (int fr = 0; fr < nframes; ++fr)
        for (int k0 = 0; k0 < cols; k0 += KT) {
            for (auto& v : lds) v = make_float2(1e30f, 1e30f);          // a read of an unwritten slot shows
            auto X = [&](int t) -> float2(&)[16] { return *reinterpret_cast<float2(*)[16]>(&regs[(size_t)t * 16]); };
            if (!DopCfg<F>::SPLIT) {
                for (int t = 0; t < NT; ++t) {
                    const int c = t % KT, p = t / KT, k = k0 + c;
                    float2(&x)[16] = X(t);
                    for (int r = 0; r < 16; ++r)
                        x[r] = k < cols ? y[((int64_t)fr * F + r * Q + p) * cols + k] : make_float2(0.f, 0.f);
                    dop_stage1<F>(x, dop_load_twiddles<F>(tw.data(), p));
                    dop_write1<F>(x, lds.data(), p, c);
                }
                for (int t = 0; t < NT; ++t) {                               // after the first barrier
                    const int c = t % KT, p = t / KT;
                    float2(&x)[16] = X(t);
                    dop_read1<F>(x, lds.data(), p, c);
                    dop_stage2<F>(x, dop_load_twiddles<F>(tw.data(), p));
                    if (F3 > 1) dop_write2<F>(x, lds.data(), p, c);           // in place: only slots this thread read
                }
                for (int t = 0; t < NT; ++t) {                               // after the second barrier
                    const int c = t % KT, p = t / KT;
                    float2(&x)[16] = X(t);
                    if (F3 > 1) {
                        dop_read2<F>(x, lds.data(), p, c);
                        dop_stage3<F>(x);
                    }
                }
            } else {
                // two-round form: one loop over the threads per barrier-separated phase, float slots
                float* ldf = reinterpret_cast<float*>(lds.data());
                std::vector<float2> zreg((size_t)NT * 16);
                auto Z = [&](int t) -> float2(&)[16] { return *reinterpret_cast<float2(*)[16]>(&zreg[(size_t)t * 16]); };
                for (int t = 0; t < NT; ++t) {
                    const int c = t % KT, p = t / KT, k = k0 + c;
                    float2(&x)[16] = X(t);
                    for (int r = 0; r < 16; ++r)
                        x[r] = k < cols ? y[((int64_t)fr * F + r * Q + p) * cols + k] : make_float2(0.f, 0.f);
                    dop_stage1<F>(x, dop_load_twiddles<F>(tw.data(), p));
                    dop_write1c<F, 0>(x, ldf, p, c);
                }
                for (int t = 0; t < NT; ++t) dop_read1c<F, 0>(Z(t), ldf, t / KT, t % KT);
                for (int t = 0; t < NT; ++t) dop_write1c<F, 1>(X(t), ldf, t / KT, t % KT);
                for (int t = 0; t < NT; ++t) {
                    const int c = t % KT, p = t / KT;
                    dop_read1c<F, 1>(Z(t), ldf, p, c);
                    float2(&x)[16] = X(t);
                    for (int r = 0; r < 16; ++r) x[r] = Z(t)[r];
                    dop_stage2<F>(x, dop_load_twiddles<F>(tw.data(), p));
                    dop_write2c<F, 0>(x, ldf, p, c);                          // no barrier: this thread's own slots
                }
                for (int t = 0; t < NT; ++t) dop_read2c<F, 0>(Z(t), ldf, t / KT, t % KT);
                for (int t = 0; t < NT; ++t) dop_write2c<F, 1>(X(t), ldf, t / KT, t % KT);
                for (int t = 0; t < NT; ++t) {
                    dop_read2c<F, 1>(Z(t), ldf, t / KT, t % KT);
                    float2(&x)[16] = X(t);
                    for (int r = 0; r < 16; ++r) x[r] = Z(t)[r];
                    dop_stage3<F>(x);
                }
            }
            for (int t = 0; t < NT; ++t) {
                const int c = t % KT, p = t / KT, k = k0 + c;
                float2(&x)[16] = X(t);
                if (k < cols)
                    for (int m = 0; m < 16; ++m) {
                        // the kernel splits the row into a per-thread and a per-register part (no carry between them)
                        const int row = (p / F3) + 16 * (p % F3) * DopCfg<F>::E + dop_out_row_reg<F>(m);
                        if (row != dop_out_row<F>(p, m)) return -2;
                        out[((int64_t)fr * F + row) * cols + k] = x[m];
                    }
            }
        }
    return 0;
}

// The in-place claim of X2 (a thread overwrites exactly the slots it read in X1) is what lets the kernel skip a
// barrier; checked here by running write2 of every thread BEFORE read1 of a later one would have happened: the loop
// above interleaves read1/write2 per thread, which is only correct if the claim holds.
extern "C" int dop_emul(int F, const void* y, void* out, int cols, int nframes) {
    const float2* yy = (const float2*)y;
    float2* oo = (float2*)out;
    switch (F) {
        case 256: return emul<256>(yy, oo, cols, nframes);
        case 512: return emul<512>(yy, oo, cols, nframes);
        case 1024: return emul<1024>(yy, oo, cols, nframes);
        case 2048: return emul<2048>(yy, oo, cols, nframes);
        case 4096: return emul<4096>(yy, oo, cols, nframes);
    }
    return -1;
}
