"""Does splitting one NLMS stream over several wavefronts buy latency?  T = 1034 (config 3) on one, two and four wavefronts
per stream (prc_set_option(PRC_OPT_NLMS_WAVES), honoured for this filter length only): ns per step of a single stream, and the throughput of
3072 concurrent streams.      python tools/nlms_waves_probe.py"""
import sys, time, torch
sys.path.insert(0, ".")
from passiveradar_amd import engine, _lib
dev = torch.device("cuda")
L = 1024
s = _lib.torch_stream_ptr()
for ns, n in ((1, 200000), (3072, 30000)):
    g = torch.Generator(device=dev); g.manual_seed(1)
    ref = torch.view_as_complex(torch.randn((ns * n, 2), generator=g, device=dev))
    srv = torch.roll(ref, 2) + 0.01 * torch.view_as_complex(torch.randn((ns * n, 2), generator=g, device=dev))
    outs = {}
    for waves in (1, 2, 4):
        _lib.set_option(_lib.OPT_NLMS_WAVES, waves)
        out = torch.empty_like(srv)
        engine.nlms_execute(ref, srv, out, n, L, 0.02, 10, None, None, ns, n, n, s); torch.cuda.synchronize()
        t = time.perf_counter()
        engine.nlms_execute(ref, srv, out, n, L, 0.02, 10, None, None, ns, n, n, s); torch.cuda.synchronize()
        dt = time.perf_counter() - t
        outs[waves] = out
        d = float((out - outs[1]).abs().max() / outs[1].abs().max())
        print(f"{ns:5d} stream(s), {waves} wavefront(s) per stream: {dt / (n - L - 10) * 1e9:7.0f} ns per step, "
              f"{ns * (n - L - 10) / dt / 1e9:.3f} GS/s   (output differs from the one-wavefront run by {d:.1e})", flush=True)
