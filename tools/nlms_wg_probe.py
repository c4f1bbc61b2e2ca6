"""How many NLMS wavefronts (independent streams) should a SIMD hold?  T = 1034 (config 3), 8, 12 or 16 wavefronts per
workgroup (= 2, 3, 4 per SIMD; prc_set_option(PRC_OPT_NLMS_WG_WAVES)): throughput at 3072 and 4096 concurrent streams.
    python tools/nlms_wg_probe.py"""
import sys, time, torch
sys.path.insert(0, ".")
from passiveradar_amd import engine, _lib
dev = torch.device("cuda")
L, n = 1024, 60000
s = _lib.torch_stream_ptr()
for ns in (3072, 4096):
    g = torch.Generator(device=dev); g.manual_seed(1)
    ref = torch.view_as_complex(torch.randn((ns * n, 2), generator=g, device=dev))
    srv = torch.roll(ref, 2) + 0.01 * torch.view_as_complex(torch.randn((ns * n, 2), generator=g, device=dev))
    base = None
    for wg in (8, 12, 16):
        _lib.set_option(_lib.OPT_NLMS_WG_WAVES, wg)
        out = torch.empty_like(srv)
        engine.nlms_execute(ref, srv, out, n, L, 0.02, 10, None, None, ns, n, n, s); torch.cuda.synchronize()
        t = time.perf_counter()
        engine.nlms_execute(ref, srv, out, n, L, 0.02, 10, None, None, ns, n, n, s); torch.cuda.synchronize()
        dt = time.perf_counter() - t
        base = out if base is None else base
        d = float((out - base).abs().max() / base.abs().max())
        print(f"{ns} streams, {wg} wavefronts per workgroup ({wg // 4} per SIMD): {ns * (n - L - 10) / dt / 1e9:.3f} GS/s, "
              f"{dt / (n - L - 10) * 1e9:.0f} ns per step of the whole batch  (output differs from the first run by {d:.1e})", flush=True)
    _lib.set_option(_lib.OPT_NLMS_WG_WAVES, 0)
