"""Does the alignment of the LS pieces matter?  Time the five-bin chain for filter lengths whose history T-1 is / is not a
multiple of 16 samples (128 bytes): T-1 = 256 -> pieces of 768 samples, every access line-aligned; T-1 = 265 (config 2) ->
pieces of 759 samples, surveillance reads and cleaned-stream writes start 56 bytes into a line."""
import sys, torch, numpy as np
sys.path.insert(0, ".")
from passiveradar_amd import engine, _lib
dev = torch.device("cuda")
C, nb = 1200000, 256
g = torch.Generator(device=dev); g.manual_seed(1)
ref = torch.view_as_complex(torch.randn((nb * C, 2), generator=g, device=dev))
srv = torch.roll(ref, 2) + 0.01 * torch.view_as_complex(torch.randn((nb * C, 2), generator=g, device=dev))
out = torch.empty_like(srv)
s = _lib.torch_stream_ptr()
for L in (247, 256, 263):
    plan = engine.LsPlan(C, L, 10, False, nb, 0)
    plan.set_profiling(True)
    acc = np.zeros(3)
    for rep in range(4):
        plan.execute(ref, srv, out, nb, C, C, 2.4e6, (0, 1, -1, 2, -2), 0.0, None, s)
        ms, k = plan.get_profile()
        if rep: acc += ms
    acc /= 3
    T = L + 10
    print(f"L={L} T-1={T-1} piece={1024-(T-1)}: corr {acc[0]/k[0]:.3f} ms  solve {acc[1]/k[1]:.3f}  fused {acc[2]/k[2]:.3f} ms per launch of {nb} chunks", flush=True)
    plan.close()
