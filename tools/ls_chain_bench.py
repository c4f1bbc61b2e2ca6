"""Five-bin LS chain (LS_Filter_Multiple, clutter_removal.py:162-187) over the filter lengths the cached-spectrum chain
carries, on the 1024-point wavefront kernels (method 3) and on the 4096-point team kernels (method 4): per-launch kernel
times from the plan's own events.  Decides where prc_ls_plan_create's AUTO switches between the two.
    python tools/ls_chain_bench.py [blocks]"""
import sys, torch, numpy as np
sys.path.insert(0, ".")
from passiveradar_amd import engine, _lib
dev = torch.device("cuda")
C, nb = 1200000, int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator(device=dev); g.manual_seed(1)
ref = torch.view_as_complex(torch.randn((nb * C, 2), generator=g, device=dev))
srv = torch.roll(ref, 2) + 0.01 * torch.view_as_complex(torch.randn((nb * C, 2), generator=g, device=dev))
outs = {}
s = _lib.torch_stream_ptr()
print(f"{nb} chunks of {C} samples, bins (0, 1, -1, 2, -2); ms per launch")
for L in (54, 118, 182, 246, 256, 374, 502, 758):
    row = []
    for method in (3, 4):
        out = torch.empty_like(srv)
        plan = engine.LsPlan(C, L, 10, False, nb, method)
        plan.set_profiling(True)
        acc = []
        for rep in range(4):
            plan.execute(ref, srv, out, nb, C, C, 2.4e6, (0, 1, -1, 2, -2), 0.0, None, s)
            ms, k = plan.get_profile()
            if rep: acc.append(ms)
        acc = np.median(np.array(acc), axis=0)
        total = float(acc.sum())
        row.append((acc[0] / k[0], acc[1] / k[1], acc[2] / k[2], total))
        outs[method] = out
        plan.close()
    d = float((outs[3] - outs[4]).abs().max() / outs[3].abs().max())
    a, b = row
    print(f"T={L + 10:4d}  1024-pt: corr {a[0]:.3f} solve {a[1]:.3f} fused {a[2]:.3f} chain {a[3]:.2f} | 4096-pt: corr {b[0]:.3f} "
          f"solve {b[1]:.3f} fused {b[2]:.3f} chain {b[3]:.2f} | 4096/1024 = {b[3] / a[3]:.3f}  (outputs differ by {d:.1e})", flush=True)
