import sys, torch, numpy as np, os
sys.path.insert(0, ".")
from passiveradar_amd import engine, _lib
dev = torch.device("cuda")
C, nb, L = 1200000, 256, 256
g = torch.Generator(device=dev); g.manual_seed(1)
ref = torch.view_as_complex(torch.randn((nb * C, 2), generator=g, device=dev))
srv = torch.roll(ref, 2) + 0.01 * torch.view_as_complex(torch.randn((nb * C, 2), generator=g, device=dev))
s = _lib.torch_stream_ptr()
out = torch.empty_like(srv)
plan = engine.LsPlan(C, L, 10, False, nb, 0)
plan.set_profiling(True)
acc = []
for rep in range(6):
    plan.execute(ref, srv, out, nb, C, C, 2.4e6, (0, 1, -1, 2, -2), 0.0, None, s)
    ms, k = plan.get_profile()
    if rep: acc.append(ms)
acc = np.median(np.array(acc), axis=0)
print(os.environ.get("PRCORE_LIB", "shipped"), "corr ms/launch %.4f  solve %.4f  fused %.4f (x%d)" % (acc[0] / k[0], acc[1] / k[1], acc[2] / k[2], k[2]))
