"""Can the 256 MiB Infinity Cache carry the LS chain's inter-bin stream and spectrum cache?  

Part 1 -- what the memory system gives a plain streaming pass as a function of its working set: a copy (read + write)
and a read-only reduction over W MB, repeated back to back, W from 16 MB to 4 GB.  Below the cache size the second and
later passes can be served on-die; the rate over W is the on-die rate the chain could hope for.

Part 2 -- the five-bin chain itself (LS_Filter_Multiple, clutter_removal.py:162-187; T = 266, 1.2 M-sample chunks, the
4096-point cached-spectrum kernels) at 4 ... 256 chunks per plan, the five bins back to back on one stream: microseconds
per chunk-bin of the first-bin kernel, the fused kernel and the solves, from the plan's own events.  A chunk's working
set in the chain is ~49 MB (reference 9.6, surveillance 9.6, two inter-bin streams 19.2, spectrum cache 10.3); if the
cache carried it, the fused kernel at 4-5 chunks (196-245 MB) would run well under the 5.8 us per chunk-bin it takes at
256 chunks (12.5 GB).

    python tools/mall_probe.py [--pmc]      (--pmc: one six-chunk chain only, for a rocprofv3 --pmc pass)
"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from passiveradar_amd import _lib, engine  # noqa: E402

dev = torch.device("cuda")


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def chain(nb, reps=6, T=266):
    C = 1200000
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    ref = torch.view_as_complex(torch.randn((nb * C, 2), generator=g, device=dev))
    srv = torch.roll(ref, 2) + 0.01 * torch.view_as_complex(torch.randn((nb * C, 2), generator=g, device=dev))
    out = torch.empty_like(srv)
    plan = engine.LsPlan(C, T - 10, 10, False, nb, 4)
    plan.set_profiling(True)
    s = _lib.torch_stream_ptr()
    acc, wall = [], []
    for rep in range(reps):
        a, b = torch.cuda.Event(True), torch.cuda.Event(True)
        a.record()
        plan.execute(ref, srv, out, nb, C, C, 2.4e6, (0, 1, -1, 2, -2), 0.0, None, s)
        b.record()
        torch.cuda.synchronize()
        ms, k = plan.get_profile()
        if rep:
            acc.append(ms)
            wall.append(a.elapsed_time(b))
    acc = np.median(np.array(acc), axis=0)
    plan.close()
    return acc[0] / k[0], acc[1] / k[1], acc[2] / k[2], float(np.median(wall))


if "--pmc" in sys.argv:
    chain(6, reps=3)
    sys.exit(0)

print("## Part 1: streaming rate against working set (torch copy_ = read + write, sum = read only)\n")
print("| working set MB | copy TB/s | read TB/s |\n|---|---|---|")
for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 4096):
    n = mb * (1 << 20) // 8                      # copy: half the set is the source, half the destination
    x = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    y = torch.empty_like(x)
    reps = max(20, min(2000, (8 << 30) // (mb << 20)))
    dt = timed(lambda: y.copy_(x), reps)
    z = torch.empty(2 * n, dtype=torch.float32, device=dev).normal_()
    dr = timed(lambda: z.sum(), reps)
    print(f"| {mb} | {2 * n * 4 / dt / 1e12:.2f} | {2 * n * 4 / dr / 1e12:.2f} |", flush=True)
    del x, y, z

print("\n## Part 2: the five-bin chain (T = 266, 4096-point cached kernels), bins back to back on one stream\n")
print("| chunks per plan | chain working set MB | first-bin us/chunk | fused us/chunk-bin | solve kernels us/launch | whole chain ms | chain us/chunk |")
print("|---|---|---|---|---|---|---|")
for nb in (4, 5, 6, 8, 12, 16, 32, 64, 256):
    c, sv, f, wall = chain(nb)
    print(f"| {nb} | {nb * 49} | {c * 1e3 / nb:.2f} | {f * 1e3 / nb:.2f} | {sv * 1e3:.1f} | {wall:.3f} | {wall * 1e3 / nb:.1f} |", flush=True)
