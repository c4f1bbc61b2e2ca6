"""Front-end throughput at the shipped PRconfig.yaml block size (4 799 250 raw int8 scalars -> 262 144 IF samples)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from passiveradar_amd.stream import HipBackend

be = HipBackend(524288, 175, 1024, 262184.87, batch=4, clutter=None)
icl, nblk = 4799250, 32
raw = torch.randint(-100, 100, (icl * nblk,), dtype=torch.int8, device="cuda")
for _ in range(2):
    out = be.front_end(raw, icl, 100000, 2400000, 13, 119)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    out = be.front_end(raw, icl, 100000, 2400000, 13, 119)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
print(f"front end: {nblk} blocks in {dt*1e3:.2f} ms -> {dt/nblk*1e6:.1f} us per block-channel "
      f"({icl*nblk/dt/1e9:.1f} G raw scalars/s, {icl*nblk*1/dt/1e9:.1f} GB/s raw in); out {out.shape}")
