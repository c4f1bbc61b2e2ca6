"""PCIe-inclusive rate at config 2: the IF stream handed over as host arrays, uploaded in
batches of hop chunks from pinned memory on a copy stream while the previous batch computes, maps downloaded."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from passiveradar_amd import scene, stream as prstream

n, R, F, fs = 2400000, 256, 512, 2.4e6
C, B, NB = n // 2, 64, 6
dev = torch.device("cuda", 0)
be = prstream.HipBackend(n, R, F, fs, batch=B, device=dev)
r0, s0 = scene.make_stream(2, C, fs, R, 1)
host_ref = torch.from_numpy(np.tile(r0, B // 2)).pin_memory()
host_srv = torch.from_numpy(np.tile(s0, B // 2)).pin_memory()
host_out = torch.empty((B, F, R + 1), dtype=torch.complex64).pin_memory()
# raw copy bandwidth
d = torch.empty_like(host_ref, device=dev)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): d.copy_(host_ref, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
print(f"pinned H2D: {host_ref.numel() * 8 / dt / 1e9:.1f} GB/s")
copy = torch.cuda.Stream(dev)
bufs = [(torch.empty_like(d), torch.empty_like(d)) for _ in range(2)]
evs = [None, None]
def upload(i):
    with torch.cuda.stream(copy):
        bufs[i % 2][0].copy_(host_ref, non_blocking=True)
        bufs[i % 2][1].copy_(host_srv, non_blocking=True)
        e = torch.cuda.Event(); e.record(copy); evs[i % 2] = e
def run(nb):
    upload(0)
    for i in range(nb):
        torch.cuda.current_stream().wait_event(evs[i % 2])
        if i + 1 < nb:
            copy.wait_stream(torch.cuda.current_stream()) if i >= 1 else None   # buffer (i+1)%2 free once batch i-1 is done
            upload(i + 1)
        rp, sp = be.padded(bufs[i % 2][0]), be.padded(bufs[i % 2][1])
        fr = be.run(rp, sp, B, 0, B)
        host_out.copy_(fr, non_blocking=True)
    torch.cuda.synchronize()
run(2)
t = time.perf_counter(); run(NB); dt = time.perf_counter() - t
print(f"host arrays -> maps on host, upload overlapped: {NB * B / dt:.0f} frames/s "
      f"({NB * B * 2 * C * 8 / dt / 1e9:.1f} GB/s of IQ in, {NB * B * F * (R + 1) * 8 / dt / 1e9:.2f} GB/s of maps out)")
rp, sp = be.padded(bufs[0][0]), be.padded(bufs[0][1])
be.run(rp, sp, B, 0, B); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(NB): be.run(rp, sp, B, 0, B)
torch.cuda.synchronize(); dt = time.perf_counter() - t
print(f"same batches resident in HBM: {NB * B / dt:.0f} frames/s")
