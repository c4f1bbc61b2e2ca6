"""Front-end throughput at the shipped PRconfig.yaml block size (4 799 250 raw int8 scalars -> 262 144 IF samples), both
kernel forms (PRC_OPT_FE_METHOD 1 = one output per thread, 2 = `up` outputs per thread), HIP events around the launches."""
import sys
import torch
sys.path.insert(0, ".")
from passiveradar_amd import _lib
from passiveradar_amd.stream import HipBackend

be = HipBackend(524288, 175, 1024, 262184.87, batch=4, clutter=None)
icl, nblk = 4799250, 32
raw = torch.randint(-100, 100, (icl * nblk,), dtype=torch.int8, device="cuda")
outs = {}
methods = tuple(int(m) for m in sys.argv[1:]) or (1, 2)
for method in methods:
    _lib.set_option(_lib.OPT_FE_METHOD, method)
    for _ in range(2):
        out = be.front_end(raw, icl, 100000, 2400000, 13, 119)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        out = be.front_end(raw, icl, 100000, 2400000, 13, 119)
    e1.record()
    torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) * 1e-3 / 5
    outs[method] = out.clone()
    nout = out.shape[0] // nblk
    gflop = nblk * (nout * 4.0 * 2381 / 13 + (icl // 2) * 6.0) / 1e9   # real tap x complex sample multiply-adds on non-zero taps + one rotation product per input
    print(f"front end method {method}: {nblk} blocks in {dt*1e3:.3f} ms -> {dt/nblk*1e6:.2f} us per block-channel "
          f"({icl*nblk/dt/1e9:.1f} GB/s raw in, {(icl + 8*nout)*nblk/dt/1e12:.3f} TB/s in+out, {gflop/dt/1e3:.1f} TFLOP/s of FIR + rotation products)")
_lib.set_option(_lib.OPT_FE_METHOD, 0)
if 1 in outs and 2 in outs:
    d = (outs[1] - outs[2]).abs().max().item() / outs[1].abs().max().item()
    print(f"max |method 1 - method 2| / peak = {d:.2e}")
