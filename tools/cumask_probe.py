"""Does the config-2 step gain from running the (VALU-bound) CAF on its own CUs beside the (HBM-bound) LS passes?
hipExtStreamCreateWithCUMask streams in place of HipBackend's LS / CAF streams; frames/s of be.run over resident chunks."""
import ctypes
import sys
import time
import torch
sys.path.insert(0, ".")
from passiveradar_amd import scene
from passiveradar_amd.stream import HipBackend

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(((bits >> (32 * w + b)) & 1) << b for b in range(32)) for w in range(8)])
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value, device=torch.device("cuda", 0))


def caf_set(n):
    """n in (0, 32, 48, 64): CUs for the CAF stream, uniform over i % 8 and over i // 32 (whichever of the two names the XCD)"""
    bits = 0
    for i in range(256):
        a, b, c = i // 32, (i // 8) % 4, i % 8
        if b != 0:
            continue
        if n == 64 or (n == 32 and (a + c) % 2 == 0) or (n == 48 and (a + c) % 4 != 0) or (n == 16 and (a + c) % 4 == 0):
            bits |= 1 << i
    return bits


Fs, N, R, F = 2.4e6, 2_400_000, 256, 512
C = N // 2
nchunks = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
be = HipBackend(N, R, F, Fs, batch=256, device=torch.device("cuda", 0))
g = torch.Generator(device="cuda").manual_seed(1)
ref = torch.view_as_complex(torch.randn((nchunks * C, 2), device="cuda", generator=g))
srv = 0.5 * ref.roll(7) + 0.1 * torch.view_as_complex(torch.randn((nchunks * C, 2), device="cuda", generator=g))
ref_pad, srv_pad = be.padded(ref), be.padded(srv)
del ref, srv
out = torch.empty((nchunks, F, R + 1), dtype=torch.complex64, device="cuda")
ALL = (1 << 256) - 1
base = (be.s_ls_all, be.s_caf)
for n in (0, 16, 32, 48, 64, -1):
    if n == 0:
        be.s_ls_all, be.s_caf = base
        tag = "torch streams (no mask)"
    elif n == -1:
        be.s_ls_all = [masked_stream(ALL) for _ in base[0]]
        be.s_caf = masked_stream(ALL)
        tag = "masked streams, all 256 CUs each"
    else:
        m = caf_set(n)
        be.s_ls_all = [masked_stream(ALL & ~m) for _ in base[0]]
        be.s_caf = masked_stream(m)
        tag = f"CAF on {n} CUs, LS on {256 - n}"
    for _ in range(2):
        be.run(ref_pad, srv_pad, nchunks, 0, nchunks, out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    reps = 3
    for _ in range(reps):
        be.run(ref_pad, srv_pad, nchunks, 0, nchunks, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    print(f"{tag:40s} {nchunks / dt:9.0f} frames/s  ({dt * 1e3 / (nchunks / 256):.2f} ms per 256 frames)  checksum {out.abs().sum().item():.6e}", flush=True)
