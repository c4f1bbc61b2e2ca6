"""find_channel_offset at the size main.py:54/:83 calls it (nd=1, nl=5e6 on 10 CPIs of 524288 samples):
device time of the drop-in against the reference's three SciPy calls on one host core."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from passiveradar_amd.signal_utils import find_channel_offset

n, nl, shift = 10 * 524288, 5000000, 123457
rng = np.random.Generator(np.random.Philox(key=5))
base = (rng.standard_normal(2 * (n + 2 * shift), dtype=np.float32) * np.float32(0.7)).view(np.complex64)
s1 = base[shift:shift + n].copy()
s2 = (base[:n] + 0.5 * (rng.standard_normal(2 * n, dtype=np.float32)).view(np.complex64)).astype(np.complex64)  # s2[n] = s1[n - shift]
find_channel_offset(s1[:100000], s2[:100000], 1, 1000)
for rep in range(3):
    t = time.perf_counter()
    off = find_channel_offset(s1, s2, 1, nl)
    dt = time.perf_counter() - t
    print(f"GPU drop-in (host arrays in, PCIe included): offset {off} in {dt * 1e3:.1f} ms")
if "--cpu" in sys.argv:
    import scipy.signal as signal
    t = time.perf_counter()
    B1 = signal.decimate(s1, 1)
    B2 = np.pad(signal.decimate(s2, 1), (nl, nl), "constant")
    xc = np.abs(signal.correlate(B1, B2, mode="valid"))
    offc = (np.argmax(xc) - nl) * 1
    print(f"SciPy calls of the reference, one core     : offset {offc} in {time.perf_counter() - t:.2f} s")
    assert offc == off
