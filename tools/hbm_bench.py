"""What HBM delivers to plain streaming kernels on this box (the practical ceiling the roofline fractions in
DESIGN.md should be read against): torch copy / fill / read-reduce over buffers far larger than the caches."""
import torch
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3
for gb in (1, 4):
    n = gb * (1 << 30) // 4
    x = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    y = torch.empty_like(x)
    dt = t(lambda: y.copy_(x)); print(f"{gb} GiB copy  (r+w): {2 * n * 4 / dt / 1e12:.2f} TB/s")
    dt = t(lambda: y.fill_(1.0)); print(f"{gb} GiB fill  (w)  : {n * 4 / dt / 1e12:.2f} TB/s")
    dt = t(lambda: x.sum()); print(f"{gb} GiB sum   (r)  : {n * 4 / dt / 1e12:.2f} TB/s")
    z = torch.empty_like(x)
    dt = t(lambda: torch.add(x, y, out=z)); print(f"{gb} GiB add (2r+w): {3 * n * 4 / dt / 1e12:.2f} TB/s")
    del x, y, z
