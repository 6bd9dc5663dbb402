"""HBM calibration with plain torch kernels: what fraction of the 8 TB/s a pure read, a pure write and a copy reach."""
import torch, time
dev = "cuda:0"
n = 1 << 30   # 1 GiB
a = torch.empty(n, dtype=torch.uint8, device=dev).view(torch.float32)
b = torch.empty(n, dtype=torch.uint8, device=dev).view(torch.float32)
def t(f, iters=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
dt = t(lambda: a.fill_(1.0)); print(f"fill  (write 1 GiB)        {n / dt / 1e12:.2f} TB/s")
dt = t(lambda: b.copy_(a)); print(f"copy  (read + write 2 GiB) {2 * n / dt / 1e12:.2f} TB/s")
dt = t(lambda: a.sum()); print(f"sum   (read 1 GiB)         {n / dt / 1e12:.2f} TB/s")
c = torch.empty(n // 4, dtype=torch.uint8, device=dev).view(torch.float32)
dt = t(lambda: torch.add(a[: n // 16], 1.0, out=c)); print(f"add   (read 256 MiB + write 256 MiB) {n / 2 / dt / 1e12:.2f} TB/s")
