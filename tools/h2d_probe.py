"""Pinned host -> device copy rate of this box (the loader's ceiling)."""
import time, torch
x = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True)
y = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
for n in (64 << 20, 256 << 20, 1 << 30):
    y[:n].copy_(x[:n], non_blocking=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        y[:n].copy_(x[:n], non_blocking=True)
    torch.cuda.synchronize()
    print("H2D %4d MiB chunks: %.1f GB/s" % (n >> 20, 5 * n / (time.perf_counter() - t0) / 1e9))
t0 = time.perf_counter()
for _ in range(5):
    x.copy_(y, non_blocking=True)
torch.cuda.synchronize()
print("D2H 1024 MiB: %.1f GB/s" % (5 * (1 << 30) / (time.perf_counter() - t0) / 1e9))
