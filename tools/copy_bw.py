"""Developer helper: achievable device-to-device copy bandwidth (read + written bytes) at the bench's sizes."""
import torch, time
for mb in (201, 402, 1024):
    n = mb * 1024 * 1024
    a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
    a.random_(0, 255)
    for _ in range(5): b.copy_(a)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(50): b.copy_(a)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 50 * 1e3
    print(f"{mb} MiB copy: {us:.1f} us  {2 * n / us / 1e6:.2f} TB/s (read+write)")
