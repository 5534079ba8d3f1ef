"""Achievable HBM bandwidth of the box: device-to-device copy of a 4 GiB buffer (read + write), best of 10."""
import torch
n = 4 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda")
b = torch.empty_like(a)
a.fill_(1)
best = 1e9
for _ in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); b.copy_(a); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print("copy 4 GiB: %.3f ms -> %.2f TB/s (read + write)" % (best, 2 * n / best / 1e9))
