#!/usr/bin/env python3
"""What the HBM of this GPU delivers to a plain streaming kernel (read-only sum, and copy = read + write), to put beside
the 8 TB/s peak the roofline figures are divided by.  2 GiB buffers, far beyond the 256 MB of last-level cache."""
import torch

n = 1 << 29                      # fp32 elements = 2 GiB
a = torch.ones(n, dtype=torch.float32, device="cuda")
b = torch.empty_like(a)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / 1e3


t_copy = timed(lambda: b.copy_(a))
t_read = timed(lambda: a.sum())
print(f"copy: {2 * 4 * n / t_copy / 1e9:.0f} GB/s (read + write), read-only sum: {4 * n / t_read / 1e9:.0f} GB/s")
