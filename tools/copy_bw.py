#!/usr/bin/env python3
"""What a plain device-to-device copy reaches on this GPU (read + write bytes per second)."""
import torch
for mb in (256, 1024, 2048, 4096):
    n = mb * (1 << 20) // 8
    a = torch.ones(n, dtype=torch.float64, device="cuda")
    b = torch.empty_like(a)
    for _ in range(5):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("copy %5d MB: %.4f ms  %.0f GB/s (read + write)" % (mb, ms, 2 * n * 8 / ms * 1e-6))
    s = a.sum()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        s = a.sum()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("read %5d MB: %.4f ms  %.0f GB/s (torch sum)" % (mb, ms, n * 8 / ms * 1e-6))
