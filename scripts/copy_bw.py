#!/usr/bin/env python3
"""Developer probe: streaming-copy bandwidth of this box's GPU (read + write bytes / time)."""
import numpy as np
import torch

dev = torch.device("cuda:0")
for gib in (1, 4):
    n = gib * (1 << 28)
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    ev = []
    for _ in range(20):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); b.copy_(a); e.record(); ev.append((s, e))
    torch.cuda.synchronize()
    t = np.array([s.elapsed_time(e) for s, e in ev])
    print("copy %d GiB: median %.3f ms -> %.0f GB/s (read+write)" % (gib, np.median(t), 2 * n * 4 / np.median(t) / 1e6))
    # read-only: sum
    ev = []
    for _ in range(10):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); a.sum(); e.record(); ev.append((s, e))
    torch.cuda.synchronize()
    t = np.array([s.elapsed_time(e) for s, e in ev])
    print("sum  %d GiB: median %.3f ms -> %.0f GB/s (read)" % (gib, np.median(t), n * 4 / np.median(t) / 1e6))
    del a, b
