#!/usr/bin/env python3
"""Host-side latency of the GPU box: kernel launch, synchronisation, small copies (explains run-to-run differences of
the per-round wall clock that the kernel times do not show)."""
import time
import torch
x = torch.zeros(1024, device="cuda")
torch.cuda.synchronize()
def t(fn, n=2000):
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
print("launch (x.add_(1))      %.1f us" % t(lambda: x.add_(1)))
print("launch + sync           %.1f us" % t(lambda: (x.add_(1), torch.cuda.synchronize()), 500))
print("D2H 8 bytes (.item())   %.1f us" % t(lambda: x[0].item(), 500))
h = torch.zeros(9298, dtype=torch.int32)
print("H2D 37 KB               %.1f us" % t(lambda: h.to("cuda"), 500))
import os
print("cpus", os.cpu_count(), "loadavg", os.getloadavg())
