#!/usr/bin/env python3
"""One fetch_unlabelled(k) at the larger BASELINE configurations on a single GPU (shape check, memory, time per step).
    python tools/scale_probe.py n d k [monte_carlo_num_rel]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ital_amd import ITAL, mvn_stream
n, d, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mc = int(sys.argv[4]) if len(sys.argv) > 4 else None
X = np.random.default_rng(0).random((n, d))
t0 = time.perf_counter()
L = ITAL(X, length_scale=float(np.sqrt(d / 12.0)), monte_carlo_num_rel=mc, device="cuda:0")
L.update({0: 1, 1: -1, 2: 1})
torch.cuda.synchronize()
print("fit + first update %.2f s, mem %.2f GB" % (time.perf_counter() - t0, torch.cuda.memory_allocated() / 2**30))
L.profile = []
np.random.seed(0)
t0 = time.perf_counter()
ret = L.fetch_unlabelled(k)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("fetch_unlabelled(%d) on %d x %d: %.3f s -> %.0f scored candidates/s; picks %s" % (k, n, d, dt, k * n / dt, ret))
by = {}
for name, t, size, e0, e1 in L.profile:
    by.setdefault((name, t), []).append(e0.elapsed_time(e1))
print({"%s_t%d" % key: round(float(np.mean(v)), 2) for key, v in sorted(by.items())})
print("peak mem %.2f GB" % (torch.cuda.max_memory_allocated() / 2**30))
