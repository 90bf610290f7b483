#!/usr/bin/env python3
"""Wall-clock split of the benchmark round on the GPU box: fetch_unlabelled vs update (each followed by a device sync)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ital_amd import ITAL, mvn_stream
n, d, k = 9298, 256, 4
X = np.random.default_rng(0).random((n, d))
rel = np.where(X[:, 0] > 0.5, 1.0, -1.0)
L = ITAL(X, length_scale=3.0, device="cuda:0")
L.update({0: 1})
for _ in range(2):
    r = L.fetch_unlabelled(k); L.update({int(i): float(rel[i]) for i in r})
L.reset(); mvn_stream.GLOBAL.reset(); L.update({0: 1})
torch.cuda.synchronize()
tf = tu = 0.0
rows = []
for _ in range(10):
    t0 = time.perf_counter()
    r = L.fetch_unlabelled(k)
    t1 = time.perf_counter()
    L.update({int(i): float(rel[i]) for i in r})
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
print("fetch ms:", " ".join("%.2f" % a for a, _ in rows))
print("update ms:", " ".join("%.2f" % b for _, b in rows))
