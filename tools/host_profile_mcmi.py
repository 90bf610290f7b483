#!/usr/bin/env python3
"""cProfile of the host side of an MCMI_min round (subsample 1000: the round is host-bound, 0.2 ms of kernels)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ital_amd import MCMI_min

n, d, k = 9298, 256, 4
X = np.random.default_rng(0).random((n, d))
rel = np.where(X[:, 0] > 0.5, 1.0, -1.0)
np.random.seed(0)
L = MCMI_min(X, length_scale=3.0, subsample=1000, device="cuda:0")
L.update({0: 1})
for _ in range(3):
    r = L.fetch_unlabelled(k)
    L.update({int(i): float(rel[i]) for i in r})
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    r = L.fetch_unlabelled(k)
    L.update({int(i): float(rel[i]) for i in r})
torch.cuda.synchronize()
print("ms per round %.3f" % ((time.perf_counter() - t0) / 20 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    r = L.fetch_unlabelled(k)
    L.update({int(i): float(rel[i]) for i in r})
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
