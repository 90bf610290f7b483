#!/usr/bin/env python3
"""cProfile of the host side of the benchmark round (fetch_unlabelled + update) on the GPU box."""
import cProfile
import os
import pstats
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ital_amd import ITAL, mvn_stream

n, d, k = 9298, 256, 4
X = np.random.default_rng(0).random((n, d))
rel = np.where(X[:, 0] > 0.5, 1.0, -1.0)
L = ITAL(X, length_scale=3.0, device="cuda:0")
L.update({0: 1})
for _ in range(3):
    r = L.fetch_unlabelled(k)
    L.update({int(i): float(rel[i]) for i in r})
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    r = L.fetch_unlabelled(k)
    L.update({int(i): float(rel[i]) for i in r})
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
