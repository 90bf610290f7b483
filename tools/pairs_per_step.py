#!/usr/bin/env python3
"""(Phi, Phi^-1) pairs the general scorer really integrates per greedy step (device counter ital_gscore_desc.pair_count)
next to the step times: pairs/s per dimension.   python tools/pairs_per_step.py n d k [monte_carlo_num_rel]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ital_amd import ITAL, mvn_stream
n, d, kmax = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mc = int(sys.argv[4]) if len(sys.argv) > 4 else None
X = np.random.default_rng(0).random((n, d))
prev_pairs, prev_ms = 0.0, 0.0
for k in range(max(3, kmax - 7), kmax + 1):
    mvn_stream.GLOBAL.reset()
    L = ITAL(X, length_scale=float(np.sqrt(d / 12.0)), monte_carlo_num_rel=mc, device="cuda:0")
    L.update({0: 1, 1: -1, 2: 1})
    L.pair_counter = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    L.profile = []
    np.random.seed(0)
    L.fetch_unlabelled(k)
    torch.cuda.synchronize()
    pairs = float(L.pair_counter.item())
    ms = sum(e0.elapsed_time(e1) for name, t, size, e0, e1 in L.profile if name.startswith("score") and t == k)
    print("t = %2d: %.3e pairs in %.1f ms = %.0f G pairs/s (%.0f pairs per candidate)" % (k, pairs - prev_pairs, ms, (pairs - prev_pairs) / ms / 1e6, (pairs - prev_pairs) / n), flush=True)
    prev_pairs = pairs
