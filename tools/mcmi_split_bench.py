#!/usr/bin/env python3
"""MCMI_min, batches of 5 .. 8 on USPS-shaped data (9298 x 256, subsample 1000): the split scorer against the single
kernel, per greedy step (HIP events).    python tools/mcmi_split_bench.py [k ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ital_amd import MCMI_min
ks = [int(a) for a in sys.argv[1:]] or [6, 8]
X = np.random.default_rng(0).random((9298, 256))
for k in ks:
    for split in (False, True):
        np.random.seed(0)
        L = MCMI_min(X, length_scale=3.0, subsample=1000, device="cuda:0")
        L.split_kernel = split
        L.update({0: 1})
        L.fetch_unlabelled(k)                       # warm-up
        L.profile = []
        np.random.seed(1)
        torch.cuda.synchronize()
        ret = L.fetch_unlabelled(k)
        torch.cuda.synchronize()
        by = {}
        for name, t, size, e0, e1 in L.profile:
            by[(name, t)] = e0.elapsed_time(e1)
        steps = {t: round(v, 3) for (name, t), v in sorted(by.items()) if name == "mcmi_score"}
        print("k=%d %-6s picks %s  score steps ms %s  total %.2f ms" % (k, "split" if split else "single", ret, steps, sum(steps.values())))
