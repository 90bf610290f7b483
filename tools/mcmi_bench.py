#!/usr/bin/env python3
"""MCMI_min on the device at the reference's config size (usps.conf: subsample 1000) and on all 9298 candidates:
time per fetch_unlabelled(k), the dense covariance block (FP64 MFMA) and the pairwise scorer per greedy step.
    python tools/mcmi_bench.py [1000 | all]      (one size only: what the counter passes of tools/profile_r6.sh run, so that
                                                  the per-launch averages of a kernel are those of ONE problem size)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ital_amd import MCMI_min

n, d, k = 9298, 256, 4
X = np.random.default_rng(0).random((n, d))
rel = np.where(X[:, 0] > 0.5, 1.0, -1.0)
only = sys.argv[1] if len(sys.argv) > 1 else None
for sub in ((1000, None) if only is None else ((1000,) if only == "1000" else (None,))):
    L = MCMI_min(X, length_scale=3.0, subsample=sub, device="cuda:0")
    L.update({0: 1})
    np.random.seed(0)
    r = L.fetch_unlabelled(k)
    L.update({int(i): float(rel[i]) for i in r})
    torch.cuda.synchronize()
    for _ in range(3):                                   # (clocks up, lazily loaded code in)
        r = L.fetch_unlabelled(k)
        L.update({int(i): float(rel[i]) for i in r})
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):                                # unprofiled: the round is one call below the C ABI
        r = L.fetch_unlabelled(k)
        L.update({int(i): float(rel[i]) for i in r})
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    L.profile = []
    for _ in range(3):                                   # kernel times: step by step with events
        r = L.fetch_unlabelled(k)
        L.update({int(i): float(rel[i]) for i in r})
    torch.cuda.synchronize()
    prof = {}
    for name, t, size, e0, e1 in L.profile:
        prof.setdefault((name, t), []).append((e0.elapsed_time(e1), size))
    nc = sub or (n - len(L.relevant_ids) - len(L.irrelevant_ids))
    m = L.gp.m
    out = {"candidates": nc, "ms_per_round": dt * 1e3, "candidates_scored_per_s": k * nc / dt}
    for (name, t), v in sorted(prof.items()):
        ms = float(np.mean([a for a, _ in v]))
        out["%s_t%d_ms" % (name, t)] = ms
        if name == "cov_block":
            out["cov_block_tflops"] = 2.0 * nc * nc * (d + m) / (ms * 1e-3) / 1e12
    print(out)
