#!/usr/bin/env python3
"""cProfile of one fetch_unlabelled(16) with monte_carlo_num_rel = 1 on 125 000 x 512 (the C5' share): where the host's
time goes next to the GPU's (the waits show up as Tensor.cpu).  Run on the GPU box from the repo root."""
import cProfile, pstats, sys, os, io
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from ital_amd import ITAL
n, d, k = 125000, 512, 16
X = np.random.default_rng(0).random((n, d))
L = ITAL(X, length_scale=float(np.sqrt(d / 12.0)), monte_carlo_num_rel=1, device="cuda:0")
L.update({0: 1, 1: -1, 2: 1})
np.random.seed(0)
pr = cProfile.Profile()
pr.enable()
L.fetch_unlabelled(k)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print(s.getvalue()[:4000])
