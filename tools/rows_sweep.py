#!/usr/bin/env python3
"""t = 3, 4 scorer time against the number of candidates and the work-item split (grid quantisation study)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ital_amd import ITAL, mvn_stream
d, k = 256, 4
for split in (1, 2, 4):
    out = []
    for n in (3072, 3200, 6144, 6300, 9216, 9298, 12288, 12400):
        X = np.random.default_rng(0).random((n + 1, d))
        L = ITAL(X, length_scale=3.0, device="cuda:0")
        L.qmc_split = split
        L.update({0: 1})
        L.fetch_unlabelled(k)
        L.profile = []
        for _ in range(3):
            L.fetch_unlabelled(k)
        torch.cuda.synchronize()
        ts = {}
        for name, t, size, e0, e1 in L.profile:
            if name == "score":
                ts.setdefault(t, []).append(e0.elapsed_time(e1))
        out.append("%d: t3 %.3f t4 %.3f" % (n, np.mean(ts[3]), np.mean(ts[4])))
    print("split", split, " | ".join(out))
