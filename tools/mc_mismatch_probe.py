#!/usr/bin/env python3
"""Monte-Carlo pattern replay: how often do the device learner's sampled sign patterns differ from the oracle's (numpy's
multivariate_normal maps its normals through an SVD of the candidate's covariance), and what do the offending covariance
matrices look like?   python tools/mc_mismatch_probe.py [n] [k] [samples per step]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if __name__ == "__main__":
    import torch
    import test_gpu_scale as tgs
    from ital_amd import ITAL, mvn_stream
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    d, mc = 512, 1
    X = np.random.default_rng(7).random((n, d))
    ls = float(np.sqrt(d / 12.0))
    mvn_stream.GLOBAL.reset()
    L = ITAL(X, length_scale=ls, monte_carlo_num_rel=mc, device="cuda:0")
    L.keep_scores = True
    L.update({0: 1, 1: -1, 2: 1})
    cand0 = np.asarray(L.get_unseen())
    stream0 = (mvn_stream.GLOBAL.state, mvn_stream.GLOBAL.draws)
    np.random.seed(11)
    picks = L.fetch_unlabelled(k)
    scores = [s.cpu().numpy() for s in L.last_scores]
    pos_of = {int(c): p for p, c in enumerate(cand0)}
    pick_pos = [pos_of[int(p)] for p in picks]
    npat = lambda t: L._mc_plan(t, 0)[1]
    draws = lambda t: npat(t) * 2 * mvn_stream.draws_per_call(t)
    normals = lambda t: npat(t) * t if L._mc_plan(t, 0)[0] else 0
    samples = tgs._sample_positions(np.random.default_rng(8), len(cand0), pick_pos, k, lambda t: per)
    ntask, bad = tgs._check_against_sub_oracle(X, ls, L, picks, cand0, scores, samples, dict(monte_carlo_num_rel=mc), stream0,
                                               draws, seed=11, normals_per_cand=normals, workers=12,
                                               patterns=[np.asarray(a) for a in L.last_patterns], allow_resampled=1.0)
    print("checked", ntask, "mismatches", len(bad))
    by_t = {}
    for t, p, mine, val in bad:
        by_t.setdefault(t, []).append((p, mine, val))
    for t in sorted(by_t):
        print("step", t, by_t[t][:4])
    # covariance spectrum of the offending candidates (oracle side, dense sub-problem)
    from oracle.ital import OracleITAL, _Appended
    for t, p, mine, val in bad[:6]:
        ids = sorted(set(L.gp.ind) | set(picks) | {int(cand0[p])})
        sub = {g: s for s, g in enumerate(ids)}
        R = OracleITAL(X[np.asarray(ids)], length_scale=ls)
        R.update({sub[int(i)]: float(y) for i, y in zip(L.gp.ind, L.gp.y)})
        R._ce_subset = None
        st = _Appended(R)
        for q in picks[: t - 1]:
            st.append(sub[int(q)])
        cov = st.covs[sub[int(cand0[p])]]
        sv = np.linalg.svd(cov, compute_uv=False)
        gaps = np.abs(np.diff(sv)) / sv[:-1]
        print("t", t, "pos", p, "singular values", sv[:4], "...", sv[-2:], "smallest relative gap %.2e" % gaps.min())
