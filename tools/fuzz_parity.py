#!/usr/bin/env python3
"""Randomised differential test: device learners against the oracle over random sizes and option combinations
(every scorer path).  Not part of the pytest suite (run time); prints one line per case and a summary.
    python tools/fuzz_parity.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ital_amd import ITAL, MCMI_min, mvn_stream  # noqa: E402
from oracle import mvn as omvn  # noqa: E402
from oracle.ital import OracleITAL, OracleMCMI  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
t_start = time.time()
for case in range(cases):
    rng = np.random.default_rng(seed0 * 1000 + case)
    n = int(rng.integers(12, 90))
    d = int(rng.integers(2, 10))
    k = int(rng.integers(1, 5))
    X = rng.random((n, d))
    if rng.random() < 0.3:                      # duplicated rows now and then
        X[int(rng.integers(0, n))] = X[int(rng.integers(0, n))]
    ls = float(np.sqrt(d / 12.0) * rng.uniform(0.5, 1.5))
    kw = {}
    kind = rng.choice(["perfect", "noisy", "motivated", "subset", "mc", "clip", "mcmi", "optimistic", "topcand", "mix"])
    if kind == "noisy":
        kw = dict(label_prob=float(rng.uniform(0.3, 0.9)), mistake_prob=float(rng.uniform(0.0, 0.4)))
    elif kind == "motivated":
        kw = dict(mistake_prob=float(rng.uniform(0.05, 0.4)))
    elif kind == "subset":
        kw = dict(change_estimation_subset=int(rng.integers(1, 6)))
    elif kind == "mc":
        kw = dict(monte_carlo_num_rel=int(rng.integers(1, 3)))
        k = int(rng.integers(3, 7))
    elif kind == "clip":
        kw = dict(clip_cov=float(rng.uniform(0.1, 0.6)), change_estimation_subset=int(rng.integers(3, 6)))
    elif kind == "optimistic":
        kw = dict(label_estimation=str(rng.choice(["optimistic", "pessimistic"])))
    elif kind == "topcand":
        kw = dict(top_candidates=int(rng.integers(3, 12)))
    elif kind == "mix":
        kw = dict(label_prob=float(rng.uniform(0.4, 0.9)), mistake_prob=float(rng.uniform(0.0, 0.3)),
                  change_estimation_subset=int(rng.integers(1, 4)), monte_carlo_num_fb=int(rng.integers(1, 3)))
    labels = {int(i): (1 if X[i, 0] > 0.5 else -1) for i in rng.choice(n, int(rng.integers(1, 5)), replace=False)}
    # rows that have an exact twin: their orthant problems are degenerate (correlation +-1 up to rounding) and the value
    # of the reference itself hangs on the last bit of the BLAS in use -- scores of such candidates are not compared
    _, inv, cnt = np.unique(X, axis=0, return_inverse=True, return_counts=True)
    twin = set(np.flatnonzero(cnt[inv.ravel()] > 1).tolist())
    mvn_stream.GLOBAL.reset()
    omvn.rng_reset()
    if kind == "mcmi":
        A = MCMI_min(X, length_scale=ls, subsample=int(rng.integers(8, n)) if rng.random() < 0.5 else None, device="cuda:0")
        B = OracleMCMI(X, length_scale=ls, subsample=A.subsample)
    else:
        A = ITAL(X, length_scale=ls, device="cuda:0", **kw)
        B = OracleITAL(X, length_scale=ls, **kw)
    A.keep_scores = True
    A.update(labels)
    B.update(labels)
    status = "ok"
    try:
        for rnd in range(2):
            np.random.seed(case * 7 + rnd)
            got = A.fetch_unlabelled(k)
            np.random.seed(case * 7 + rnd)
            want = [int(i) for i in B.fetch_unlabelled(k)]
            cand0 = B.trace[0][0]
            pos = {c: i for i, c in enumerate(cand0)}
            worst = 0.0
            for t, (cand, vals, _) in enumerate(B.trace):
                mine = A.last_scores[t].cpu().numpy()[[pos[c] for c in cand]]
                keep = np.array([c not in twin for c in cand])
                mine, vals = mine[keep], vals[keep]
                both = ~(np.isnan(mine) | np.isnan(vals))
                if not np.array_equal(np.isnan(mine), np.isnan(vals)):
                    status = "NAN-MISMATCH"
                if both.any():
                    worst = max(worst, float(np.max(np.abs(mine[both] - vals[both]) / np.maximum(np.abs(vals[both]), 1e-9))))
            if got != want:
                status = "PICKS %s != %s" % (got, want)
            elif worst > 1e-5:
                status = "SCORES rel err %.2e" % worst
            if status != "ok":
                break
            fb = {i: (1 if X[i, 0] > 0.5 else -1) for i in got}
            A.update(fb)
            B.update(fb)
        if kind != "mcmi" and status == "ok" and mvn_stream.GLOBAL.draws != omvn.rng_draws():
            status = "STREAM %d != %d" % (mvn_stream.GLOBAL.draws, omvn.rng_draws())
    except ValueError as e:
        # top_candidates below k: the reference itself fails with np.argmax([]) (ital.py:130) -- both sides must
        ok_both = "empty sequence" in str(e)
        if ok_both:
            try:
                np.random.seed(case * 7)
                (A if "A_raised" not in locals() else B).fetch_unlabelled(k)
                ok_both = False
            except ValueError:
                pass
        status = "ok" if ok_both else "EXC ValueError: %s" % str(e)[:80]
    except Exception as e:  # noqa: BLE001
        status = "EXC %s: %s" % (type(e).__name__, str(e)[:80])
    bad += status != "ok"
    print("case %3d %-10s n=%3d d=%2d k=%d %-60s %s" % (case, kind, n, d, k, str(kw)[:60], status), flush=True)
print("%d cases, %d failures, %.0f s" % (cases, bad, time.time() - t_start))
sys.exit(1 if bad else 0)
