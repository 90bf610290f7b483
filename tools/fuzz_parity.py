#!/usr/bin/env python3
"""Randomised differential test: device learners against the oracle over random sizes and option combinations
(every scorer path).  Not part of the pytest suite (run time); prints one line per case and a summary.
    python tools/fuzz_parity.py [cases] [seed] [only-case]      (a third argument reruns one case verbosely)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ital_amd import ITAL, MCMI_min, mvn_stream  # noqa: E402
from oracle import mvn as omvn  # noqa: E402
from oracle.ital import OracleITAL, OracleMCMI  # noqa: E402
from oracle.baselines import OracleBorderDiv, OracleEMOC, OracleEntropy  # noqa: E402
from ital_amd.baselines import EMOC, BorderlineDiversitySampling, EntropySampling  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
only = int(sys.argv[3]) if len(sys.argv) > 3 else None
bad = 0
t_start = time.time()
for case in range(cases):
    if only is not None and case != only:
        continue
    rng = np.random.default_rng(seed0 * 1000 + case)
    n = int(rng.integers(12, 90))
    d = int(rng.integers(2, 10))
    k = int(rng.integers(1, 5))
    X = rng.random((n, d))
    if rng.random() < 0.3:                      # duplicated rows now and then
        X[int(rng.integers(0, n))] = X[int(rng.integers(0, n))]
    ls = float(np.sqrt(d / 12.0) * rng.uniform(0.5, 1.5))
    kw = {}
    kind = rng.choice(["perfect", "noisy", "motivated", "subset", "mc", "clip", "mcmi", "optimistic", "topcand", "mix", "emoc", "entropy", "borderdiv", "bigk"])
    if kind == "noisy":
        kw = dict(label_prob=float(rng.uniform(0.3, 0.9)), mistake_prob=float(rng.uniform(0.0, 0.4)))
    elif kind == "motivated":
        kw = dict(mistake_prob=float(rng.uniform(0.05, 0.4)))
    elif kind == "subset":
        kw = dict(change_estimation_subset=int(rng.integers(1, 6)))
    elif kind == "bigk":                          # full enumeration of larger batches on a small candidate set
        k = int(rng.integers(5, 7))
        n = int(rng.integers(12, 26))
        X = X[:n] if n <= len(X) else rng.random((n, d))
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
    if kind in ("emoc", "entropy", "borderdiv"):
        dcls, ocls = {"emoc": (EMOC, OracleEMOC), "entropy": (EntropySampling, OracleEntropy),
                      "borderdiv": (BorderlineDiversitySampling, OracleBorderDiv)}[kind]
        if kind == "borderdiv":
            kw = dict(alpha=float(rng.uniform(0.1, 0.9)))
        A = dcls(X, length_scale=ls, device="cuda:0", **kw)
        B = ocls(X, length_scale=ls, **kw)
    elif kind == "mcmi":
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
            def fetch(learner):
                np.random.seed(case * 7 + rnd)
                try:
                    return [int(i) for i in learner.fetch_unlabelled(k)], None
                except ValueError as e:            # top_candidates below k: np.argmax([]) in the reference (ital.py:130)
                    return None, str(e)
            got, err_a = fetch(A)
            want, err_b = fetch(B)
            if err_a or err_b:
                if not (err_a and err_b and "empty sequence" in err_a and "empty sequence" in err_b):
                    status = "ERRORS differ: %r vs %r" % (err_a, err_b)
                break
            worst = 0.0
            if kind == "emoc":
                keep = np.array([c not in twin for c in B.last_candidates])
                worst = float(np.max(np.abs(A.last_scores[keep] - B.last_scores[keep]) / np.abs(B.last_scores[keep]))) if keep.any() else 0.0
                worst *= 1e-5 / 1e-6                 # EMOC scores agree to 1e-6 (the bar of the golden tests; 2.8e-7 seen at d = 2)
            traced = [] if kind in ("emoc", "borderdiv") else [(tr[0], tr[1]) for tr in B.trace]
            pos = {c: i for i, c in enumerate(traced[0][0])} if traced else {}
            for t, (cand, vals) in enumerate(traced):
                mine = A.last_scores[t].cpu().numpy()[[pos[c] for c in cand]]
                keep = np.array([c not in twin for c in cand])
                # a twin inside the change-estimation subset or the batch so far puts the same degeneracy into every
                # candidate's problem (observed: a common 1e-5 shift of all scores): loosen the tolerance for the step
                fixed = set(int(i) for i in (getattr(B, "_ce_subset", None) or [])) | set(want[:t])
                tol_t = 1e-3 if (twin & fixed) else 1e-5
                mine, vals = mine[keep], vals[keep]
                both = ~(np.isnan(mine) | np.isnan(vals))
                if not np.array_equal(np.isnan(mine), np.isnan(vals)):
                    status = "NAN-MISMATCH"
                if only is not None:
                    print("round", rnd, "step", t, "ce subset", getattr(B, "_ce_subset", None), "twins", sorted(twin))
                    for c, a_, b_ in zip(np.array(cand)[keep], mine, vals):
                        print("   cand %3d  device % .12e  oracle % .12e  rel %.2e" % (c, a_, b_, abs(a_ - b_) / max(abs(b_), 1e-9)))
                if both.any():
                    worst = max(worst, float(np.max(np.abs(mine[both] - vals[both]) / np.maximum(np.abs(vals[both]), 1e-9))) * 1e-5 / tol_t)
            if got != want and [int(inv.ravel()[i]) for i in got] == [int(inv.ravel()[i]) for i in want]:
                status = "ok"                       # a tie between identical rows resolved the other way: same batch
                break
            if got != want and traced:
                # a numerical tie: at the first step that differs the oracle itself rates the two samples equal to 1e-12
                # (candidates that carry no information: all MI values agree to the last bits, the last bit picks)
                t0 = next(i for i in range(min(len(got), len(want))) if got[i] != want[i])
                cand0, vals0 = traced[t0]
                at = {int(c): float(v) for c, v in zip(cand0, vals0)}
                if got[t0] in at and abs(at[got[t0]] - at[want[t0]]) <= 1e-12 * abs(at[want[t0]]):
                    status = "ok"
                    print("case %3d: numerical tie at step %d (oracle MI %.17g vs %.17g), batch not compared further"
                          % (case, t0, at[got[t0]], at[want[t0]]), flush=True)
                    break
            if got != want:
                status = "PICKS %s != %s" % (got, want)
            elif worst > 1e-5:
                status = "SCORES rel err %.2e" % worst
            if status != "ok":
                break
            fb = {i: (1 if X[i, 0] > 0.5 else -1) for i in got}
            A.update(fb)
            B.update(fb)
        if kind not in ("mcmi", "emoc", "entropy", "borderdiv") and status == "ok" and mvn_stream.GLOBAL.draws != omvn.rng_draws():
            status = "STREAM %d != %d" % (mvn_stream.GLOBAL.draws, omvn.rng_draws())
    except Exception as e:  # noqa: BLE001
        status = "EXC %s: %s" % (type(e).__name__, str(e)[:80])
    bad += status != "ok"
    print("case %3d %-10s n=%3d d=%2d k=%d %-60s %s" % (case, kind, n, d, k, str(kw)[:60], status), flush=True)
print("%d cases, %d failures, %.0f s" % (cases, bad, time.time() - t_start))
sys.exit(1 if bad else 0)
