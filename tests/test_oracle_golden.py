"""Pins the oracle (oracle/gp.py, oracle/ital.py) to the golden vectors produced by the real reference
(tests/golden/make_golden.py, serial mode, fresh process): GP state, predictive mean/variance, the full MI
vector of every greedy step and the picks -- for every scorer mode the reference has."""
import os
import sys

import numpy as np
import pytest

from oracle import mvn
from oracle.gp import OracleGP, rbf_kernel
from oracle.ital import OracleITAL, OracleMCMI

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden  # noqa: E402  (fixture table only; nothing of the reference is imported here)


def replay(golden_dir, name, rounds=None):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    spec = make_golden.FIXTURES[name]
    mvn.rng_reset()
    np.random.seed(0)
    cls = OracleITAL if spec["learner"] == "ITAL" else OracleMCMI
    L = cls(z["X"], length_scale=float(z["length_scale"]), **spec["kw"])
    L.update({int(z["query"]): 1})
    rel = z["rel"]
    for r in range(int(z["rounds"]) if rounds is None else rounds):
        m, v = L.gp.predict_stored(cov_mode="diag")
        assert L.gp.ind == z[f"r{r}_ind"].tolist()
        np.testing.assert_allclose(m, z[f"r{r}_rel_mean"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(v, z[f"r{r}_var"], rtol=1e-8, atol=1e-12)
        ret = L.fetch_unlabelled(int(z["k"]))
        assert [int(i) for i in ret] == z[f"r{r}_ret"].tolist()
        for t, (cand, vals, pick) in enumerate(L.trace):
            assert cand == z[f"r{r}_s{t}_cand"].tolist()
            ref = z[f"r{r}_s{t}_mi"]
            np.testing.assert_allclose(vals, ref, rtol=1e-9, atol=1e-12)
            assert pick == int(z[f"r{r}_s{t}_pick"])
        L.update({int(i): float(rel[i]) for i in ret})
    return L, z


@pytest.mark.parametrize("name,rounds", [("usps500", 1), ("synth200_noisy", 1), ("synth200_motivated", 1),
                                         ("synth200_optimistic", 1), ("synth200_topcand", 2), ("synth200_topcand_float", 3), ("iris_ce5", 1),
                                         ("usps500_mcmi", 2), ("synth300_mcmi", 2), ("synth80_mcrel", 1),
                                         ("synth60_mcfb", 2), ("synth50_mcboth", 1), ("synth50_clip", 2)])
def test_oracle_reproduces_reference(golden_dir, name, rounds):
    replay(golden_dir, name, rounds)


def test_oracle_gp_predict(golden_dir):
    L, z = replay(golden_dir, "synth200_topcand")
    np.testing.assert_allclose(L.rel_mean, z["final_rel_mean"], rtol=0, atol=1e-12)
    assert L.top_results(10).tolist() == z["top_results_10"].tolist()
    pm, pv = L.gp.predict(z["predict_X"], cov_mode="diag")
    np.testing.assert_allclose(pm, z["predict_mean"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(pv, z["predict_var"], rtol=1e-8, atol=1e-12)


def test_mvndst_call_log(golden_dir):
    """The reference's own first mvndst calls (inputs and outputs logged by the fixture generator)."""
    z = np.load(os.path.join(golden_dir, "synth96_k6.npz"))
    n_all = z["mvn_n"]
    mvn.rng_reset()
    j = 0
    for c in range(len(n_all)):
        n = int(n_all[c])
        if n < 2:
            continue
        if j >= int(z["mvnlog_count"]):
            break
        nc = n * (n - 1) // 2
        _, v, _ = mvn.mvndst(z["mvnlog_lower"][j, :n], z["mvnlog_lower"][j, :n], z["mvnlog_infin"][j, :n],
                             z["mvnlog_correl"][j, :nc], maxpts=100 * n, abseps=1e-4, releps=1e-4)
        assert abs(v - z["mvn_val"][c]) <= 1e-15, (c, n)
        j += 1
    assert j == int(z["mvnlog_count"])


def test_update_equals_fit():
    rng = np.random.default_rng(3)
    X = rng.random((60, 5))
    a = OracleGP(X, 0.7)
    a.fit([1, 5, 9, 20], [1, -1, 1, -1])
    b = OracleGP(X, 0.7)
    b.update([1, 5], [1, -1])
    b.update([9, 20], [1, -1])
    np.testing.assert_allclose(a.predict_stored(), b.predict_stored(), atol=1e-12)
    K = rbf_kernel(X[:3], X[:4], 0.7, 1.0)
    assert K.shape == (3, 4) and abs(K[1, 1] - 1.0) < 1e-15
