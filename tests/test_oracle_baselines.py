"""CPU: the oracle's restatements of EMOC / EntropySampling / BorderlineDiversitySampling against the golden vectors the
real reference produced (tests/golden/make_golden_baselines.py): picks bit-exact, scores to rounding (the batch
entropies of steps >= 3 included: they pin the one-worker pool schedule and the MVNDST stream handling)."""
import os

import numpy as np
import pytest

from oracle import mvn
from oracle.baselines import OracleBorderDiv, OracleEMOC, OracleEntropy

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


@pytest.mark.parametrize("name", ["emoc_synth150", "emoc_usps500"])
def test_emoc(name):
    g = _load(name)
    L = OracleEMOC(g["X"], length_scale=float(g["length_scale"]))
    L.update({int(g["query"]): 1})
    for r in range(int(g["rounds"])):
        ret = L.fetch_unlabelled(int(g["k"]))
        assert np.array_equal(L.last_candidates, g[f"r{r}_cand"])
        np.testing.assert_allclose(L.last_scores, g[f"r{r}_scores"], rtol=1e-9, atol=0)
        assert ret == g[f"r{r}_ret"].tolist()
        L.update({int(i): float(g["rel"][i]) for i in ret})
    np.testing.assert_allclose(L.rel_mean, g["final_rel_mean"], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("name", ["entropy_synth80", "entropy_usps300"])
def test_entropy(name):
    g = _load(name)
    mvn.rng_reset()
    L = OracleEntropy(g["X"], length_scale=float(g["length_scale"]))
    L.update({int(g["query"]): 1})
    for r in range(int(g["rounds"])):
        before = mvn.rng_state()
        ret = L.fetch_unlabelled(int(g["k"]))
        assert ret == g[f"r{r}_ret"].tolist()
        for t, (cand, ent) in enumerate(L.trace):
            np.testing.assert_allclose(ent, g[f"r{r}_s{t}_ent"], rtol=1e-9, atol=1e-13)
        assert mvn.rng_state() == before       # the parent's generator never moves
        L.update({int(i): float(g["rel"][i]) for i in ret})


def test_border_div():
    g = _load("borderdiv_synth150")
    L = OracleBorderDiv(g["X"], length_scale=float(g["length_scale"]), alpha=float(g["kw_alpha"]))
    L.update({int(g["query"]): 1})
    for r in range(int(g["rounds"])):
        ret = L.fetch_unlabelled(int(g["k"]))
        assert ret == g[f"r{r}_ret"].tolist()
        L.update({int(i): float(g["rel"][i]) for i in ret})
