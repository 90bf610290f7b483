"""Parity of the GP-sharing comparison learners (EMOC, entropy, border_div) with the golden
vectors of the real reference and with the oracle, through the C ABI.  Run on the GPU box: python -m pytest tests -m gpu."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return "cuda:0"


def _load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


@pytest.mark.parametrize("na,nb,d,m", [(1, 1, 3, 0), (70, 130, 21, 9), (257, 1000, 16, 5), (129, 63, 40, 17),
                                       (4200, 4100, 24, 6), (4097, 4225, 40, 0)])
def test_cov_abs_rowsum_vs_cov_block(dev, na, nb, d, m):
    """The fused row sums equal the sums over the materialised block, with and without accumulation.  In the two large
    cases (more than 1024 tiles of 128 x 128) the block comes from the LDS-staged kernel, the sums from the register-tiled
    one: same accumulation order over the features, so the comparison is as tight as for the small ones."""
    from ital_amd import _lib
    from ital_amd.gp import _pad16, _ptr, _stream
    lib = _lib.lib()
    rng = np.random.default_rng(na + nb)
    ldx = _pad16(d)
    Xa = torch.zeros((na, ldx), dtype=torch.float64, device=dev)
    Xb = torch.zeros((nb, ldx), dtype=torch.float64, device=dev)
    Xa[:, :d] = torch.from_numpy(rng.random((na, d))).to(dev)
    Xb[:, :d] = torch.from_numpy(rng.random((nb, d))).to(dev)
    an, bn = (Xa * Xa).sum(1).contiguous(), (Xb * Xb).sum(1).contiguous()
    Va = torch.from_numpy(rng.normal(size=(max(m, 1), na)) * 0.2).to(dev).contiguous()
    Vb = torch.from_numpy(rng.normal(size=(max(m, 1), nb)) * 0.2).to(dev).contiguous()
    ls = float(np.sqrt(d / 12.0))
    ldo = _pad16(nb)
    block = torch.zeros((na, ldo), dtype=torch.float64, device=dev)
    _lib.check(lib.ital_cov_block(_ptr(Xa), _ptr(an), na, _ptr(Xb), _ptr(bn), nb, ldx, _ptr(Va), na, _ptr(Vb), nb, m,
                                  1.3, ls, _ptr(block), ldo, _stream()))
    want = block[:, :nb].abs().sum(1).cpu().numpy()
    for work_rows in (1, 7, 64):                       # number of column splits the work area allows
        work = torch.empty(na * work_rows, dtype=torch.float64, device=dev)
        out = torch.full((na,), 5.0, dtype=torch.float64, device=dev)
        _lib.check(lib.ital_cov_abs_rowsum(_ptr(Xa), _ptr(an), na, _ptr(Xb), _ptr(bn), nb, ldx, _ptr(Va), na, _ptr(Vb),
                                           nb, m, 1.3, ls, _ptr(work), work.numel(), 0, _ptr(out), _stream()))
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-13)
        _lib.check(lib.ital_cov_abs_rowsum(_ptr(Xa), _ptr(an), na, _ptr(Xb), _ptr(bn), nb, ldx, _ptr(Va), na, _ptr(Vb),
                                           nb, m, 1.3, ls, _ptr(work), work.numel(), 1, _ptr(out), _stream()))
        np.testing.assert_allclose(out.cpu().numpy(), 2 * want, rtol=1e-13)
    # two identical launches give identical bits (fixed summation order, no atomics)
    a = torch.empty(na, dtype=torch.float64, device=dev)
    b = torch.empty(na, dtype=torch.float64, device=dev)
    for o in (a, b):
        _lib.check(lib.ital_cov_abs_rowsum(_ptr(Xa), _ptr(an), na, _ptr(Xb), _ptr(bn), nb, ldx, _ptr(Va), na, _ptr(Vb),
                                           nb, m, 1.3, ls, _ptr(work), work.numel(), 0, _ptr(o), _stream()))
    assert torch.equal(a, b)
    assert lib.ital_cov_abs_rowsum(_ptr(Xa), _ptr(an), na, _ptr(Xb), _ptr(bn), nb, ldx, _ptr(Va), na, _ptr(Vb), nb, m,
                                   1.3, ls, _ptr(work), na - 1, 0, _ptr(a), _stream()) != 0     # work area too small


@pytest.mark.parametrize("name", ["emoc_synth150", "emoc_usps500"])
def test_emoc_golden(dev, name):
    from ital_amd.baselines import EMOC
    g = _load(name)
    L = EMOC(g["X"], length_scale=float(g["length_scale"]), device=dev)
    L.update({int(g["query"]): 1})
    for r in range(int(g["rounds"])):
        ret = L.fetch_unlabelled(int(g["k"]))
        np.testing.assert_allclose(L.last_scores, g[f"r{r}_scores"], rtol=1e-6)
        assert ret == g[f"r{r}_ret"].tolist()
        assert all(type(i) is int for i in ret)
        L.update({int(i): float(g["rel"][i]) for i in ret})
    np.testing.assert_allclose(L.rel_mean, g["final_rel_mean"], rtol=0, atol=1e-9)


def test_emoc_vs_oracle_with_queries_and_unnameable(dev):
    from ital_amd.baselines import EMOC
    from oracle.baselines import OracleEMOC
    rng = np.random.default_rng(5)
    X = rng.random((90, 7))
    q = [rng.random(7), rng.random(7)]
    A = EMOC(X, q, length_scale=0.7, var=1.4, noise=1e-4, device=dev)
    B = OracleEMOC(X, q, length_scale=0.7, var=1.4, noise=1e-4)
    fb = {3: 1, 9: -1, 11: 0, 40: 1}
    A.update(fb)
    B.update(fb)
    for k in (5, 200):                                  # more than there are candidates: all of them, ranked
        got, want = A.fetch_unlabelled(k), B.fetch_unlabelled(k)
        np.testing.assert_allclose(A.last_scores, B.last_scores, rtol=1e-7)
        assert got == want
    assert 11 not in got and len(got) == 86


@pytest.mark.parametrize("name", ["entropy_synth80", "entropy_usps300"])
def test_entropy_golden(dev, name):
    from ital_amd import mvn_stream
    from ital_amd.baselines import EntropySampling
    g = _load(name)
    mvn_stream.GLOBAL.reset()
    L = EntropySampling(g["X"], length_scale=float(g["length_scale"]), device=dev)
    L.keep_scores = True
    L.update({int(g["query"]): 1})
    for r in range(int(g["rounds"])):
        ret = L.fetch_unlabelled(int(g["k"]))
        cand = g[f"r{r}_cand"].tolist()
        pos = {c: i for i, c in enumerate(cand)}
        for t in range(len(ret)):
            live = [c for c in cand if c not in ret[:t]]
            mine = L.last_scores[t].cpu().numpy()[[pos[c] for c in live]]
            # t <= 1 closed forms; t >= 2: the same lattice rule on the same random stream as the reference's worker
            np.testing.assert_allclose(mine, g[f"r{r}_s{t}_ent"], rtol=1e-5, atol=1e-12)
        assert ret == g[f"r{r}_ret"].tolist()
        assert mvn_stream.GLOBAL.draws == 0             # the process-wide stream stays where it was
        L.update({int(i): float(g["rel"][i]) for i in ret})


def test_entropy_vs_oracle_more_than_candidates(dev):
    from ital_amd import mvn_stream
    from ital_amd.baselines import EntropySampling
    from oracle import mvn as omvn
    from oracle.baselines import OracleEntropy
    rng = np.random.default_rng(8)
    X = rng.random((9, 3))
    mvn_stream.GLOBAL.reset()
    omvn.rng_reset()
    A = EntropySampling(X, length_scale=0.5, device=dev)
    B = OracleEntropy(X, length_scale=0.5)
    fb = {0: 1, 4: -1, 5: 1, 8: -1}
    A.update(fb)
    B.update(fb)
    assert A.fetch_unlabelled(7) == B.fetch_unlabelled(7)          # five candidates, five-dimensional orthants at the end
    A.update({1: 1, 2: 1, 3: 1, 6: 1, 7: 1})
    with pytest.raises(ValueError):
        A.fetch_unlabelled(2)                                       # max() of an empty sequence in the reference


def test_border_div_golden(dev):
    from ital_amd.baselines import BorderlineDiversitySampling
    g = _load("borderdiv_synth150")
    L = BorderlineDiversitySampling(g["X"], length_scale=float(g["length_scale"]), alpha=float(g["kw_alpha"]), device=dev)
    L.update({int(g["query"]): 1})
    for r in range(int(g["rounds"])):
        ret = L.fetch_unlabelled(int(g["k"]))
        assert ret == g[f"r{r}_ret"].tolist()
        L.update({int(i): float(g["rel"][i]) for i in ret})
