"""Parity of the device MCMI_min (reference ital/mcmi.py) with the reference's golden vectors and with the oracle,
through the C ABI.  Run on the GPU box: python -m pytest tests -m gpu."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden  # noqa: E402  (fixture table only)

# conditional-entropy sums are O(-0.5 * N_c); the dense reference route (explicit (m+t)^2 inverse, gp.py:295-344)
# and the whitened closed form agree to ~1e-10 relative, far inside the 1e-5 bar of BASELINE.json
CE_RTOL = 1e-8


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return "cuda:0"


def test_cov_block_vs_oracle(dev):
    from ital_amd import GaussianProcess
    from ital_amd import _lib
    from ital_amd.gp import _ptr, _stream
    from oracle.gp import OracleGP
    rng = np.random.default_rng(11)
    n, d = 333, 21
    X = rng.random((n, d))
    ls = float(np.sqrt(d / 12.0))
    gp = GaussianProcess(X, ls, var=1.2, device=dev)
    ref = OracleGP(X, ls, var=1.2)
    idx = rng.choice(n, 9, replace=False).tolist()
    y = np.where(rng.random(9) > 0.5, 1.0, -1.0)
    gp.update(idx, y)
    ref.update(idx, y)
    a = np.sort(rng.choice(n, 70, replace=False))
    b = np.sort(rng.choice(n, 130, replace=False))
    ta, tb = torch.as_tensor(a, device=dev), torch.as_tensor(b, device=dev)
    Xa, Xb = gp.Xd.index_select(0, ta), gp.Xd.index_select(0, tb)
    Va = gp.V[: gp.m].index_select(1, ta).contiguous()
    Vb = gp.V[: gp.m].index_select(1, tb).contiguous()
    na, nb = gp.xnorm[ta].contiguous(), gp.xnorm[tb].contiguous()   # named: the pointers must outlive the launch
    out = torch.full((len(a), 144), -7.0, dtype=torch.float64, device=dev)
    _lib.check(_lib.lib().ital_cov_block(_ptr(Xa), _ptr(na), len(a), _ptr(Xb),
                                         _ptr(nb), len(b), gp.ldx, _ptr(Va), len(a), _ptr(Vb),
                                         len(b), gp.m, 1.2, ls, _ptr(out), 144, _stream()))
    got = out.cpu().numpy()
    want = ref.predict_stored(np.concatenate((a, b)), cov_mode="full")[1][: len(a), len(a):]
    np.testing.assert_allclose(got[:, : len(b)], want, rtol=0, atol=2e-9)
    assert np.all(got[:, len(b):] == -7.0)          # padding columns untouched


def test_cov_block_lds_tiles_vs_oracle(dev):
    """A block of more than 1024 tiles of 128 x 128 takes the LDS-staged kernel: ragged edges in both directions, checked
    against the oracle's full predictive covariance, and bit for bit against the register-tiled kernel on a sub-block
    small enough to be routed to it (same expression, same accumulation order over the features)."""
    from ital_amd import GaussianProcess
    from ital_amd import _lib
    from ital_amd.gp import _ptr, _stream
    from oracle.gp import OracleGP
    rng = np.random.default_rng(12)
    n, d = 4300, 21
    X = rng.random((n, d))
    ls = float(np.sqrt(d / 12.0))
    gp = GaussianProcess(X, ls, var=1.2, device=dev)
    ref = OracleGP(X, ls, var=1.2)
    idx = rng.choice(n, 7, replace=False).tolist()
    y = np.where(rng.random(7) > 0.5, 1.0, -1.0)
    gp.update(idx, y)
    ref.update(idx, y)
    a0, a1, b0, b1 = 0, 4210, 70, 4300            # 33 x 34 tiles
    Xa, Xb = gp.Xd[a0:a1].contiguous(), gp.Xd[b0:b1].contiguous()
    Va, Vb = gp.V[: gp.m, a0:a1].contiguous(), gp.V[: gp.m, b0:b1].contiguous()
    na, nb = gp.xnorm[a0:a1].contiguous(), gp.xnorm[b0:b1].contiguous()
    ldo = b1 - b0 + 6
    out = torch.full((a1 - a0, ldo), -7.0, dtype=torch.float64, device=dev)
    lib = _lib.lib()
    _lib.check(lib.ital_cov_block(_ptr(Xa), _ptr(na), a1 - a0, _ptr(Xb), _ptr(nb), b1 - b0, gp.ldx, _ptr(Va), a1 - a0,
                                  _ptr(Vb), b1 - b0, gp.m, 1.2, ls, _ptr(out), ldo, _stream()))
    got = out.cpu().numpy()
    want = ref.predict_stored(np.arange(n), cov_mode="full")[1][a0:a1, b0:b1]
    np.testing.assert_allclose(got[:, : b1 - b0], want, rtol=0, atol=2e-9)
    assert np.all(got[:, b1 - b0:] == -7.0)         # padding columns untouched
    sub = torch.empty((300, 500), dtype=torch.float64, device=dev)
    _lib.check(lib.ital_cov_block(_ptr(Xa[3900:]), _ptr(na[3900:]), 300, _ptr(Xb[3700:]), _ptr(nb[3700:]), 500, gp.ldx,
                                  _ptr(Va[:, 3900:]), a1 - a0, _ptr(Vb[:, 3700:]), b1 - b0, gp.m, 1.2, ls, _ptr(sub), 500,
                                  _stream()))
    assert torch.equal(sub, out[3900:4200, 3700:4200])


@pytest.mark.parametrize("name", ["usps500_mcmi", "synth300_mcmi"])
def test_golden_mcmi(dev, golden_dir, name):
    from ital_amd import MCMI_min
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    spec = make_golden.FIXTURES[name]
    np.random.seed(0)
    L = MCMI_min(z["X"], length_scale=float(z["length_scale"]), device=dev, **spec["kw"])
    L.keep_scores = True
    L.update({int(z["query"]): 1})
    rel = z["rel"]
    for r in range(int(z["rounds"])):
        ret = L.fetch_unlabelled(int(z["k"]))
        cand0 = z[f"r{r}_s0_cand"].tolist()
        pos = {c: i for i, c in enumerate(cand0)}
        for t in range(len(ret)):
            cand = z[f"r{r}_s{t}_cand"].tolist()
            mine = L.last_scores[t].cpu().numpy()[[pos[c] for c in cand]]
            np.testing.assert_allclose(mine, z[f"r{r}_s{t}_mi"], rtol=CE_RTOL, atol=0, err_msg=f"{name} r{r} step {t}")
        assert ret == z[f"r{r}_ret"].tolist(), (name, r)          # selected indices bit-exact
        assert len(L.candidates) == len(cand0) - len(ret) and not set(ret) & set(L.candidates)
        L.update({int(i): float(rel[i]) for i in ret})
    np.testing.assert_allclose(L.rel_mean, z["final_rel_mean"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("seed,n,d,k,sub", [(0, 140, 6, 4, None), (1, 260, 18, 5, 90), (2, 64, 3, 6, None)])
def test_mcmi_against_oracle(dev, seed, n, d, k, sub):
    from ital_amd import MCMI_min
    from oracle.ital import OracleMCMI
    rng = np.random.default_rng(seed)
    X = rng.random((n, d))
    ls = float(np.sqrt(d / 12.0))
    labels = {int(i): (1 if X[i, 0] > 0.5 else -1) for i in rng.choice(n, 4, replace=False)}
    A = MCMI_min(X, length_scale=ls, subsample=sub, device=dev)
    A.keep_scores = True
    B = OracleMCMI(X, length_scale=ls, subsample=sub)
    A.update(labels)
    B.update(labels)
    for _ in range(2):
        np.random.seed(seed)
        got = A.fetch_unlabelled(k)
        np.random.seed(seed)
        want = [int(i) for i in B.fetch_unlabelled(k)]
        cand0 = B.trace[0][0]
        pos = {c: i for i, c in enumerate(cand0)}
        for t, (cand, vals, _) in enumerate(B.trace):
            mine = A.last_scores[t].cpu().numpy()[[pos[c] for c in cand]]
            np.testing.assert_allclose(mine, vals, rtol=CE_RTOL, atol=0)
        assert got == want
        fb = {i: (1 if X[i, 0] > 0.5 else -1) for i in got}
        A.update(fb)
        B.update(fb)


@pytest.mark.parametrize("n,d,k,sub", [(90, 5, 8, None), (110, 12, 7, None), (130, 7, 6, None), (400, 16, 5, 120)])
def test_mcmi_split_scorer_against_the_oracle(dev, n, d, k, sub):
    """Batches of 5 .. 8 (three shipped configurations use batch_size = 6, reference configs/toy*.conf) run as a preparation
    kernel + one workgroup per candidate and group of label patterns (the only form since round 4: the single kernel
    needed 256 + 184 registers and scratch there).  Picks and every conditional entropy of two rounds against the oracle;
    a workspace that served a larger block before (stale data where the new call keeps its counters) changes nothing."""
    from ital_amd import MCMI_min
    from oracle.ital import OracleMCMI
    rng = np.random.default_rng(10 + k)
    X = rng.random((n, d))
    ls = float(np.sqrt(d / 12.0))
    labels = {int(i): (1 if X[i, 0] > 0.5 else -1) for i in rng.choice(n, 3, replace=False)}
    A = MCMI_min(X, length_scale=ls, subsample=sub, device=dev)
    A.keep_scores = True
    A.round_call = False              # step by step: the scores of every step are kept
    B = OracleMCMI(X, length_scale=ls, subsample=sub)
    A.update(labels)
    B.update(labels)
    for rnd in range(2):
        np.random.seed(5 + rnd)
        got = A.fetch_unlabelled(k)
        np.random.seed(5 + rnd)
        want = [int(i) for i in B.fetch_unlabelled(k)]
        assert got == want
        pos = {c: i for i, c in enumerate(B.trace[0][0])}
        for t, (cand, vals, _) in enumerate(B.trace):
            np.testing.assert_allclose(A.last_scores[t].cpu().numpy()[[pos[c] for c in cand]], vals, rtol=CE_RTOL, atol=0)
        fb = {i: (1 if X[i, 0] > 0.5 else -1) for i in got}
        A.update(fb)
        B.update(fb)
        if rnd == 0:
            # leave garbage in the workspace where a block of another size keeps its ticket counters
            A._fetch_bufs[1]["mcmi_work"].fill_(float("nan"))


def test_mcmi_batches_above_four_need_the_workspace(dev):
    """C ABI: ital_mcmi_score_step with t >= 5 and no workspace is refused (-22), not silently run another way."""
    import ctypes
    from ital_amd import _lib
    d = _lib.ItalMcmiDesc()
    d.t, d.n_i, d.n_all, d.ld_cov, d.ldc = 5, 8, 8, 8, 8
    d.batch.kmax = 8
    rc = _lib.lib().ital_mcmi_score_step(ctypes.byref(d), None)
    assert rc == -22 and b"workspace" in _lib.lib().ital_last_error()


@pytest.mark.parametrize("n,d,k,sub", [(70, 5, 4, None), (400, 12, 6, 300), (1200, 16, 3, 1000), (40, 3, 8, None)])
def test_mcmi_round_as_one_call_equals_the_steps(dev, n, d, k, sub):
    """One rank: ital_mcmi_round enqueues covariance block, scoring / arg-min steps and covariance columns in one call;
    `round_call = False` drives the same entry points step by step.  Same picks and candidate lists over three rounds
    (a duplicated row: a tie of the arg-min, first position wins), same state afterwards."""
    from ital_amd import MCMI_min
    rng = np.random.default_rng(30 + k)
    X = rng.random((n, d))
    X[n // 2] = X[n // 3]
    ls = float(np.sqrt(d / 12.0))
    labels = {int(i): (1 if X[i, 0] > 0.5 else -1) for i in rng.choice(n, 3, replace=False)}
    out = []
    for one_call in (True, False):
        L = MCMI_min(X, length_scale=ls, subsample=sub, device=dev)
        L.round_call = one_call
        L.update(labels)
        res = []
        for _ in range(3):
            np.random.seed(6)
            ret = L.fetch_unlabelled(k)
            res.append((ret, list(L.candidates)))
            L.update({i: (1 if X[i, 0] > 0.5 else -1) for i in ret})
        out.append((res, np.asarray(L.rel_mean).copy()))
    assert out[0][0] == out[1][0]
    np.testing.assert_array_equal(out[0][1], out[1][1])


def test_gather_block_equals_indexing(dev):
    """ital_gather_block: rows, whitened columns, norms, means and variances of a candidate list in one launch; samples
    outside the rank's rows contribute zeros."""
    import torch
    from ital_amd import MCMI_min, _lib
    rng = np.random.default_rng(5)
    X = rng.random((300, 20))
    L = MCMI_min(X, length_scale=1.2, device=dev)
    L.update({3: 1, 77: -1, 150: 1, 299: -1, 12: 1})
    gp = L.gp
    cand = rng.permutation(300)[:130].astype(np.int64)
    Xc, Vc, ldc, xnc, muc, s2c = L._gather_block(cand)
    idx = torch.as_tensor(cand, device=gp.device)
    torch.testing.assert_close(Xc, gp.Xd.index_select(0, idx), rtol=0, atol=0)
    torch.testing.assert_close(Vc[: gp.m, : len(cand)], gp.V[: gp.m].index_select(1, idx), rtol=0, atol=0)
    assert float(Vc[:, len(cand):].abs().max()) == 0.0
    for got, src in ((xnc, gp.xnorm), (muc, gp.mu), (s2c, gp.s2)):
        torch.testing.assert_close(got, src[idx], rtol=0, atol=0)
    # a rank that owns rows 100 .. 199 only
    nc = len(cand)
    out = [torch.full((nc, gp.ldx), 7.0, dtype=torch.float64, device=gp.device),
           torch.full((gp.cap, ldc), 7.0, dtype=torch.float64, device=gp.device)] + \
          [torch.full((nc,), 7.0, dtype=torch.float64, device=gp.device) for _ in range(3)]
    p = lambda t: t.data_ptr()
    _lib.check(_lib.lib().ital_gather_block(p(idx), nc, 100, 100, p(gp.Xd[100:]), p(gp.xnorm[100:]), gp.ldx,
                                            gp.V.data_ptr() + 8 * 100, gp.ldv, gp.m, p(gp.mu[100:]), p(gp.s2[100:]),
                                            p(out[0]), p(out[1]), ldc, p(out[2]), p(out[3]), p(out[4]),
                                            torch.cuda.current_stream().cuda_stream))
    own = torch.as_tensor((cand >= 100) & (cand < 200), device=gp.device)
    want = gp.Xd.index_select(0, idx) * own[:, None]
    torch.testing.assert_close(out[0], want, rtol=0, atol=0)
    torch.testing.assert_close(out[3], gp.mu[idx] * own, rtol=0, atol=0)
    torch.testing.assert_close(out[1][: gp.m, :nc], gp.V[: gp.m].index_select(1, idx) * own[None, :], rtol=0, atol=0)


def test_mcmi_edge_cases(dev):
    from ital_amd import MCMI_min
    rng = np.random.default_rng(3)
    X = rng.random((9, 4))
    L = MCMI_min(X, length_scale=0.6, device=dev)
    with pytest.raises(RuntimeError):
        L.fetch_unlabelled(2)
    L.update({0: 1, 1: -1})
    assert L.fetch_unlabelled(0) == []
    ret = L.fetch_unlabelled(20)                    # more than there are candidates, and more than the device batch limit
    assert sorted(ret) == list(range(2, 9)) and L.candidates == []
    big = MCMI_min(rng.random((40, 4)), length_scale=0.6, device=dev)
    big.update({0: 1})
    with pytest.raises(NotImplementedError):
        big.fetch_unlabelled(9)
