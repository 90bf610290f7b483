"""Parity of the HIP path (through the C ABI) with the oracle and with the reference's golden vectors.
Run on the GPU box: python -m pytest tests -m gpu."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden  # noqa: E402  (fixture table only)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return "cuda:0"


def _learners():
    from ital_amd import ITAL, GaussianProcess, mvn_stream
    return ITAL, GaussianProcess, mvn_stream


# ------------------------------------------------------------------------------------------ kernels vs oracle
@pytest.mark.parametrize("n,d,c", [(1, 4, 1), (63, 16, 3), (64, 17, 16), (1000, 50, 5), (4097, 256, 16), (333, 100, 1)])
def test_rbf_cols(dev, n, d, c):
    from oracle.gp import rbf_kernel
    _, GP, _ = _learners()
    rng = np.random.default_rng(n + d)
    X = rng.random((n, d))
    gp = GP(X, 0.9 * np.sqrt(d / 12.0), var=1.3, device=dev)
    idx = rng.integers(0, n, size=c).tolist()
    got = gp.rbf_cols(idx).cpu().numpy()
    want = rbf_kernel(X[idx], X, gp.length_scale, 1.3)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-15)


def test_gp_update_predict_vs_oracle(dev):
    from oracle.gp import OracleGP
    _, GP, _ = _learners()
    rng = np.random.default_rng(5)
    X = rng.random((700, 33))
    ls = 0.8 * np.sqrt(33 / 12.0)
    gp = GP(X, ls, device=dev, capacity=16)
    ref = OracleGP(X, ls)
    perm = rng.permutation(700)
    at = 0
    for c in (1, 4, 17, 3, 40, 70):  # crosses the 16-row chunking, the capacity growth and two 64-column blocks of the factor
        idx = perm[at:at + c].tolist()
        y = np.where(rng.random(c) > 0.5, 1.0, -1.0)
        at += c
        gp.update(idx, y)
        ref.update(idx, y)
        m, v = gp.predict_stored(cov_mode="diag")
        mr, vr = ref.predict_stored(cov_mode="diag")
        np.testing.assert_allclose(m, mr, rtol=0, atol=2e-9)
        np.testing.assert_allclose(v, vr, rtol=0, atol=2e-9)
    gp.check_status()
    assert gp.ind == ref.ind
    Xt = rng.random((37, 33))
    pm, pv = gp.predict(Xt, cov_mode="diag")
    pmr, pvr = ref.predict(Xt, cov_mode="diag")
    np.testing.assert_allclose(pm, pmr, rtol=0, atol=2e-9)
    np.testing.assert_allclose(pv, pvr, rtol=0, atol=2e-9)
    sub = perm[500:507].tolist()
    mf, cf = gp.predict_stored(sub, cov_mode="full")
    mfr, cfr = ref.predict_stored(sub, cov_mode="full")
    np.testing.assert_allclose(cf, cfr, rtol=0, atol=2e-9)


# ------------------------------------------------------------------------------------------ golden vectors
def _run_fixture(dev, golden_dir, name, rounds=None, mi_atol=1e-10, force_generic=False):
    ITAL, _, mvn_stream = _learners()
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    spec = make_golden.FIXTURES[name]
    mvn_stream.GLOBAL.reset()
    np.random.seed(0)          # as the fixture generator (the change-estimation subset is drawn from numpy's global RNG)
    L = ITAL(z["X"], length_scale=float(z["length_scale"]), device=dev, **spec["kw"])
    L.keep_scores = True
    L.force_generic = force_generic
    L.update({int(z["query"]): 1})
    rel = z["rel"]
    for r in range(int(z["rounds"]) if rounds is None else rounds):
        m, v = L.gp.predict_stored(cov_mode="diag")
        np.testing.assert_allclose(m, z[f"r{r}_rel_mean"], rtol=0, atol=1e-10)
        np.testing.assert_allclose(v, z[f"r{r}_var"], rtol=0, atol=1e-9)
        ret = L.fetch_unlabelled(int(z["k"]))
        if f"r{r}_ce_subset" in z:
            assert L._ce_subset == z[f"r{r}_ce_subset"].tolist()
        cand0 = z[f"r{r}_s0_cand"].tolist()
        pos = {c: i for i, c in enumerate(cand0)}
        for t in range(len(ret)):
            cand = z[f"r{r}_s{t}_cand"].tolist()
            mine = L.last_scores[t].cpu().numpy()[[pos[c] for c in cand]]
            np.testing.assert_allclose(mine, z[f"r{r}_s{t}_mi"], rtol=1e-8, atol=mi_atol, err_msg=f"{name} r{r} step {t}")
        assert ret == z[f"r{r}_ret"].tolist(), (name, r)   # selected indices bit-exact
        L.update({int(i): float(rel[i]) for i in ret})
    np.testing.assert_allclose(L.rel_mean, z["final_rel_mean"] if rounds is None else L.rel_mean, rtol=0, atol=1e-9)
    return L, z


@pytest.mark.parametrize("name", ["usps500", "butterflies", "synth300", "synth96_k6", "usps2007",
                                  "synth200_optimistic", "synth200_topcand", "synth200_topcand_float"])
def test_golden_fixture(dev, golden_dir, name):
    L, z = _run_fixture(dev, golden_dir, name)
    assert L.top_results(10).tolist() == z["top_results_10"].tolist()
    pm, pv = L.gp.predict(z["predict_X"], cov_mode="diag")
    np.testing.assert_allclose(pm, z["predict_mean"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(pv, z["predict_var"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("name", ["synth200_noisy", "synth200_motivated", "iris_ce5", "synth80_mcrel", "synth60_mcfb",
                                  "synth50_mcboth", "synth50_clip"])
def test_golden_fixture_general_scorer(dev, golden_dir, name):
    """Noisy user models (general / motivated), the change-estimation subset and the Monte-Carlo switches
    (patterns / feedback sampled from numpy's global RNG in the reference's order): ital_score_generic."""
    _run_fixture(dev, golden_dir, name)


@pytest.mark.parametrize("name", ["usps500", "synth96_k6", "synth200_optimistic", "synth200_topcand"])
def test_general_scorer_equals_fast_path(dev, golden_dir, name):
    """The perfect-user case through the general scorer must give the golden MI vectors and picks as well."""
    _run_fixture(dev, golden_dir, name, force_generic=True)


# ------------------------------------------------------------------------------------------ HIP vs oracle, seeded
@pytest.mark.parametrize("seed,n,d,k,mode", [(0, 150, 8, 4, "mean"), (1, 257, 20, 5, "mean"), (2, 90, 3, 3, "pessimistic"),
                                             (3, 120, 40, 4, "optimistic"),
                                             # low dimensions, larger batches: saturated orthant probabilities whose sums are formed
                                             # again in MVKBRV's order (qmc_exact_kernel<5 .. 8>, csrc/qmc_exact.h)
                                             (4, 40, 3, 6, "pessimistic"), (5, 26, 2, 8, "optimistic"), (6, 36, 2, 7, "pessimistic")])
def test_against_oracle(dev, seed, n, d, k, mode):
    from oracle import mvn as omvn
    from oracle.ital import OracleITAL
    ITAL, _, mvn_stream = _learners()
    rng = np.random.default_rng(seed)
    X = rng.random((n, d))
    ls = float(np.sqrt(d / 12.0))
    labels = {int(i): (1 if X[i, 0] > 0.5 else -1) for i in rng.choice(n, 5, replace=False)}
    mvn_stream.GLOBAL.reset()
    omvn.rng_reset()
    A = ITAL(X, length_scale=ls, label_estimation=mode, device=dev)
    A.keep_scores = True
    B = OracleITAL(X, length_scale=ls, label_estimation=mode)
    A.update(labels)
    B.update(labels)
    for _ in range(2):
        got = A.fetch_unlabelled(k)
        want = [int(i) for i in B.fetch_unlabelled(k)]
        cand0 = B.trace[0][0]
        pos = {c: i for i, c in enumerate(cand0)}
        for t, (cand, vals, _) in enumerate(B.trace):
            mine = A.last_scores[t].cpu().numpy()[[pos[c] for c in cand]]
            np.testing.assert_allclose(mine, vals, rtol=1e-8, atol=1e-10)
        assert got == want
        fb = {i: (1 if X[i, 0] > 0.5 else -1) for i in got}
        A.update(fb)
        B.update(fb)
    assert mvn_stream.GLOBAL.draws == omvn.rng_draws()   # the replayed stream stands where the serial reference's does


@pytest.mark.parametrize("seed,n,d,k,kw", [
    (0, 90, 6, 3, dict(label_prob=0.75, mistake_prob=0.05)),
    (1, 70, 10, 4, dict(label_prob=0.5, mistake_prob=0.25)),
    (2, 80, 5, 4, dict(label_prob=1.0, mistake_prob=0.2)),
    (3, 60, 4, 3, dict(change_estimation_subset=4)),
    (4, 50, 4, 3, dict(change_estimation_subset=3, label_prob=0.8, mistake_prob=0.1)),
    (5, 64, 7, 3, dict(label_prob=0.6, mistake_prob=0.1, label_estimation="optimistic")),
    (6, 40, 3, 5, dict(change_estimation_subset=6)),
    (7, 60, 5, 6, dict(monte_carlo_num_rel=1)),
    (8, 40, 4, 4, dict(label_prob=0.6, mistake_prob=0.2, monte_carlo_num_rel=2, monte_carlo_num_fb=2)),
    (9, 40, 4, 4, dict(change_estimation_subset=3, monte_carlo_num_rel=1)),
    (10, 26, 3, 15, dict(monte_carlo_num_rel=1)),          # orthant dimensions up to 15 (lattice prime 1361)
    (11, 24, 3, 5, dict(change_estimation_subset=12)),     # subset + batch = 17 dimensions
    (14, 14, 3, 3, dict(change_estimation_subset=None)),   # the whole (small) candidate set as estimation subset
    (12, 36, 3, 7, dict(clip_cov=0.4)),                    # grouped orthant probabilities from the sixth pick on
    (13, 30, 3, 3, dict(change_estimation_subset=6, clip_cov=0.25, label_prob=0.7, mistake_prob=0.1)),
    # noisy users with the estimates that compare terms for exact equality, low dimension (saturated probabilities: the
    # pipeline's gen_exact_kernel forms those sums again in the reference's order)
    (15, 24, 2, 4, dict(label_prob=0.7, mistake_prob=0.1, label_estimation="pessimistic")),
    (16, 24, 3, 5, dict(mistake_prob=0.2, label_estimation="optimistic")),
])
def test_general_scorer_against_oracle(dev, seed, n, d, k, kw):
    from oracle import mvn as omvn
    from oracle.ital import OracleITAL
    ITAL, _, mvn_stream = _learners()
    rng = np.random.default_rng(seed)
    X = rng.random((n, d))
    ls = float(np.sqrt(d / 12.0))
    labels = {int(i): (1 if X[i, 0] > 0.5 else -1) for i in rng.choice(n, 4, replace=False)}
    mvn_stream.GLOBAL.reset()
    omvn.rng_reset()
    A = ITAL(X, length_scale=ls, device=dev, **kw)
    A.keep_scores = True
    B = OracleITAL(X, length_scale=ls, **kw)
    A.update(labels)
    B.update(labels)
    for rnd in range(2):
        np.random.seed(100 + rnd)
        got = A.fetch_unlabelled(k)
        np.random.seed(100 + rnd)
        want = [int(i) for i in B.fetch_unlabelled(k)]
        cand0 = B.trace[0][0]
        pos = {c: i for i, c in enumerate(cand0)}
        for t, (cand, vals, _) in enumerate(B.trace):
            mine = A.last_scores[t].cpu().numpy()[[pos[c] for c in cand]]
            np.testing.assert_allclose(mine, vals, rtol=1e-7, atol=1e-10, err_msg=f"round {rnd} step {t}")
        assert got == want
        fb = {i: (1 if X[i, 0] > 0.5 else -1) for i in got}
        A.update(fb)
        B.update(fb)
    assert mvn_stream.GLOBAL.draws == omvn.rng_draws()


def test_batch_of_eight_against_oracle(dev):
    """Full enumeration up to ITAL_MAX_T = 8 (512 orthant calls of dimension 8 per candidate at the last step)."""
    from oracle import mvn as omvn
    from oracle.ital import OracleITAL
    ITAL, _, mvn_stream = _learners()
    rng = np.random.default_rng(21)
    X = rng.random((13, 3))
    mvn_stream.GLOBAL.reset()
    omvn.rng_reset()
    A = ITAL(X, length_scale=0.5, device=dev)
    A.keep_scores = True
    B = OracleITAL(X, length_scale=0.5)
    A.update({0: 1, 1: -1})
    B.update({0: 1, 1: -1})
    got = A.fetch_unlabelled(8)
    want = [int(i) for i in B.fetch_unlabelled(8)]
    pos = {c: i for i, c in enumerate(B.trace[0][0])}
    for t, (cand, vals, _) in enumerate(B.trace):
        mine = A.last_scores[t].cpu().numpy()[[pos[c] for c in cand]]
        np.testing.assert_allclose(mine, vals, rtol=1e-8, atol=1e-10, err_msg=f"step {t}")
    assert got == want
    assert mvn_stream.GLOBAL.draws == omvn.rng_draws()


def test_duplicates_of_labelled_rows_nan_semantics(dev):
    """Exact duplicates of labelled samples: zero predictive variance -> norm.cdf(0, mu, 0) = NaN, and a NaN wins
    np.argmax (reference ital.py:130, :367).  Duplicates inside a batch exercise the
    linearly-dependent branch of MVNDST."""
    from oracle import mvn as omvn
    from oracle.ital import OracleITAL
    ITAL, _, mvn_stream = _learners()
    rng = np.random.default_rng(33)
    X = rng.random((40, 4))
    X[7] = X[3]          # 3 gets labelled: 7 is an exact duplicate of a labelled row
    X[20] = X[11]        # two identical unlabelled rows: both can enter one batch
    for noise in (1e-6, 1e-2):
        mvn_stream.GLOBAL.reset()
        omvn.rng_reset()
        A = ITAL(X, length_scale=0.8, noise=noise, device=dev)
        A.keep_scores = True
        B = OracleITAL(X, length_scale=0.8, noise=noise)
        A.update({3: 1, 30: -1})
        B.update({3: 1, 30: -1})
        got = A.fetch_unlabelled(4)
        want = [int(i) for i in B.fetch_unlabelled(4)]
        pos = {c: i for i, c in enumerate(B.trace[0][0])}
        for t, (cand, vals, _) in enumerate(B.trace):
            mine = A.last_scores[t].cpu().numpy()[[pos[c] for c in cand]]
            assert np.array_equal(np.isnan(mine), np.isnan(vals)), f"noise {noise} step {t}"
            ok = ~np.isnan(vals)
            np.testing.assert_allclose(mine[ok], vals[ok], rtol=1e-6, atol=1e-9, err_msg=f"noise {noise} step {t}")
        assert got == want, (noise, got, want)


# ------------------------------------------------------------------------------------------ API behaviour / edge cases
def test_api_edge_cases(dev):
    ITAL, _, mvn_stream = _learners()
    rng = np.random.default_rng(9)
    X = rng.random((12, 6))
    L = ITAL(X, length_scale=0.7, device=dev)
    with pytest.raises(RuntimeError):
        L.fetch_unlabelled(2)                      # not fitted yet
    L.update({0: 1, 1: -1, 2: 0})                  # 2 is "unnameable": never trained on, never a candidate again
    assert L.rounds == 1 and L.relevant_ids == {0} and L.irrelevant_ids == {1} and L.unnameable_ids == {2}
    assert L.gp.ind == [0, 1]
    assert L.fetch_unlabelled(0) == []
    ret = L.fetch_unlabelled(3)
    assert len(ret) == 3 and len(set(ret)) == 3 and not set(ret) & {0, 1, 2}
    assert all(isinstance(i, int) for i in ret)
    with pytest.raises(RuntimeError, match="Cannot change feedback"):
        L.update({0: -1})
    L.update({0: 1})                                # repeating a label is ignored
    assert L.rounds == 1
    L.update({i: 1 for i in ret})
    rest = L.fetch_unlabelled(8)                    # only 6 unseen samples are left
    assert len(rest) == 6 and sorted(rest + ret + [0, 1, 2]) == list(range(12))
    assert L.top_results(3).tolist() == np.argsort(L.rel_mean)[::-1][:3].tolist()
    T = ITAL(X, length_scale=0.7, top_candidates=2, device=dev)
    T.update({0: 1})
    assert len(T.fetch_unlabelled(2)) == 2
    with pytest.raises(ValueError, match="empty sequence"):     # more picks than top_candidates allows: np.argmax([])
        T.fetch_unlabelled(3)
    T3 = ITAL(X, length_scale=0.7, top_candidates=3, device=dev)
    T3.update({0: 1})
    mvn_stream.GLOBAL.reset()
    with pytest.raises(ValueError, match="empty sequence"):
        T3.fetch_unlabelled(4)
    # the reference gets through three greedy steps first: one candidate is left at t = 3, 2 * 2^3 calls of 24 uniforms
    assert mvn_stream.GLOBAL.draws == 16 * mvn_stream.draws_per_call(3)
    L.reset()
    assert L.rounds == 0 and L.gp.m == 0 and L.rel_mean is None
    with pytest.raises(NotImplementedError):
        M = ITAL(rng.random((60, 6)), length_scale=0.7, change_estimation_subset=None, device=dev)   # 59-dimensional orthants
        M.update({0: 1})
        M.fetch_unlabelled(2)



def test_gp_attributes_of_the_reference(dev):
    """K, K_inv, w and kernel() of reference ital/gp.py, rebuilt from the device state."""
    from oracle.gp import OracleGP, rbf_kernel
    _, GP, _ = _learners()
    rng = np.random.default_rng(12)
    X = rng.random((50, 6))
    gp = GP(X, 0.9, var=1.1, noise=1e-4, device=dev)
    ref = OracleGP(X, 0.9, var=1.1, noise=1e-4)
    assert gp.K is None and gp.w is None
    idx, y = [4, 9, 17, 30], [1.0, -1.0, 1.0, -1.0]
    gp.update(idx, y)
    ref.update(idx, y)
    np.testing.assert_allclose(gp.K, ref.K, rtol=0, atol=1e-12)
    np.testing.assert_allclose(gp.K_inv, ref.K_inv, rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(gp.w, ref.w, rtol=1e-7, atol=1e-8)
    Z = rng.random((7, 6))
    np.testing.assert_allclose(gp.kernel(Z), rbf_kernel(X[idx], Z, 0.9, 1.1), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(gp.kernel(Z, X[:5]), rbf_kernel(Z, X[:5], 0.9, 1.1), rtol=1e-12, atol=1e-15)


def test_updated_prediction_api(dev):
    from oracle.ital import OracleITAL
    ITAL, _, _ = _learners()
    rng = np.random.default_rng(8)
    X = rng.random((70, 5))
    A = ITAL(X, length_scale=0.7, device=dev)
    B = OracleITAL(X, length_scale=0.7)
    A.update({1: 1, 2: -1, 3: 1})
    B.update({1: 1, 2: -1, 3: 1})
    fb = {10: 1, 20: -1, 30: 0, 40: 1}
    test = [5, 10, 33, 40, 60]
    for mode in (None, "diag", "full"):
        got, want = A.updated_prediction(fb, test, cov_mode=mode), B.updated_prediction(fb, test, cov_mode=mode)
        if mode is None:
            np.testing.assert_allclose(got, want, rtol=0, atol=1e-8)
        else:
            np.testing.assert_allclose(got[0], want[0], rtol=0, atol=1e-8)
            np.testing.assert_allclose(got[1], want[1], rtol=0, atol=1e-8)
    m0 = A.updated_prediction({30: 0}, test, cov_mode=None)          # nothing labelled: the stored prediction
    np.testing.assert_allclose(m0, A.rel_mean[test], rtol=0, atol=0)
    assert A.gp.m == 3                                                # the model itself is untouched


def test_small_and_degenerate_candidate_sets(dev):
    """One candidate, k above the number of candidates, every scorer path; nothing left to fetch."""
    from oracle import mvn as omvn
    from oracle.ital import OracleITAL
    ITAL, _, mvn_stream = _learners()
    rng = np.random.default_rng(17)
    X = rng.random((7, 3))
    for kw in (dict(), dict(label_prob=0.7, mistake_prob=0.2), dict(change_estimation_subset=2), dict(monte_carlo_num_rel=1)):
        mvn_stream.GLOBAL.reset()
        omvn.rng_reset()
        A = ITAL(X, length_scale=0.6, device=dev, **kw)
        B = OracleITAL(X, length_scale=0.6, **kw)
        lab = {0: 1, 1: -1, 2: 1}
        A.update(lab)
        B.update(lab)
        np.random.seed(5)
        got = A.fetch_unlabelled(6)                      # 4 candidates only
        np.random.seed(5)
        want = [int(i) for i in B.fetch_unlabelled(6)]
        assert got == want and sorted(got) == [3, 4, 5, 6], kw
        A.update({i: 1 for i in got[:3]})
        B.update({i: 1 for i in got[:3]})
        assert A.fetch_unlabelled(3) == [int(i) for i in B.fetch_unlabelled(3)] == [got[3]]      # a single candidate
        A.update({got[3]: -1})
        assert A.fetch_unlabelled(2) == [] and A.get_unseen() == []
        assert mvn_stream.GLOBAL.draws == omvn.rng_draws()


def test_queries_constructor(dev):
    from oracle.ital import OracleITAL
    ITAL, _, _ = _learners()
    rng = np.random.default_rng(4)
    X = rng.random((80, 7))
    Q = rng.random((2, 7))
    A = ITAL(X, queries=Q, length_scale=0.8, device=dev)
    B = OracleITAL(X, queries=Q, length_scale=0.8)
    np.testing.assert_allclose(A.rel_mean, B.rel_mean, rtol=0, atol=1e-10)
    assert A.fetch_unlabelled(2) == [int(i) for i in B.fetch_unlabelled(2)]


# ------------------------------------------------------------------------------------------ full-size properties
def test_full_size_properties(dev):
    """BASELINE config 2 shape (9298 x 256, k = 4): determinism (bitwise), no seen ids, row-permutation equivariance of the
    closed-form steps, and the predictive means / variances of 64 sampled rows against a dense solve.  The MI VALUES of this
    size are compared with the oracle in tests/test_gpu_scale.py::test_c2_usps_shaped_9298x256_k4 (sub-problem oracle, 64
    sampled candidates + the winner per greedy step, two rounds)."""
    from oracle.gp import rbf_kernel
    ITAL, _, mvn_stream = _learners()
    rng = np.random.default_rng(0)
    n, d = 9298, 256
    X = rng.random((n, d))
    labels = {0: 1, 17: -1, 4000: 1, 9000: -1}

    def run(Xm, lab, k):
        mvn_stream.GLOBAL.reset()
        L = ITAL(Xm, length_scale=3.0, device=dev)
        L.keep_scores = True
        L.update(lab)
        return L, L.fetch_unlabelled(k)

    L1, r1 = run(X, labels, 4)
    L2, r2 = run(X, labels, 4)
    assert r1 == r2 and not set(r1) & set(labels)
    for a, b in zip(L1.last_scores, L2.last_scores):
        assert torch.equal(a, b)                                   # bitwise reproducible
    perm = rng.permutation(n)
    inv = np.argsort(perm)
    Lp, rp = run(X[perm], {int(inv[i]): y for i, y in labels.items()}, 2)
    assert [int(perm[i]) for i in rp] == r1[:2]                    # t <= 2 is closed form: equivariant under row permutation
    # scores of step 1 against the dense formula on a sample
    ids = rng.choice(np.setdiff1d(np.arange(n), list(labels)), 64, replace=False)
    T = list(labels)
    K = rbf_kernel(X[T], X[T], 3.0, 1.0) + 1e-6 * np.eye(len(T))
    kt = rbf_kernel(X[T], X[ids], 3.0, 1.0)
    mu = np.linalg.solve(K, np.array(list(labels.values()), dtype=float)) @ kt
    var = 1.0 - np.sum(kt * np.linalg.solve(K, kt), axis=0)
    m, v = L1.gp.predict_stored(cov_mode="diag")
    np.testing.assert_allclose(m[ids], mu, atol=1e-9)
    np.testing.assert_allclose(v[ids], var, atol=1e-9)
