"""Size-independent properties of the device path: tie-breaking, determinism across work-item
splits, stream bookkeeping, update-after-fetch invariants, MCMI determinism."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return "cuda:0"


def test_ties_resolve_to_lowest_list_position(dev):
    """Identical rows have identical scores: np.argmax takes the first (reference ital.py:130)."""
    from ital_amd import ITAL, mvn_stream
    rng = np.random.default_rng(1)
    base = rng.random((30, 5))
    X = np.concatenate([base, base[5:15]])          # rows 30..39 duplicate rows 5..14
    mvn_stream.GLOBAL.reset()
    L = ITAL(X, length_scale=0.9, device=dev)
    L.keep_scores = True
    L.update({0: 1, 1: -1})
    ret = L.fetch_unlabelled(2)
    s0 = L.last_scores[0].cpu().numpy()             # list position = data index - 2 (rows 0, 1 are labelled)
    np.testing.assert_array_equal(s0[3:13], s0[28:38])
    best = np.flatnonzero(s0 == s0.max())
    assert ret[0] == int(best[0]) + 2               # first maximum, although a duplicate scores the same
    if len(best) > 1:
        assert ret[0] < int(best[1]) + 2


@pytest.mark.parametrize("n,k", [(700, 5), (100, 5), (3300, 4)])
def test_scores_do_not_depend_on_the_workspace_slabs(dev, n, k):
    """The lattice scorer walks the candidates in slabs that fit its workspace: one slab against many (down to a
    handful of candidates per slab) must give the same bits."""
    from ital_amd import ITAL, mvn_stream
    rng = np.random.default_rng(2)
    X = rng.random((n, 12))
    out = []
    for work_bytes in (1 << 30, 1 << 22, 1 << 19):
        mvn_stream.GLOBAL.reset()
        L = ITAL(X, length_scale=1.0, device=dev)
        L.qmc_work_bytes = work_bytes
        L.keep_scores = True
        L.update({3: 1, 4: -1, 5: 1})
        ret = L.fetch_unlabelled(k)
        out.append((ret, [s.cpu().numpy() for s in L.last_scores], mvn_stream.GLOBAL.draws))
    for ret, scores, draws in out[1:]:
        assert ret == out[0][0] and draws == out[0][2]
        for a, b in zip(scores, out[0][1]):
            live = ~np.isnan(a)
            np.testing.assert_array_equal(a[live], b[live])


@pytest.mark.parametrize("n,k,work_bytes", [(2500, 4, 1 << 30), (300, 6, 1 << 30), (3300, 4, 1 << 20), (33, 3, 1 << 30),
                                            (1500, 7, 1 << 19)])     # t = 7 in slabs of 8 candidates: more slabs than n / 32 + 64
def test_selection_inside_the_scoring_launch_equals_the_separate_launch(dev, n, k, work_bytes):
    """One rank, small problems: the last block of a step's scoring launch selects (select_tail) instead of a launch of
    ital_select_fused.  Same picks, same batch state, same alive flags -- also with several slabs and with ties."""
    from ital_amd import ITAL, mvn_stream
    rng = np.random.default_rng(4)
    X = rng.random((n, 10))
    X[n // 2] = X[n // 3]                      # an exact twin: equal scores, the lower list position has to win
    out = []
    for mode in ("round", "steps", "separate"):
        # round: the whole round through ital_fetch_round (candidate list kept / compacted on the device); steps: the same
        # kernels launched step by step from Python; separate: the selection as a launch of its own (ital_select_fused)
        mvn_stream.GLOBAL.reset()
        L = ITAL(X, length_scale=0.9, device=dev)
        L.round_call = mode == "round"
        L.select_in_scorer = mode != "separate"
        L.qmc_work_bytes = work_bytes
        L.keep_scores = True
        L.update({0: 1, 1: -1})
        picks, live_scores = [], []
        for rnd in range(3):
            unseen = np.asarray(L._unseen_array())
            ret = L.fetch_unlabelled(k)
            picks.append(ret)
            b = L._fetch_bufs
            state = {q: b[q].cpu().numpy().copy() for q in ("bidx", "bgpos", "bsort", "bmu", "sig", "XB", "XBn", "VB", "ret")}
            state["alive"] = b["alive"][: len(unseen)].cpu().numpy().copy()
            for t, s_ in enumerate(L.last_scores):          # scores of the positions still alive at step t
                s_ = s_.cpu().numpy()[: len(unseen)].copy()
                s_[np.searchsorted(unseen, ret[:t])] = 0.0
                live_scores.append(s_)
            L.update({i: 1.0 if X[i, 0] > 0.5 else -1.0 for i in ret})
        assert (L._dev_list is not None) == (mode == "round")
        out.append((picks, state, mvn_stream.GLOBAL.draws, live_scores))
    for other in out[1:]:
        assert out[0][0] == other[0] and out[0][2] == other[2]
        for key, a in out[0][1].items():
            np.testing.assert_array_equal(a, other[1][key], err_msg=key)
        for x, y in zip(out[0][3], other[3]):
            np.testing.assert_array_equal(x, y)


def test_device_candidate_list_follows_arbitrary_feedback(dev):
    """The candidate list kept on the device between rounds is reused only when the host's list is the previous one minus
    exactly the previous batch; partial feedback, feedback for other samples, unnameable feedback and reset() must fall back
    to a fresh upload -- same picks as the step-by-step path either way."""
    from ital_amd import ITAL, mvn_stream
    rng = np.random.default_rng(6)
    X = rng.random((400, 8))
    lab = lambda i: 1.0 if X[i, 0] > 0.5 else -1.0   # noqa: E731
    out = []
    for rounds_in_c in (True, False):
        mvn_stream.GLOBAL.reset()
        L = ITAL(X, length_scale=0.8, device=dev)
        L.round_call = rounds_in_c
        L.update({5: 1, 9: -1})
        log, reused = [], []
        def fetch(k):                                  # noqa: E306
            ret = L.fetch_unlabelled(k)
            log.append(ret)
            reused.append(bool(L.last_round[0] == 2) if rounds_in_c else None)
            return ret
        r = fetch(4); L.update({i: lab(i) for i in r})                   # the loop of the reference: whole batch labelled
        r = fetch(4); L.update({i: lab(i) for i in r[:3]})               # one pick left without feedback: candidate again
        r = fetch(3); fb = {i: lab(i) for i in r}; fb[17] = 1; L.update(fb)   # feedback for a sample outside the batch
        r = fetch(4); L.update({r[0]: 0, r[1]: lab(r[1]), r[2]: lab(r[2]), r[3]: 0})   # unnameable: seen, not trained on
        r = fetch(4); L.update({i: lab(i) for i in r})
        r = fetch(2); L.update({i: lab(i) for i in r})
        r = fetch(4)
        L.reset(); mvn_stream.GLOBAL.reset(); L.update({5: 1, 9: -1})
        r = fetch(4); L.update({i: lab(i) for i in r})
        r = fetch(4)
        out.append((log, reused, mvn_stream.GLOBAL.draws))
    assert out[0][0] == out[1][0] and out[0][2] == out[1][2]
    #                 first  whole  partial  extra  unnameable  whole  whole  (no update)  reset  whole
    assert out[0][1] == [False, True, False, False, True, True, True, False, True]


def test_stream_position_is_a_function_of_the_work_done(dev):
    """Uniforms consumed by a fetch = sum over steps of live candidates * 2 * 2^t * draws per call."""
    from ital_amd import ITAL, mvn_stream
    rng = np.random.default_rng(3)
    X = rng.random((90, 6))
    mvn_stream.GLOBAL.reset()
    L = ITAL(X, length_scale=0.8, device=dev)
    L.update({0: 1})
    L.fetch_unlabelled(5)
    n = 89
    want = sum((n - (t - 1)) * (2 << t) * mvn_stream.draws_per_call(t) for t in range(1, 6))
    assert mvn_stream.GLOBAL.draws == want
    before = mvn_stream.GLOBAL.draws
    L.fetch_unlabelled(2)                            # closed forms draw nothing
    assert mvn_stream.GLOBAL.draws == before


def test_fetch_never_returns_seen_samples_and_update_keeps_state_consistent(dev):
    from ital_amd import ITAL
    rng = np.random.default_rng(4)
    X = rng.random((60, 4))
    L = ITAL(X, length_scale=0.7, device=dev)
    L.update({10: 1})
    seen = {10}
    for r in range(6):
        ret = L.fetch_unlabelled(3)
        assert len(set(ret)) == 3 and not set(ret) & seen
        fb = {i: (1 if X[i, 0] > 0.5 else (-1 if r % 2 else 0)) for i in ret}     # some "unnameable" feedback
        L.update(fb)
        seen |= set(ret)
        assert L.relevant_ids | L.irrelevant_ids | L.unnameable_ids == seen
        assert L.gp.ind == [10] + [i for i in L.gp.ind[1:]] and len(L.gp.ind) == L.gp.m
        m, v = L.gp.predict_stored(cov_mode="diag")
        assert np.all(v >= 0) and np.all(np.isfinite(m))
        lab = np.array(L.gp.ind)
        np.testing.assert_allclose(m[lab], L.gp.y, atol=1e-4)      # the GP interpolates its labels (noise 1e-6)


def test_mcmi_is_deterministic_and_permutation_equivariant(dev):
    from ital_amd import MCMI_min
    rng = np.random.default_rng(5)
    X = rng.random((120, 7))
    perm = rng.permutation(120)
    inv = np.argsort(perm)

    def run(Xm, q):
        L = MCMI_min(Xm, length_scale=0.9, device=dev)
        L.keep_scores = True
        L.update({int(q): 1})
        return L.fetch_unlabelled(3), [s.cpu().numpy() for s in L.last_scores]

    r1, s1 = run(X, 17)
    r2, s2 = run(X, 17)
    assert r1 == r2 and all(np.array_equal(a, b, equal_nan=True) for a, b in zip(s1, s2))
    rp, _ = run(X[perm], inv[17])
    assert [int(perm[i]) for i in rp] == r1


def test_one_million_candidates(dev):
    """BASELINE config 5's row count on one GPU (n = 1e6, d = 32, k = 3): 64-bit item / offset arithmetic, determinism,
    and the closed-form first step against the host formula on the device's own mean and variance."""
    from scipy.special import ndtr
    from ital_amd import ITAL, mvn_stream
    n, d = 1_000_000, 32
    X = np.random.default_rng(0).random((n, d), dtype=np.float64)
    picks = []
    for _ in range(2):
        mvn_stream.GLOBAL.reset()
        L = ITAL(X, length_scale=float(np.sqrt(d / 12.0)), device=dev)
        L.keep_scores = True
        L.update({7: 1, 500_000: -1, 999_999: 1})
        picks.append(L.fetch_unlabelled(3))
    assert picks[0] == picks[1] and len(set(picks[0])) == 3 and not set(picks[0]) & {7, 500_000, 999_999}
    mu, var = L.gp.predict_stored(cov_mode="diag")
    s0 = L.last_scores[0].cpu().numpy()                     # list position -> data index: the three labelled rows are cut out
    cand = np.setdiff1d(np.arange(n), [7, 500_000, 999_999])
    sub = np.random.default_rng(1).choice(len(cand), 2000, replace=False)
    m, v = mu[cand[sub]], var[cand[sub]]
    p_irr = ndtr(-m / np.sqrt(v))
    su = v                                                   # variance is far from the clamp on this data
    g = su / (su + 1e-6)
    mi = np.zeros(len(sub))
    for f, pr in ((-1.0, p_irr), (1.0, 1.0 - p_irr)):
        mu_u = m + g * (f - m)
        q = ndtr(-mu_u / np.sqrt(1e-6 * g))
        pu = q if f < 0 else 1.0 - q
        mi += pr * (np.log(pu + 1e-12) - np.log(pr + 1e-12))
    np.testing.assert_allclose(s0[sub], mi, rtol=1e-9, atol=1e-12)
    assert cand[int(np.argmax(np.where(np.isnan(s0), -np.inf, s0)))] == picks[0][0]


@pytest.mark.parametrize("kw,n,k", [(dict(label_prob=0.6, mistake_prob=0.2), 300, 4), (dict(mistake_prob=0.15), 200, 5),
                                    (dict(monte_carlo_num_rel=1), 400, 9), (dict(label_estimation="pessimistic", mistake_prob=0.1), 150, 3),
                                    (dict(monte_carlo_num_rel=1), 250, 15),     # 10-14 variables: three chains per lane
                                    # round 5: change-estimation subsets through the pipeline's wide form (calls of nr, |E| and
                                    # |E| + 1 variables in one step; members of E among the candidates; sampled feedback)
                                    (dict(change_estimation_subset=5), 200, 4),
                                    (dict(change_estimation_subset=3, label_prob=0.7, mistake_prob=0.2), 150, 3),
                                    (dict(change_estimation_subset=4, mistake_prob=0.2, monte_carlo_num_fb=2), 120, 4),
                                    (dict(change_estimation_subset=8, monte_carlo_num_rel=2), 160, 6),
                                    # a noise so large that the limits do not decide the calls after the simulated update: the
                                    # cooperative preparation (gen_prep_pu_kernel) flags its candidates, gen_prep_kernel takes them
                                    (dict(monte_carlo_num_rel=1, noise=0.5), 150, 9),
                                    (dict(monte_carlo_num_rel=2, noise=0.05), 120, 8)])
def test_general_scorer_pipeline_equals_its_single_kernel(dev, kw, n, k):
    """Plain mode of ital_score_generic: the pipeline of kernels on internal streams (gen_pipeline.hip: verdict / build /
    lattice sums / combine up to 6 variables, prepare / lattice sums / combine beyond; slabs and chunks of the workspace)
    against the one kernel that does everything per candidate -- same calls, same stream offsets, same order of the terms.
    The lattice sums run with more chains per lane in the pipeline (the single kernel keeps the runtime evaluator):
    last-bit differences, the lanes' partial sums are formed in another order.  (The fourth case passes the pipeline's
    exact-order recomputation of sums near 0 / 1 -- label_estimation 'pessimistic', qmc_exact.h -- which the single kernel
    does not have: no such sum occurs in it.)"""
    from ital_amd import ITAL, mvn_stream
    rng = np.random.default_rng(17)
    X = rng.random((n, 9))
    out = []
    for pipeline, work_bytes in ((True, 1 << 30), (True, 1 << 21), (False, 1 << 30)):
        mvn_stream.GLOBAL.reset()
        np.random.seed(3)
        L = ITAL(X, length_scale=0.9, device=dev, **kw)
        L.generic_pipeline = pipeline
        L.qmc_work_bytes = work_bytes                      # 2 MiB: many small slabs
        L.keep_scores = True
        L.update({1: 1, 2: -1})
        ret = L.fetch_unlabelled(k)
        out.append((ret, [s.cpu().numpy() for s in L.last_scores], mvn_stream.GLOBAL.draws))
    for ret, scores, draws in out[1:]:
        assert ret == out[0][0] and draws == out[0][2]
    for a, b in zip(out[0][1], out[1][1]):                 # slab size does not matter
        np.testing.assert_array_equal(a, b)
    for t, (a, b) in enumerate(zip(out[0][1], out[2][1])):
        live = np.isfinite(b)
        # (one and two variables are closed forms in both; the pipeline evaluates the simulated update in registers, the
        # single kernel in LDS slabs: the same sums, contracted differently by the compiler)
        np.testing.assert_allclose(a[live], b[live], rtol=1e-12, atol=1e-15)


def test_sharded_rows_holder_equals_the_full_matrix(dev):
    """sharding.ShardedRows (a rank passes only its own row block) against the full matrix, one rank."""
    from ital_amd import ITAL, mvn_stream, sharding
    rng = np.random.default_rng(23)
    X = rng.random((500, 20))
    res = []
    for data in (X, sharding.ShardedRows(X, 500, 0)):
        mvn_stream.GLOBAL.reset()
        L = ITAL(data, length_scale=1.2, device=dev)
        L.update({4: 1, 9: -1})
        res.append((L.fetch_unlabelled(4), L.rel_mean.copy(), len(L.get_unseen())))
    assert res[0][0] == res[1][0] and res[0][2] == res[1][2] == 498
    np.testing.assert_array_equal(res[0][1], res[1][1])
    with pytest.raises(ValueError):
        ITAL(sharding.ShardedRows(X[:100], 500, 0), length_scale=1.2, device=dev)   # not this rank's block


def test_labelled_set_capacity_growth_inside_a_session(dev):
    """The labelled-set buffers (V, L, the batch buffers that depend on their leading dimension) grow on demand: a session
    that starts with room for 16 labelled samples gives the picks and means of one that never has to grow."""
    from ital_amd import ITAL, mvn_stream
    rng = np.random.default_rng(31)
    X = rng.random((400, 10))
    rel = np.where(X[:, 0] > 0.5, 1.0, -1.0)
    res = []
    for cap in (16, None):
        mvn_stream.GLOBAL.reset()
        ITAL.gp_capacity = cap
        try:
            L = ITAL(X, length_scale=1.0, device=dev)
        finally:
            ITAL.gp_capacity = None
        L.update({0: 1})
        picks = []
        for _ in range(7):                                  # 1 + 7 x 4 = 29 labelled samples: crosses 16
            ret = L.fetch_unlabelled(4)
            picks.append(ret)
            L.update({i: float(rel[i]) for i in ret})
        res.append((picks, L.rel_mean.copy(), L.gp.cap))
    assert res[0][0] == res[1][0] and res[0][2] >= 32 and res[1][2] >= 64
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=0, atol=1e-12)


def test_state_dict_resumes_a_session(dev):
    """state_dict() / load_state_dict(): a session saved after two rounds and restored on a fresh learner continues with
    the same batches (incl. the position of the replayed mvndst stream) and the same predictive means."""
    import pickle
    from ital_amd import ITAL, mvn_stream
    rng = np.random.default_rng(12)
    X = rng.random((500, 9))
    lab = lambda i: 1.0 if X[i, 0] > 0.5 else -1.0   # noqa: E731
    mvn_stream.GLOBAL.reset()
    A = ITAL(X, length_scale=0.85, device=dev)
    A.update({4: 1, 7: -1})
    for _ in range(2):
        r = A.fetch_unlabelled(4)
        A.update({r[0]: lab(r[0]), r[1]: 0, r[2]: lab(r[2]), r[3]: lab(r[3])})      # one unnameable per round
    blob = pickle.dumps(A.state_dict())
    want = [A.fetch_unlabelled(4)]
    A.update({i: lab(i) for i in want[0]})
    want.append(A.fetch_unlabelled(3))
    mean_a = np.asarray(A.rel_mean).copy()
    mvn_stream.GLOBAL.reset()                          # "another process"
    B = ITAL(X, length_scale=0.85, device=dev)
    B.load_state_dict(pickle.loads(blob))
    assert B.gp.ind == A.gp.ind[: len(B.gp.ind)] and B.rounds == 3 and len(B.unnameable_ids) == 2
    got = [B.fetch_unlabelled(4)]
    B.update({i: lab(i) for i in got[0]})
    got.append(B.fetch_unlabelled(3))
    assert got == want
    np.testing.assert_allclose(np.asarray(B.rel_mean), mean_a, rtol=0, atol=1e-12)
    with pytest.raises(ValueError):
        ITAL(X, length_scale=0.5, device=dev).load_state_dict(pickle.loads(blob))


def test_state_dict_resumes_numpys_generator_too(dev):
    """Options that draw from numpy's global legacy generator (here MCMI_min's subsample, np.random.choice as the reference,
    mcmi.py:61-63): the saved session's generator state travels in the state_dict, the resumed session picks what the
    uninterrupted one picks."""
    import pickle
    from ital_amd import MCMI_min
    rng = np.random.default_rng(13)
    X = rng.random((300, 6))
    lab = lambda i: 1.0 if X[i, 0] > 0.5 else -1.0   # noqa: E731
    np.random.seed(21)
    A = MCMI_min(X, length_scale=0.7, subsample=60, device=dev)
    A.update({4: 1, 7: -1})
    r = A.fetch_unlabelled(3)
    A.update({i: lab(i) for i in r})
    blob = pickle.dumps(A.state_dict())
    want = A.fetch_unlabelled(3)
    np.random.seed(99)                                   # "another process": some other generator state
    B = MCMI_min(X, length_scale=0.7, subsample=60, device=dev)
    B.load_state_dict(pickle.loads(blob))
    assert B.fetch_unlabelled(3) == want


def test_unseen_bookkeeping_survives_sets_changed_behind_update(dev):
    """relevant_ids / irrelevant_ids / unnameable_ids are public attributes as in the reference (which rebuilds the candidate
    list from them on every call, retrieval_base.py:78-87): a set ASSIGNED anew, or ids moved between the sets, must show in
    get_unseen() and in the next batch although update() was not involved."""
    from ital_amd import ITAL, mvn_stream
    rng = np.random.default_rng(14)
    X = rng.random((200, 5))
    mvn_stream.GLOBAL.reset()
    L = ITAL(X, length_scale=0.6, device=dev)
    L.update({3: 1, 9: -1})
    first = L.fetch_unlabelled(3)
    assert 50 in L.get_unseen()
    L.unnameable_ids = {50, 51}                           # same-size replacement would not change the sizes check either
    assert 50 not in L.get_unseen() and 51 not in L.get_unseen() and len(L.get_unseen()) == 196
    L.unnameable_ids = {60, 61}
    assert 50 in L.get_unseen() and 60 not in L.get_unseen() and len(L.get_unseen()) == 196
    L.irrelevant_ids.add(70)                              # in-place change: the sizes differ
    assert 70 not in L.get_unseen()
    L.unnameable_ids.remove(61)                           # in-place swap: the sizes do NOT differ (round 5: the sets count
    L.unnameable_ids.add(62)                              # their changes, retrieval_base.IdSet)
    assert 61 in L.get_unseen() and 62 not in L.get_unseen()
    L.unnameable_ids.remove(62)
    L.unnameable_ids.add(61)
    got = L.fetch_unlabelled(3)
    assert not (set(got) & {3, 9, 60, 61, 70})
    mvn_stream.GLOBAL.reset()
    F = ITAL(X, length_scale=0.6, device=dev)
    F.update({3: 1, 9: -1})
    assert F.fetch_unlabelled(3) == first
    F.unnameable_ids = {60, 61}
    F.irrelevant_ids.add(70)
    assert F.fetch_unlabelled(3) == got


def test_fit_on_an_existing_learner_starts_over(dev):
    """fit(data) on a learner that already fetched (reference retrieval_base.py:34-45: a new GP, reset()): batch buffers,
    device candidate list and prepared round descriptors of the old data must not survive."""
    from ital_amd import ITAL, mvn_stream
    rng = np.random.default_rng(31)
    X1, X2 = rng.random((300, 7)), rng.random((420, 21))
    mvn_stream.GLOBAL.reset()
    L = ITAL(X1, length_scale=0.8, device=dev)
    L.update({0: 1})
    r1 = L.fetch_unlabelled(3)
    L.update({i: 1.0 for i in r1})
    L.fetch_unlabelled(3)
    L.fit(X2)
    L.length_scale = 1.3
    L.fit(X2)
    mvn_stream.GLOBAL.reset()
    L.update({5: 1, 6: -1})
    got = L.fetch_unlabelled(4)
    mvn_stream.GLOBAL.reset()
    F = ITAL(X2, length_scale=1.3, device=dev)
    F.update({5: 1, 6: -1})
    assert got == F.fetch_unlabelled(4)
