"""ORACLE (test infrastructure): serial numpy restatement of the reference learner + MI scorer.

Follows reference ital/retrieval_base.py (bookkeeping), ital/ital.py (ITAL.fetch_unlabelled,
MutualInformation, AppendedMutualInformation, group_cov) and ital/mcmi.py, in the reference's own
*serial* evaluation order (`parallelized=False`), so that together with the bit-exact MVNDST
restatement (oracle/mvn.py) it reproduces the reference's random stream and therefore its picks.
Dense (needs the N x N kernel) and slow by construction; for small cases only.
"""
import itertools
import math

import numpy as np
from scipy.special import ndtr  # the very function the reference's norm.cdf evaluates

from . import mvn
from .gp import OracleGP

EPS = 1e-12  # reference ital/ital.py:144


def norm_cdf0(mean, sd):
    """scipy.stats.norm.cdf(0, mean, sd) as the reference calls it (ital.py:367-369, mcmi.py:117):
    ndtr((0 - mean)/sd), NaN where sd is not positive (scipy's scale check)."""
    mean = np.asarray(mean, dtype=np.float64)
    sd = np.asarray(sd, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        z = (0.0 - mean) / sd
        p = ndtr(z)
    return np.where(sd > 0, p, np.nan)


class OracleLearnerBase:
    """retrieval_base.py:7-193."""

    def __init__(self, data=None, queries=(), length_scale=0.1, var=1.0, noise=1e-6):
        self.length_scale, self.var, self.noise = length_scale, var, noise
        self.fit(data, queries)

    def fit(self, data, queries=()):
        self.data = data
        self.queries = list(queries)
        if data is None:
            self.gp = None
            return
        full = np.concatenate((data, self.queries)) if len(self.queries) else data
        self.gp = OracleGP(full, self.length_scale, self.var, self.noise)
        self.reset()

    def reset(self):
        self.rounds = 0
        self.relevant_ids, self.irrelevant_ids, self.unnameable_ids = set(), set(), set()
        if len(self.queries):
            n = len(self.data)
            self.gp.fit(np.arange(n, n + len(self.queries)), [1] * len(self.queries))
            self.rel_mean = self.gp.predict_stored()[:n]
        else:
            self.gp.reset()
            self.rel_mean = None

    def top_results(self, k=None):
        order = np.argsort(self.rel_mean)[::-1]
        return order if k is None else order[:k]

    def get_unseen(self):
        seen = self.relevant_ids | self.irrelevant_ids | self.unnameable_ids
        return [i for i in range(len(self.data)) if i not in seen]

    def partition_feedback(self, feedback):
        rel, irr, unn = [], [], []
        for i, fb in feedback.items():
            if fb > 0:
                if i in self.irrelevant_ids:
                    raise RuntimeError("Cannot change feedback once given.")
                if i not in self.relevant_ids:
                    rel.append(i)
            elif fb < 0:
                if i in self.relevant_ids:
                    raise RuntimeError("Cannot change feedback once given.")
                if i not in self.irrelevant_ids:
                    irr.append(i)
            else:
                unn.append(i)
        return rel, irr, unn

    def update(self, feedback):
        rel, irr, unn = self.partition_feedback(feedback)
        if rel or irr:
            self.gp.update(rel + irr, np.concatenate((np.ones(len(rel)), -np.ones(len(irr)))))
            self.rel_mean = self.gp.predict_stored()[: len(self.data)]
            self.relevant_ids.update(rel)
            self.irrelevant_ids.update(irr)
            self.rounds += 1
        self.unnameable_ids.update(unn)

    def updated_prediction(self, feedback, test_ind, cov_mode="full"):
        rel, irr, _ = self.partition_feedback(feedback)
        if not rel and not irr:
            return self.gp.predict_stored(test_ind, cov_mode=cov_mode)
        rel.sort()
        irr.sort()
        y = np.concatenate((np.ones(len(rel)), -np.ones(len(irr))))
        return self.gp.updated_prediction(rel + irr, y, test_ind, cov_mode=cov_mode)


# --------------------------------------------------------------------------- orthant probabilities

def group_cov(corr, th):
    """Connected components of |corr| > th (ital.py:590-616)."""
    adj = np.abs(corr) > th
    left = list(range(corr.shape[0]))
    groups = []
    while left:
        grp = []
        new = [j for j in np.nonzero(adj[left[0]])[0]]
        while new:
            grp += new
            left = [u for u in left if u not in new]
            reach = np.nonzero(adj[grp].max(axis=0))[0]
            new = [j for j in reach if j in left]
        groups.append(np.array(grp, dtype=int))
    return groups


_TRIL = {}


def _tril(n):
    if n not in _TRIL:
        _TRIL[n] = np.tril_indices(n, -1)
    return _TRIL[n]


def _mvn_call(pivot, infin, correl, n):
    return mvn.mvndst(pivot, pivot, infin, correl, maxpts=100 * n, abseps=1e-4, releps=1e-4)[1]


def prob_rel(rel, mean, cov, clip_cov=0):
    """P(sign pattern) under N(mean, cov) (ital.py:345-383)."""
    n = len(rel)
    if n > 5 and 0 < clip_cov < 1:
        return _grouped_prob_rel(rel, mean, cov, clip_cov)
    if n == 1:
        with np.errstate(invalid="ignore"):
            p_irr = float(norm_cdf0(mean[0], np.sqrt(cov[0, 0])))
        return 1.0 - p_irr if rel[0] else p_irr
    sd = np.sqrt(np.diag(cov))
    pivot = -np.asarray(mean, dtype=np.float64) / sd
    i, j = _tril(n)
    correl = cov[i, j] / (sd[i] * sd[j])
    return _mvn_call(pivot, np.asarray(rel, dtype=np.int32), correl, n)


def _grouped_prob_rel(rel, mean, cov, th):  # ital.py:386-429
    sd = np.sqrt(np.diag(cov))
    corr = cov / np.outer(sd, sd)
    pivot = -np.asarray(mean, dtype=np.float64) / sd
    infin = np.asarray(rel, dtype=np.int32)
    groups = group_cov(corr, th)
    if len(groups) == 1:
        i, j = np.tril_indices(len(rel), -1)
        return _mvn_call(pivot, infin, corr[i, j], len(rel))
    p = 1.0
    for g in groups:
        if len(g) == 1:
            q = float(norm_cdf0(mean[g[0]], sd[g[0]]))
            p *= (1 - q) if rel[g[0]] else q
        else:
            sub = corr[np.ix_(g, g)]
            i, j = np.tril_indices(len(g), -1)
            p *= _mvn_call(pivot[g], infin[g], sub[i, j], len(g))
    return p


# --------------------------------------------------------------------------- ITAL

class OracleITAL(OracleLearnerBase):
    """ital.py:12-134 (serial path)."""

    def __init__(self, data=None, queries=(), length_scale=0.1, var=1.0, noise=1e-6, label_prob=1.0,
                 mistake_prob=0.0, top_candidates=None, change_estimation_subset=0, clip_cov=0,
                 label_estimation="mean", monte_carlo_num_rel=None, monte_carlo_num_fb=None, parallelized=False):
        super().__init__(data, queries, length_scale, var, noise)
        self.label_prob, self.mistake_prob = label_prob, mistake_prob
        self.top_candidates = top_candidates
        self.change_estimation_subset = change_estimation_subset
        self.clip_cov = clip_cov
        self.label_estimation = label_estimation
        self.monte_carlo_num_rel, self.monte_carlo_num_fb = monte_carlo_num_rel, monte_carlo_num_fb
        self.trace = []  # per greedy step: (candidate list, MI values, pick) -- for the tests

    # ---- iterators (ital.py:278-342)
    def _rel_iter(self, n, mean, cov):
        num = n * self.monte_carlo_num_rel if self.monte_carlo_num_rel is not None else None
        if num is None or 2 ** (n - 1) < num:
            return itertools.product([False, True], repeat=n), 0
        # scipy.stats.multivariate_normal.rvs draws from the global numpy RandomState
        return np.random.multivariate_normal(mean, cov, num) > 0, num

    def _fb_iter(self, ret, rel):
        n = len(ret)
        if self.label_prob >= 1 and self.mistake_prob <= 0:
            return [[1 if rel[r] else -1 for r in ret]], 1
        if self.label_prob >= 1:
            num = n * self.monte_carlo_num_fb if self.monte_carlo_num_fb is not None else None
            if num is None or 2 ** (n - 1) < num:
                return itertools.product([-1, 1], repeat=n), 0
            s = np.random.choice([1, -1], (num, n), p=[1.0 - self.mistake_prob, self.mistake_prob])
            s[:, [i for i, r in enumerate(ret) if not rel[r]]] *= -1
            return s, num
        num = n * self.monte_carlo_num_fb if self.monte_carlo_num_fb is not None else None
        if num is None or 3 ** n < 2 * num:
            return itertools.product([-1, 0, 1], repeat=n), 0
        s = np.random.choice([0, 1, -1], (num, n), p=[1.0 - self.label_prob, self.label_prob * (1.0 - self.mistake_prob),
                                                      self.label_prob * self.mistake_prob])
        s[:, [i for i, r in enumerate(ret) if not rel[r]]] *= -1
        return s, num

    def _likelihood(self, feedback, rel):  # ital.py:453-481
        if any(v != 0 and k not in rel for k, v in feedback.items()):
            return 0
        p = 1.0
        for i, r in rel.items():
            fb = feedback.get(i, 0)
            if fb == 0:
                p *= 1.0 - self.label_prob
            elif fb == 2 * r - 1:
                p *= self.label_prob * (1.0 - self.mistake_prob)
            else:
                p *= self.label_prob * self.mistake_prob
        return p

    def _updated_prob_rel(self, rel, feedback):  # ital.py:432-450: variables re-ordered by data index
        ret = sorted(rel.keys())
        mean, cov = self.updated_prediction(feedback, ret)
        return prob_rel(np.array([rel[i] for i in ret]), mean, cov, self.clip_cov)

    # ---- MI (ital.py:183-275)
    def mutual_information(self, ret, mean, cov, rel_it=None, patterns=None):
        """`patterns`: explicit sign patterns instead of the ones rel_iter would enumerate / sample (tests: the Monte-Carlo
        estimate for a GIVEN sample, independent of LAPACK's sign conventions in multivariate_normal)."""
        ret = [int(i) for i in ret]
        if rel_it is not None:
            return self._mi_sub(ret, rel_it, mean, cov)
        mi = 0.0
        it, mc = self._rel_iter(len(ret), mean, cov) if patterns is None else (patterns, len(patterns))
        for reli in it:
            rel = {ret[i]: r for i, r in enumerate(reli)}
            pr = prob_rel(reli, mean, cov, self.clip_cov)
            log_pr = math.log(pr + EPS) if pr + EPS > 0 else float("nan")
            fbs, fb_mc = self._fb_iter(ret, rel)
            for fbi in fbs:
                if not any(fb != 0 for fb in fbi):
                    continue
                feedback = {ret[i]: fb for i, fb in enumerate(fbi)}
                pu = self._updated_prob_rel(rel, feedback)
                cur = (math.log(pu + EPS) if pu + EPS > 0 else float("nan")) - log_pr
                cur *= self._likelihood(feedback, rel) if fb_mc == 0 else 1.0 / fb_mc
                if self.label_estimation == "optimistic":
                    if cur > mi:
                        mi = cur
                elif self.label_estimation == "pessimistic":
                    if mi == 0 or cur < mi:
                        mi = cur
                else:
                    if mc == 0:
                        cur *= pr
                    mi += cur
        if mc > 0:
            mi /= mc
        return mi

    def _mi_sub(self, ret, rel_it, mean, cov):  # ital.py:227-275
        mean = np.asarray(mean, dtype=np.float64)
        mean_it = mean[rel_it]
        cov_it = cov[np.ix_(rel_it, rel_it)]
        rel_vec = mean > 0
        rel = {i: bool(r) for i, r in zip(ret, rel_vec)}
        sub_ids = [ret[i] for i in rel_it]
        mi = 0.0
        it, mc = self._rel_iter(len(sub_ids), mean_it, cov_it)
        for reli in it:
            rel_sub = {}
            for i, r in zip(rel_it, reli):
                rel_vec[i] = r
                rel[ret[i]] = r
                rel_sub[ret[i]] = r
            pr = prob_rel(reli, mean_it, cov_it, self.clip_cov)
            log_pr = math.log(prob_rel(rel_vec, mean, cov, self.clip_cov) + EPS)
            fbs, fb_mc = self._fb_iter(sub_ids, rel)
            for fbi in fbs:
                if not any(fb != 0 for fb in fbi):
                    continue
                feedback = {ret[i]: fb for i, fb in zip(rel_it, fbi)}
                pu = self._updated_prob_rel(rel, feedback)
                cur = math.log(pu + EPS) - log_pr
                cur *= self._likelihood(feedback, rel_sub) if fb_mc == 0 else 1.0 / fb_mc
                if mc == 0:
                    cur *= pr
                mi += cur
            for i in rel_it:
                rel_vec[i] = mean[i] > 0
                rel[ret[i]] = bool(rel_vec[i])
        if mc > 0:
            mi /= mc
        return mi

    # ---- greedy batch construction (ital.py:84-134 with AppendedMutualInformation, :485-586)
    def fetch_unlabelled(self, k, show_progress=False, forced=None, patterns=None):
        """`forced` (tests): the picks to append instead of the arg-max of every step -- the trace then holds this
        restatement's MI of every candidate GIVEN that batch (what a differing pick has to be judged against: a numerical
        tie of the two top values).  `patterns` (tests): per greedy step a dict candidate -> list of sign patterns to use
        instead of the Monte-Carlo sampler's (the estimate for a given sample; numpy's generator is then not touched)."""
        cand = self.get_unseen()
        k = min(k, len(cand))
        if self.change_estimation_subset is None:
            self._ce_subset = cand
        elif self.change_estimation_subset > 0:
            self._ce_subset = sorted(np.random.choice(cand, min(len(cand), self.change_estimation_subset), replace=False))
        else:
            self._ce_subset = None
        if self.top_candidates is not None:
            top = self.top_candidates
            if isinstance(top, float):
                top = min(len(cand), int(top * (len(self.queries) + len(self.relevant_ids) + len(self.irrelevant_ids))))
            if 0 < top < len(cand):
                sel = np.argpartition(self.rel_mean[cand], -top)[-top:]
                cand = [cand[i] for i in sel]
        state = _Appended(self)
        self.trace = []
        for step in range(k):
            given = None if patterns is None else patterns[step]
            vals = [state.score(i, patterns=None if given is None else given[i]) for i in cand]
            best = int(np.argmax(vals))  # first maximum; NaN wins (ital.py:130)
            self.trace.append((list(cand), np.array(vals, dtype=np.float64), cand[best]))
            if forced is not None:
                best = cand.index(int(forced[step]))
            state.append(cand[best])
            del cand[best]
        return state.ret


class _Appended:
    """AppendedMutualInformation (ital.py:485-586)."""

    def __init__(self, learner):
        self.L = learner
        self.ret = []
        n = len(learner.data)
        if learner._ce_subset is not None:
            sub = [int(i) for i in learner._ce_subset]
            self.pos = {ind: i for i, ind in enumerate(sub)}
            self.ext = list(sub)
            self.ret_pos = []
            self._refresh_sub()
        else:
            self.covs = learner.gp.predict_stored(cov_mode="diag")[1].reshape(-1, 1, 1)  # clamped (gp.py:229)

    def _refresh_sub(self):
        n = len(self.L.data)
        out = np.setdiff1d(np.arange(n), self.ext)
        self.cov_pos = {int(ind): i for i, ind in enumerate(out)}
        self.covs = self.L.gp.predict_cov_batch(self.ext, out)
        self.cov_base = self.L.gp.predict_stored(self.ext, cov_mode="full")[1]

    def score(self, i, patterns=None):
        L = self.L
        if L._ce_subset is not None:
            if i in self.pos:
                ret, rel_it, cov = self.ext, self.ret_pos + [self.pos[i]], self.cov_base
            else:
                ret, rel_it, cov = self.ext + [i], self.ret_pos + [len(self.ext)], self.covs[self.cov_pos[i]]
            return L.mutual_information(ret, L.rel_mean[ret], cov, rel_it=rel_it)
        ids = self.ret + [i]
        return L.mutual_information(ids, L.rel_mean[ids], self.covs[i], patterns=patterns)

    def append(self, i):
        self.ret.append(i)
        L = self.L
        if L._ce_subset is not None:
            if i not in self.pos:
                self.pos[i] = len(self.ext)
                self.ext.append(i)
                self._refresh_sub()
            self.ret_pos.append(self.pos[i])
        else:
            self.covs = L.gp.predict_cov_batch(self.ret, np.arange(len(L.data)))


# --------------------------------------------------------------------------- MCMI_min

class OracleMCMI(OracleLearnerBase):
    """ital/mcmi.py:12-124 (serial)."""

    def __init__(self, data=None, queries=(), length_scale=0.1, var=1.0, noise=1e-6, subsample=None, parallelized=False):
        super().__init__(data, queries, length_scale, var, noise)
        self.subsample = subsample
        self.trace = []

    def conditional_entropy(self, ret):  # mcmi.py:101-124
        best = None
        for reli in itertools.product([False, True], repeat=len(ret)):
            mean, var = self.updated_prediction({i: 1 if r else -1 for i, r in zip(ret, reli)}, self.candidates,
                                                cov_mode="diag")
            p_irr = norm_cdf0(mean, np.sqrt(var))
            p_rel = 1.0 - p_irr
            cur = float(np.sum(p_irr * np.log(p_irr + EPS) + p_rel * np.log(p_rel + EPS)))
            if best is None or cur < best:
                best = cur
        return best

    def fetch_unlabelled(self, k, show_progress=False):  # mcmi.py:48-81
        self.candidates = self.get_unseen()
        if self.subsample and self.subsample < len(self.candidates):
            self.candidates = np.random.choice(self.candidates, self.subsample, replace=False).tolist()
        k = min(k, len(self.candidates))
        ret = []
        self.trace = []
        for _ in range(k):
            vals = [self.conditional_entropy(ret + [i]) for i in self.candidates]
            best = int(np.argmin(vals))
            self.trace.append((list(self.candidates), np.array(vals, dtype=np.float64), self.candidates[best]))
            ret.append(self.candidates[best])
            del self.candidates[best]
        return ret
