"""ORACLE (test infrastructure): numpy restatement of the comparison learners that share ITAL's GP and orthant
integrator -- EMOC, EntropySampling and BorderlineDiversitySampling of reference ital/baseline_methods.py.

Dense (N x N kernel matrix), serial, for small cases only.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package.

EntropySampling evaluates its batch entropies in a multiprocessing.Pool (baseline_methods.py:250-258): the forked
workers inherit the parent's MVNDST generator state and the parent's own state never moves.  The restatement is
the one-worker schedule of that pool: tasks in list order on a stream that starts at the parent's state and
carries on through the greedy steps of one fetch; the parent's state is put back afterwards.
"""
import itertools

import numpy as np

from . import mvn
from .ital import OracleLearnerBase, _mvn_call, _tril, norm_cdf0


class OracleEMOC(OracleLearnerBase):
    """baseline_methods.py:338-381."""

    def emoc_scores(self, ind):
        gp = self.gp
        ind = np.asarray(ind, dtype=np.int64)
        mean, variance = gp.predict_stored(ind, cov_mode="diag")                                   # :361
        k_diff = np.hstack((gp.K_all[np.ix_(ind, gp.ind)] @ gp.K_inv, np.zeros((len(ind), 1)) - 1))  # :364
        denom = variance + gp.noise
        ad_pos = ((1 - mean) / denom)[:, None] * k_diff
        ad_neg = ((-1 - mean) / denom)[:, None] * k_diff
        moc = np.array([np.abs(np.vstack((p, n)) @ gp.K_all[np.r_[gp.ind, [i]], :]).mean(axis=-1)
                        for i, p, n in zip(ind, ad_pos, ad_neg)])                                  # :370-373
        with np.errstate(invalid="ignore"):
            prob_neg = norm_cdf0(mean, np.sqrt(variance))                                          # :376
        return (1 - prob_neg) * moc[:, 0] + prob_neg * moc[:, 1]

    def fetch_unlabelled(self, k):
        candidates = np.array(self.get_unseen())
        k = min(k, len(candidates))
        self.last_scores = self.emoc_scores(candidates)
        self.last_candidates = candidates
        return candidates[np.argsort(self.last_scores)[::-1][:k]].tolist()


def single_entropy(mean, var):  # baseline_methods.py:263-267
    with np.errstate(invalid="ignore", divide="ignore"):
        p = float(norm_cdf0(mean, np.sqrt(var)))
    prob_irr = max(1e-8, min(1.0 - 1e-8, p))
    return -1 * (prob_irr * np.log(prob_irr) + (1.0 - prob_irr) * np.log(1.0 - prob_irr))


def batch_entropy(mean, cov):  # baseline_methods.py:270-287
    n = len(mean)
    stdev = np.sqrt(np.diag(cov))
    pivot = -np.asarray(mean, dtype=np.float64) / stdev
    i, j = _tril(n)
    correl = cov[i, j] / (stdev[i] * stdev[j])
    entropy = 0.0
    for rel in itertools.product([False, True], repeat=n):
        pr = _mvn_call(pivot, np.array(rel, dtype=np.int32), correl, n)
        if pr > 1e-12:
            entropy += pr * np.log(pr)
    return -1 * entropy


class OracleEntropy(OracleLearnerBase):
    """baseline_methods.py:229-287 (one-worker schedule of its pool, see the module docstring)."""

    def fetch_unlabelled(self, k):
        rel_mean, rel_var = self.gp.predict_stored(cov_mode="diag")
        n = len(self.data)
        rel_mean, rel_var = rel_mean[:n], rel_var[:n]
        candidates = self.get_unseen()
        self.trace = []
        if not candidates:
            raise ValueError("max() arg is an empty sequence")       # baseline_methods.py:247
        ent = [single_entropy(rel_mean[c], rel_var[c]) for c in candidates]
        self.trace.append((list(candidates), np.array(ent)))
        max_ind = int(np.argmax(ent))                 # max(range, key=...) keeps the first maximum as argmax does
        ret = [candidates[max_ind]]
        saved = mvn.rng_state()
        try:
            for _ in range(1, k):
                del candidates[max_ind]
                if len(candidates) == 0:
                    break
                covs = self.gp.predict_cov_batch(ret, candidates)
                ent = [batch_entropy(rel_mean[ret + [c]], covs[i]) for i, c in enumerate(candidates)]
                self.trace.append((list(candidates), np.array(ent)))
                max_ind = int(np.argmax(ent))
                ret.append(candidates[max_ind])
        finally:
            mvn.rng_set_state(saved)
        return ret


class OracleBorderDiv(OracleLearnerBase):
    """baseline_methods.py:64-108."""

    def __init__(self, data=None, queries=(), length_scale=0.1, var=1.0, noise=1e-6, alpha=0.5):
        OracleLearnerBase.__init__(self, data, queries, length_scale, var, noise)
        self.alpha = alpha

    def fetch_unlabelled(self, k):
        K = self.gp.K_all
        candidates = self.get_unseen()
        min_ind = int(np.argmin(np.abs(self.rel_mean[candidates])))
        ret = [candidates[min_ind]]
        for _ in range(1, k):
            del candidates[min_ind]
            if len(candidates) == 0:
                break
            angle = K[np.ix_(candidates, ret)].copy()
            angle /= np.sqrt(K[candidates, candidates])[:, None]
            angle /= np.sqrt(K[ret, ret])[None, :]
            scores = self.alpha * np.abs(self.rel_mean[candidates]) + (1.0 - self.alpha) * angle.max(axis=-1)
            min_ind = int(np.argmin(scores))
            ret.append(candidates[min_ind])
        return ret
