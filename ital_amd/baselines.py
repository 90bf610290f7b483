"""Ranking baselines on the device GP state (the cheap part): the comparison learners of the
reference whose acquisition score is a function of the predictive mean / variance / covariance the streaming GP already
keeps on the GPU -- RandomRetrieval, TopscoringSampling, BorderlineSampling, VarianceSampling (with and without
`use_correlations`), UncertaintySampling (reference ital/baseline_methods.py:12-58, :112-155, :203-227).  They exist so
that the reference's comparison tables run through `ital_amd.harness` against the same GP; the remaining baselines
(EMOC, SUD, RBMAL, TCAL, USDM, AdaptAL, entropy) are out of scope.
"""
import numpy as np
import torch

from ._lib import check
from .gp import _ptr, _stream
from .retrieval_base import ActiveRetrievalBase


class _RankingLearner(ActiveRetrievalBase):
    """Takes the first k unseen samples of a ranking (python ints, as the reference returns list elements)."""

    def _first_unseen(self, ranking, k):
        seen = self.relevant_ids | self.irrelevant_ids | self.unnameable_ids
        out = []
        for i in ranking:
            if i not in seen:
                out.append(i)
                if len(out) >= k:
                    break
        return out


class RandomRetrieval(ActiveRetrievalBase):
    def fetch_unlabelled(self, k):
        cand = self.get_unseen()
        return np.random.choice(cand, min(k, len(cand)), replace=False)


class TopscoringSampling(_RankingLearner):
    """Maximum predictive mean."""

    def fetch_unlabelled(self, k):
        return self._first_unseen(np.argsort(self.rel_mean)[::-1], k)


class BorderlineSampling(_RankingLearner):
    """Minimum absolute predictive mean."""

    def fetch_unlabelled(self, k):
        return self._first_unseen(np.argsort(np.abs(self.rel_mean)), k)


class UncertaintySampling(_RankingLearner):
    """Minimum certainty |mu| / sqrt(sigma^2 + noise) (Kapoor et al.)."""

    def fetch_unlabelled(self, k):
        mean, var = self.gp.predict_stored(cov_mode="diag")
        return self._first_unseen(np.argsort(np.abs(mean) / np.sqrt(var + self.gp.noise)), k)


class VarianceSampling(_RankingLearner):
    """Maximum predictive variance; with `use_correlations` greedily the batch with the largest sum of variances minus
    sum of covariances (one device covariance column per member, `ital_cross_cov_cols`)."""

    def __init__(self, data=None, queries=[], length_scale=0.1, var=1.0, noise=1e-6, use_correlations=False, **placement):
        ActiveRetrievalBase.__init__(self, data, queries, length_scale, var, noise, **placement)
        self.use_correlations = use_correlations

    def _cov_column(self, i):
        """Posterior covariance of sample i with every sample (numpy, all ranks)."""
        gp = self.gp
        row = gp._gather_rows([int(i)])
        nrm = torch.empty(1, dtype=torch.float64, device=gp.device)
        check(gp._lib.ital_row_norms(_ptr(row), 1, gp.ldx, _ptr(nrm), _stream()))
        w = torch.zeros((1, gp.cap), dtype=torch.float64, device=gp.device)
        w[0, : gp.m] = gp.gather_columns(gp.V[: max(gp.m, 1)], [int(i)])[: gp.m, 0]
        out = torch.empty(gp.ldv, dtype=torch.float64, device=gp.device)
        check(gp._lib.ital_cross_cov_cols(_ptr(gp.Xd), _ptr(gp.xnorm), gp.n, gp.ldx, _ptr(row), _ptr(nrm), 1, _ptr(w), gp.cap,
                                          _ptr(gp.V), gp.ldv, gp.m, float(self.var), float(self.length_scale), _ptr(out),
                                          gp.ldv, _stream()))
        return gp._full(out)

    def fetch_unlabelled(self, k):
        _, var = self.gp.predict_stored(cov_mode="diag")
        if not self.use_correlations:
            return self._first_unseen(np.argsort(var)[::-1], k)
        labelled = self.relevant_ids | self.irrelevant_ids
        start = np.where(np.isin(np.arange(var.size), list(labelled)), 0.0, var)       # baseline_methods.py:135
        ret = [int(np.argmax(start))]
        s2 = self.gp._full(self.gp.s2)                       # unclamped, as predict_cov_batch (gp.py:254)
        within = 0.0                                         # variances minus covariances among the members so far
        cols = []
        for _ in range(1, k):
            blocked = labelled | self.unnameable_ids | set(ret)
            cand = np.array([i for i in range(var.size) if i not in blocked], dtype=np.int64)
            if len(cand) == 0:
                break
            cols.append(self._cov_column(ret[-1]))
            members = np.array(ret)
            diag_members = np.array([cols[j][members[j]] for j in range(len(ret))])
            cov_members = sum(cols[a][members[b]] for a in range(len(ret)) for b in range(a))
            within = diag_members.sum() - cov_members
            scores = within + s2[cand] - sum(c[cand] for c in cols)
            ret.append(int(cand[np.argmax(scores)]))
        return ret


LEARNERS = {"random": RandomRetrieval, "topscoring": TopscoringSampling, "border": BorderlineSampling,
            "var": VarianceSampling, "unc": UncertaintySampling}
