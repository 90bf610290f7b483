"""Comparison learners on the device GP state: the baselines of the reference whose
acquisition score is a function of what the streaming GP already keeps on the GPU -- RandomRetrieval,
TopscoringSampling, BorderlineSampling, BorderlineDiversitySampling, VarianceSampling (with and without
`use_correlations`), UncertaintySampling, EntropySampling (the orthant kernel of the ITAL scorer) and EMOC (the FP64
MFMA covariance tiles of MCMI) -- reference ital/baseline_methods.py:12-155, :203-287, :338-381.  They exist so that the
reference's comparison tables run through `ital_amd.harness` against the same GP; the remaining baselines (SUD, RBMAL,
TCAL, USDM, AdaptAL) use models of their own and are out of scope.
"""
import numpy as np
import torch
from scipy.special import ndtr

from . import mvn_stream, sharding
from ._lib import check
from .gp import _pad16, _ptr, _stream
from .ital import ITAL
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


class BorderlineDiversitySampling(ActiveRetrievalBase):
    """Small |mean| traded against the largest kernel-space cosine to the members chosen so far (Brinker; reference
    baseline_methods.py:64-108): one RBF column per member (`ital_rbf_cols`) instead of slices of K_all."""

    def __init__(self, data=None, queries=[], length_scale=0.1, var=1.0, noise=1e-6, alpha=0.5, **placement):
        ActiveRetrievalBase.__init__(self, data, queries, length_scale, var, noise, **placement)
        self.alpha = alpha

    def fetch_unlabelled(self, k):
        gp = self.gp
        candidates = self._unseen_array()
        absmean = np.abs(self.rel_mean)
        min_ind = int(np.argmin(absmean[candidates]))                # raises on an empty list as the reference does
        ret = [int(candidates[min_ind])]
        diversity = None
        for _ in range(1, k):
            candidates = np.delete(candidates, min_ind)
            if len(candidates) == 0:
                break
            if diversity is not None:
                diversity = np.delete(diversity, min_ind)
            col = gp._full(gp.rbf_cols([ret[-1]])[0])               # k(x_j, x_ret) for every sample j
            # cosine: the diagonal of the kernel matrix is `var` (up to the rounding of |a|^2 + |a|^2 - 2 a.a)
            angle = col[candidates] / np.sqrt(self.var) / np.sqrt(col[ret[-1]])
            diversity = angle if diversity is None else np.maximum(diversity, angle)
            scores = self.alpha * absmean[candidates] + (1.0 - self.alpha) * diversity
            min_ind = int(np.argmin(scores))
            ret.append(int(candidates[min_ind]))
        return ret


class EMOC(ActiveRetrievalBase):
    """Expected model output change (Freytag et al.; reference baseline_methods.py:338-381).

    The reference multiplies, per candidate i, the change of the weight vector for either label with K_all[[T, i], :]
    and averages the absolute values over all N samples; that product is (+-1 - mu_i) / (s2_i + noise) times the
    posterior covariance of i with every sample, so the score is
        (P(+) |1 - mu_i| + P(-) |-1 - mu_i|) / (s2_i + noise) * mean_j |Sigma_ij|
    and the N x N part is `ital_cov_abs_rowsum` (FP64 MFMA tiles, nothing materialised).  Rows are sharded across ranks;
    the column blocks of the other ranks arrive by broadcast, one at a time."""

    def _mean_abs_cov(self):
        gp = self.gp
        dev = gp.device
        lib = gp._lib
        with torch.cuda.device(dev):
            st = _stream()
            out = torch.zeros(max(gp.n, 1), dtype=torch.float64, device=dev)
            work = torch.empty(max(gp.n, 1) * 64, dtype=torch.float64, device=dev)
            m = gp.m
            for r in range(gp.world if gp.collective else 1):
                if gp.collective:
                    r0, r1 = sharding.row_range(gp.n_total, gp.world, r)
                    nb = r1 - r0
                    ldb = _pad16(max(nb, 1))
                    Xb = torch.zeros((max(nb, 1), gp.ldx), dtype=torch.float64, device=dev)
                    bn = torch.zeros(max(nb, 1), dtype=torch.float64, device=dev)
                    Vb = torch.zeros((max(m, 1), ldb), dtype=torch.float64, device=dev)
                    if r == gp.rank and nb:
                        Xb.copy_(gp.Xd[:nb])
                        bn.copy_(gp.xnorm[:nb])
                        Vb[:m, :nb] = gp.V[:m, :nb]
                    for buf in (Xb, bn, Vb):
                        sharding.broadcast(buf, r, gp.group)
                else:
                    nb, ldb, Xb, bn, Vb = gp.n, gp.ldv, gp.Xd, gp.xnorm, gp.V
                if gp.n and nb:
                    check(lib.ital_cov_abs_rowsum(_ptr(gp.Xd), _ptr(gp.xnorm), gp.n, _ptr(Xb), _ptr(bn), nb, gp.ldx,
                                                  _ptr(gp.V), gp.ldv, _ptr(Vb), ldb, m, float(self.var),
                                                  float(self.length_scale), _ptr(work), work.numel(), int(r > 0),
                                                  _ptr(out), st))
                    torch.cuda.current_stream().synchronize()       # the broadcast block is a temporary
            # queries are labelled points outside the data matrix (the reference appends them as rows of K_all,
            # retrieval_base.py:40): their whitened columns are L^-1 K_T,q = L^T[:, q] - noise L^-1 e_q
            qpos = [t for t, i in enumerate(gp.ind) if i >= gp.n_total]
            if qpos and gp.n:
                nq = len(qpos)
                Lm = gp.L[:m, :m]
                E = torch.zeros((m, nq), dtype=torch.float64, device=dev)
                E[qpos, list(range(nq))] = 1.0
                Vq = torch.zeros((m, _pad16(nq)), dtype=torch.float64, device=dev)
                Vq[:, :nq] = Lm.t()[:, qpos] - float(gp.noise) * torch.linalg.solve_triangular(Lm, E, upper=False)
                Xq, qn = gp.XT[qpos].contiguous(), gp.XTn[qpos].contiguous()
                check(lib.ital_cov_abs_rowsum(_ptr(gp.Xd), _ptr(gp.xnorm), gp.n, _ptr(Xq), _ptr(qn), nq, gp.ldx,
                                              _ptr(gp.V), gp.ldv, _ptr(Vq), Vq.shape[1], m, float(self.var),
                                              float(self.length_scale), _ptr(work), work.numel(), 1, _ptr(out), st))
                torch.cuda.current_stream().synchronize()
            return gp._full(out) / (gp.n_total + len(qpos))

    def emoc_scores(self, ind):
        ind = np.asarray(ind, dtype=np.int64)
        mean, variance = self.gp.predict_stored(ind, cov_mode="diag")
        spread = self._mean_abs_cov()[ind]
        denom = variance + self.gp.noise
        moc_pos = np.abs((1 - mean) / denom) * spread
        moc_neg = np.abs((-1 - mean) / denom) * spread
        with np.errstate(divide="ignore", invalid="ignore"):
            sd = np.sqrt(variance)
            prob_neg = np.where(sd > 0, ndtr((0.0 - mean) / sd), np.nan)   # scipy.stats.norm.cdf(0, mean, sd)
        return (1 - prob_neg) * moc_pos + prob_neg * moc_neg

    def fetch_unlabelled(self, k):
        candidates = self._unseen_array()
        k = min(k, len(candidates))
        self.last_scores = self.emoc_scores(candidates)
        return candidates[np.argsort(self.last_scores)[::-1][:k]].tolist()


class EntropySampling(ITAL):
    """Greedy batch of maximum joint entropy of the relevance signs (Konyushkova et al.; reference
    baseline_methods.py:229-287).  The batch entropy -sum_r p(r) log p(r) over all sign patterns of (members, candidate)
    runs on the general orthant scorer (`ital_score_generic`, fb_mode 3); selection is the scorer's own arg-max.

    The reference evaluates the batch entropies in a multiprocessing.Pool whose forked workers inherit the MVNDST
    generator state (which worker takes which candidates is not reproducible; the parent's state never moves).  Here
    the candidates take the stream in list order, i.e. the one-worker schedule of that pool, and the process-wide
    stream is put back afterwards."""

    def __init__(self, data=None, queries=[], length_scale=0.1, var=1.0, noise=1e-6, **placement):
        ITAL.__init__(self, data, queries, length_scale, var, noise, **placement)

    def _fb_mode(self):
        return 3

    def _needs_generic(self):
        return True

    def fetch_unlabelled(self, k, show_progress=False):
        if self.gp.m == 0:
            raise RuntimeError("fetch_unlabelled() needs a fitted relevance model: call update() first or pass queries")
        unseen = self._unseen_array()
        if len(unseen) == 0:
            raise ValueError("max() arg is an empty sequence")       # baseline_methods.py:247
        k = min(int(k), len(unseen))
        if k <= 0:
            return []
        why = self._unsupported(k, len(unseen))
        if why is not None:
            raise NotImplementedError("ital_amd device scorer: %s is not implemented" % why)
        self._ce_subset = None
        stream = mvn_stream.GLOBAL
        saved = (stream.state, stream.draws)
        try:
            return self._fetch_generic(k, unseen)
        finally:
            stream.state, stream.draws = saved


LEARNERS = {"random": RandomRetrieval, "topscoring": TopscoringSampling, "border": BorderlineSampling,
            "border_div": BorderlineDiversitySampling, "var": VarianceSampling, "unc": UncertaintySampling,
            "entropy": EntropySampling, "EMOC": EMOC}
