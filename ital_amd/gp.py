"""Streaming Gaussian process on MI355X: host-side mirror of reference ital/gp.py `GaussianProcess`.

Same public surface for the hot path (fit / update / reset / predict_stored / predict, attributes
ind, y, X, length_scale, var, noise) but a different state: the reference stores the dense N x N kernel
`K_all` (gp.py:128) and an explicit inverse re-computed at every update (gp.py:194); this class keeps, per
GPU, the row shard of X and the Cholesky-whitened block

    L L^T = K[T,T] + noise*I,   V = L^-1 K[T,:]  (m x n, one contiguous n-vector per labelled point),
    mu = V^T alpha (alpha = L^-1 y),   s2 = var - colsum(V^2)          

so an update appends c rows (rank-c Cholesky append) and N never appears squared.  All arithmetic runs in
the HIP kernels of libital_hip.so; torch only owns the device buffers and the stream.
"""
import ctypes
import os
import time

import numpy as np
import torch

from . import _lib, sharding
from ._lib import check


def _pad16(v):
    return (int(v) + 15) // 16 * 16


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


class GaussianProcess(object):
    """GP on a row shard of `data` (rows [row0, row1) on this rank).

    # Arguments (as reference ital/gp.py:103-120, plus keyword-only placement):
    - data: n-by-d array of all samples.
    - length_scale, var, noise: kernel hyper-parameters.
    - device: torch device of this rank.
    - rank / world / group: row sharding over `world` processes (torch.distributed group for the exchange).
    """

    def __init__(self, data, length_scale, var=1.0, noise=1e-6, pdist=None, *, device=None, rank=0, world=1,
                 group=None, capacity=None):
        if pdist is not None:
            raise NotImplementedError("pre-computed distances are a dense-kernel feature of the reference (gp.py:116-128)")
        if not torch.cuda.is_available():
            raise RuntimeError("ital_amd.GaussianProcess needs a HIP device (no CPU fallback)")
        self._lib = _lib.lib()
        self.device = torch.device(device if device is not None else "cuda:0")
        self.rank, self.world, self.group = int(rank), int(world), group
        # the exchange steps run whenever there is more than one rank -- or, with ITAL_FORCE_COLLECTIVES=1, also on a
        # one-rank process group (lets a single-GPU box drive the RCCL code path end to end)
        self.collective = self.world > 1 or (group is not None and os.environ.get("ITAL_FORCE_COLLECTIVES") == "1")
        local = None
        if isinstance(data, sharding.ShardedRows):
            local = data
        else:
            data = np.asarray(data, dtype=np.float64)
        if data.ndim != 2:
            raise ValueError("data must be an n-by-d array")
        self.n_total, self.d = data.shape
        self.row0, self.row1 = sharding.row_range(self.n_total, self.world, self.rank)
        if local is not None and (local.row0, local.row1) != (self.row0, self.row1):
            raise ValueError("ShardedRows holds rows [%d, %d), this rank owns [%d, %d)"
                             % (local.row0, local.row1, self.row0, self.row1))
        self.n = self.row1 - self.row0
        self.ldx = _pad16(self.d)
        self.ldv = _pad16(max(self.n, 1))
        self.length_scale = length_scale
        self.length_scale_sq = length_scale * length_scale
        self.var = var
        self.noise = noise
        self.X_host = data  # the reference keeps a copy too (gp.py:122)
        with torch.cuda.device(self.device):
            Xp = torch.zeros((max(self.n, 1), self.ldx), dtype=torch.float64, device=self.device)
            if self.n:
                rows = local.local if local is not None else data[self.row0:self.row1]
                Xp[: self.n, : self.d] = torch.from_numpy(np.ascontiguousarray(rows)).to(self.device)
            self.Xd = Xp
            self.xnorm = torch.empty(max(self.n, 1), dtype=torch.float64, device=self.device)
            check(self._lib.ital_row_norms(_ptr(self.Xd), self.n, self.ldx, _ptr(self.xnorm), _stream()))
            self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._gather_bufs = {}
            if capacity is None:
                # room for 256 labelled samples (a session of 60 rounds of 4) unless the whitened block V would take more
                # than 2 GiB up front; growing later costs a reallocation and a copy of V and of the batch buffers (13 ms
                # at 9298 rows: visible in a 3 ms round)
                capacity = int(min(256, max(64, (2 << 30) // (8 * self.ldv))))
            self._alloc(capacity)
        self.reset()

    # ------------------------------------------------------------------ storage
    def _alloc(self, cap):
        cap = _pad16(cap)
        dev = self.device
        old = getattr(self, "cap", 0)
        V = torch.zeros((cap, self.ldv), dtype=torch.float64, device=dev)
        L = torch.zeros((cap, cap), dtype=torch.float64, device=dev)
        alpha = torch.zeros(cap, dtype=torch.float64, device=dev)
        XT = torch.zeros((cap, self.ldx), dtype=torch.float64, device=dev)
        XTn = torch.zeros(cap, dtype=torch.float64, device=dev)
        if old:
            V[:old] = self.V
            L[:old, :old] = self.L
            alpha[:old] = self.alpha
            XT[:old] = self.XT
            XTn[:old] = self.XTn
        self.V, self.L, self.alpha, self.XT, self.XTn, self.cap = V, L, alpha, XT, XTn, cap

    @property
    def X(self):
        return self.X_host

    @property
    def K_all(self):
        raise AttributeError("the dense N x N kernel matrix is never formed by ital_amd (reference gp.py:128 is "
                             "replaced by column streaming); use rbf_cols()")

    # attributes of the reference's GP that derive from the m x m Gram matrix of the labelled samples: reconstructed on
    # demand from the Cholesky factor kept on the device (K = L L^T includes the noise term, as gp.py:156)
    def _factor(self):
        return self.L[: self.m, : self.m].cpu().numpy()

    @property
    def K(self):
        if self.m == 0:
            return None
        f = self._factor()
        return f @ f.T

    @property
    def K_inv(self):
        if self.m == 0:
            return None
        fi = np.linalg.inv(self._factor())
        return fi.T @ fi

    @property
    def w(self):
        if self.m == 0:
            return None
        import scipy.linalg
        return scipy.linalg.solve_triangular(self._factor(), self.alpha[: self.m].cpu().numpy(), lower=True, trans="T")

    def kernel(self, a, b=None):
        """RBF kernel between the rows of `a` and of `b` (default: the labelled samples vs `a`, as reference
        gp.py:390-416), computed by ital_cov_block (FP64 MFMA) with an empty whitened part."""
        if b is None:
            if self.m == 0:
                raise RuntimeError("the GP has not been fitted")
            a, b = self.XT[: self.m, : self.d].cpu().numpy(), a
        a = np.atleast_2d(np.asarray(a, dtype=np.float64))
        b = np.atleast_2d(np.asarray(b, dtype=np.float64))
        dev = self.device
        A = torch.zeros((len(a), self.ldx), dtype=torch.float64, device=dev)
        B = torch.zeros((len(b), self.ldx), dtype=torch.float64, device=dev)
        A[:, : self.d] = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        B[:, : self.d] = torch.from_numpy(np.ascontiguousarray(b)).to(dev)
        an = torch.empty(len(a), dtype=torch.float64, device=dev)
        bn = torch.empty(len(b), dtype=torch.float64, device=dev)
        out = torch.empty((len(a), _pad16(len(b))), dtype=torch.float64, device=dev)
        st = _stream()
        check(self._lib.ital_row_norms(_ptr(A), len(a), self.ldx, _ptr(an), st))
        check(self._lib.ital_row_norms(_ptr(B), len(b), self.ldx, _ptr(bn), st))
        check(self._lib.ital_cov_block(_ptr(A), _ptr(an), len(a), _ptr(B), _ptr(bn), len(b), self.ldx, 0, 0, 0, 0, 0,
                                       float(self.var), float(self.length_scale), _ptr(out), out.shape[1], st))
        return out[:, : len(b)].cpu().numpy()

    def reset(self):
        """Back to the state right after __init__ (reference gp.py:132-138)."""
        self.ind = []
        self.y = None
        self.m = 0
        self.appends = []          # sizes of the update() calls that built the labelled set (state_dict / replay)
        self.mu = torch.zeros(max(self.n, 1), dtype=torch.float64, device=self.device)
        self.s2 = torch.full((max(self.n, 1),), float(self.var), dtype=torch.float64, device=self.device)
        self.mu_all = torch.zeros(self.n_total, dtype=torch.float64, device=self.device) if self.collective else self.mu[: self.n]
        self._mean_host = None
        self.status.zero_()

    # ------------------------------------------------------------------ fitting
    def fit(self, ind, y):
        """Fits to a subset of the data (reference gp.py:141-161)."""
        self.reset()
        return self.update(ind, y)

    def update(self, ind, y, row_cache=None):
        """Adds labelled samples (reference gp.py:164-200), as a rank-c Cholesky append.

        `row_cache`: optional (device matrix [k, ldx], {data index: row of that matrix}) holding feature rows that are
        already replicated on every rank -- the batch state of the last fetch_unlabelled: the winners' rows travelled in
        the selection records, so labelling them needs no further exchange between the ranks."""
        ind = [int(i) for i in ind]
        y = np.asarray(y, dtype=np.float64).reshape(-1)
        if len(ind) != len(y):
            raise ValueError("ind and y differ in length")
        if len(ind) == 0:
            return self
        hc = getattr(self, "host_clock", None)       # bench.py: where the host's time between two rounds goes
        t0 = time.perf_counter() if hc is not None else 0.0
        if row_cache is not None and all(i in row_cache[1] for i in ind):
            if len(ind) <= 16:
                self._append_staged(row_cache[0], [row_cache[1][i] for i in ind], y)
            else:
                slots = torch.as_tensor([row_cache[1][i] for i in ind], dtype=torch.int64, device=self.device)
                self._append(row_cache[0].index_select(0, slots), y)
        else:
            self._append(self._gather_rows(ind), y)
        self.ind += ind
        self.y = y.copy() if self.y is None else np.concatenate((self.y, y))
        self.appends.append(len(ind))
        t1 = time.perf_counter() if hc is not None else 0.0
        self._replicate_mean()
        if hc is not None:
            hc["append_s"] = hc.get("append_s", 0.0) + (t1 - t0)
            hc["means_s"] = hc.get("means_s", 0.0) + (time.perf_counter() - t1)
        return self

    def update_points(self, points, y, ind=None):
        """Adds labelled feature vectors that are not rows of the data matrix (the reference appends queries as
        extra rows, ital/retrieval_base.py:40,56-58)."""
        pts = np.atleast_2d(np.asarray(points, dtype=np.float64))
        y = np.asarray(y, dtype=np.float64).reshape(-1)
        rows = torch.zeros((len(pts), self.ldx), dtype=torch.float64, device=self.device)
        rows[:, : self.d] = torch.from_numpy(np.ascontiguousarray(pts)).to(self.device)
        self._append(rows, y)
        self.ind += list(ind) if ind is not None else list(range(self.n_total + self.m - len(y), self.n_total + self.m))
        self.y = y.copy() if self.y is None else np.concatenate((self.y, y))
        self.appends.append(-len(y))          # negative: feature vectors that are not rows of the data (queries)
        self._replicate_mean()
        return self

    def _owned(self, ind):
        """Host-side ownership of global indices: (mask as a float64 device column, clamped local indices on the device).
        Worked out with numpy -- a device-side mask would cost a nonzero() and a host synchronisation per use."""
        idx = np.asarray([int(i) for i in ind], dtype=np.int64)
        own = (idx >= self.row0) & (idx < self.row1)
        loc = np.where(own, idx - self.row0, 0)
        return (torch.from_numpy(own.astype(np.float64)).to(self.device),
                torch.from_numpy(loc).to(self.device), bool(own.any()))

    def _gather_rows(self, ind):
        """Feature rows of global indices, replicated on every rank (owners contribute, the rest adds zeros)."""
        if not self.collective:
            idx = torch.as_tensor(ind, dtype=torch.int64, device=self.device)
            return self.Xd.index_select(0, idx)       # every row is local: no ownership test, no host synchronisation
        mask, loc, any_own = self._owned(ind)
        if any_own:
            rows = self.Xd.index_select(0, loc) * mask[:, None]
        else:
            rows = torch.zeros((len(ind), self.ldx), dtype=torch.float64, device=self.device)
        sharding.all_reduce_sum(rows, self.group)
        return rows

    def gather_columns(self, mat, ind):
        """mat[:, local column of each global index] replicated on every rank ([rows, len(ind)]); `mat` holds one column
        per local data row (V, or covariance columns)."""
        mask, loc, any_own = self._owned(ind)
        if any_own:
            out = mat.index_select(1, loc) * mask[None, :]
        else:
            out = torch.zeros((mat.shape[0], len(loc)), dtype=torch.float64, device=self.device)
        if self.collective:
            sharding.all_reduce_sum(out, self.group)
        return out

    def _append_staged(self, matrix, slots, y):
        """_append for up to 16 samples whose feature rows are rows `slots` of a replicated device matrix (the batch state
        of the last fetch): ONE call (ital_gp_append: staging + Cholesky append in one launch, then the whitening sweep)
        instead of gather, copies, norms, a label upload and three calls."""
        c = len(slots)
        if self.m + c > self.cap:
            self._alloc(max(2 * self.cap, self.m + c))
        if getattr(self, "_ybuf", None) is None:
            self._ybuf = torch.empty(16, dtype=torch.float64, device=self.device)
        d = getattr(self, "_append_desc", None)
        key = (self.V.data_ptr(), self.mu.data_ptr(), self.cap)
        if d is None or d[0] != key:          # the pointers only change when the labelled-set buffers are re-allocated
            a = _lib.ItalAppendDesc()
            a.ldx, a.X, a.xnorm, a.n = self.ldx, _ptr(self.Xd), _ptr(self.xnorm), self.n
            a.XT, a.XTn, a.L, a.ldl, a.alpha, a.ybuf = _ptr(self.XT), _ptr(self.XTn), _ptr(self.L), self.cap, _ptr(self.alpha), \
                _ptr(self._ybuf)
            a.V, a.ldv, a.mu, a.s2, a.status = _ptr(self.V), self.ldv, _ptr(self.mu), _ptr(self.s2), _ptr(self.status)
            self._append_desc = d = (key, a)
        a = d[1]
        a.rows, a.m = _ptr(matrix), self.m
        a.var, a.length_scale, a.noise = float(self.var), float(self.length_scale), float(self.noise)
        a.lb.c = c
        for j in range(c):
            a.lb.slot[j] = int(slots[j])
            a.lb.y[j] = float(y[j])
        check(self._lib.ital_gp_append(ctypes.byref(a), _stream()))
        self.m += c

    def _append(self, rows, y):
        lib, st = self._lib, _stream()
        c_total = rows.shape[0]
        if self.m + c_total > self.cap:
            self._alloc(max(2 * self.cap, self.m + c_total))
        yd = torch.from_numpy(np.ascontiguousarray(y)).to(self.device)
        for c0 in range(0, c_total, 16):
            c = min(16, c_total - c0)
            m = self.m
            self.XT[m:m + c] = rows[c0:c0 + c]
            check(lib.ital_row_norms(_ptr(self.XT[m:m + c]), c, self.ldx, _ptr(self.XTn[m:m + c]), st))
            check(lib.ital_chol_append(_ptr(self.XT), _ptr(self.XTn), self.ldx, _ptr(self.L), self.cap, _ptr(self.alpha),
                                       _ptr(yd[c0:c0 + c]), m, c, float(self.var), float(self.length_scale),
                                       float(self.noise), _ptr(self.status), st))
            L21 = self.L[m:]
            check(lib.ital_whiten_append(_ptr(self.Xd), _ptr(self.xnorm), self.n, self.ldx, _ptr(self.XT[m:m + c]),
                                         _ptr(self.XTn[m:m + c]), c, _ptr(L21), self.cap,
                                         L21.data_ptr() + 8 * m, _ptr(self.alpha[m:m + c]), _ptr(self.V), self.ldv, m,
                                         float(self.var), float(self.length_scale), _ptr(self.mu), _ptr(self.s2), st))
            self.m += c

    def check_status(self, status=None):
        """Raises if a kernel flagged a numerical failure.  `status`: the word as already downloaded (e.g. the OR over all
        ranks that the selection step returns); otherwise this rank's word is read (synchronises).  Bit 1 comes from the
        replicated Cholesky append and is therefore the same on every rank."""
        s = int(self.status.item()) if status is None else int(status)
        if s & 1:
            raise np.linalg.LinAlgError("kernel matrix of the labelled samples is not positive definite "
                                        "(the reference would warn at gp.py:31)")
        if s & 2:
            raise np.linalg.LinAlgError("singular conditional covariance in the orthant integrator "
                                        "(duplicate samples in the batch?)")

    # ------------------------------------------------------------------ prediction
    def _full(self, t):
        """Local shard vector -> full-length numpy array (all ranks; a collective when the rows are sharded)."""
        return self._full_device(t).cpu().numpy()

    def _full_device(self, t, keep=None):
        """Local shard vector -> full-length device vector, replicated on every rank (one all-gather of equal-sized,
        padded shards when the rows are sharded).  `keep`: name under which the send / receive buffers are kept between
        calls (the means are replicated after every update: no allocation, and no copy at all when the shards are equal)."""
        loc = t[: self.n]
        if not self.collective:
            return loc
        pad = (self.n_total + self.world - 1) // self.world
        bufs = self._gather_bufs.get(keep) if keep else None
        if bufs is None:
            bufs = (torch.zeros(pad, dtype=loc.dtype, device=self.device),
                    torch.empty((self.world, pad), dtype=loc.dtype, device=self.device))
            if keep:
                self._gather_bufs[keep] = bufs
        send, recv = bufs
        if self.n == pad and loc.is_contiguous():
            send = loc                       # equal shards: the local vector is the send buffer, no staging copy
        else:
            send[: self.n] = loc
        sharding.gather_records(send, recv, self.group)
        if pad * self.world == self.n_total:
            return recv.view(-1)
        parts = []
        for r in range(self.world):
            a, b = sharding.row_range(self.n_total, self.world, r)
            parts.append(recv[r, : b - a])
        return torch.cat(parts)

    def _replicate_mean(self):
        """Called by every rank at the end of an update: the predictive means of ALL samples as a device vector on every
        rank (asynchronous, on the stream).  What reads them later -- `rel_mean`, top_results(), the candidate restriction
        of `top_candidates` -- is then a local operation: reading an attribute on one rank only cannot dead-lock the
        others (the reference refreshes `rel_mean` inside update() too, retrieval_base.py:58,120)."""
        self.mu_all = self._full_device(self.mu, keep="mu")
        self._mean_host = None

    def mean_host(self):
        """Predictive mean of every sample as a numpy array (downloaded on first use after an update; no collective)."""
        if self._mean_host is None:
            self._mean_host = self.mu_all.cpu().numpy()
        return self._mean_host

    def topk_mean(self, k):
        """Indices of the k samples of largest predictive mean, descending (NaN first, ties by descending index), selected
        on the device (ital_topk): nothing of size N is downloaded.  Local operation on the replicated means."""
        k = int(k)
        n = self.n_total
        if not (1 <= k <= min(_lib.ITAL_TOPK_MAX, n)):
            raise ValueError("k outside 1..min(%d, number of samples)" % _lib.ITAL_TOPK_MAX)
        if getattr(self, "_topk_work", None) is None:
            self._topk_work = torch.empty(int(self._lib.ital_topk_workspace()), dtype=torch.uint8, device=self.device)
        vals = torch.empty(k, dtype=torch.float64, device=self.device)
        idx = torch.empty(k, dtype=torch.int64, device=self.device)
        check(self._lib.ital_topk(_ptr(self.mu_all), n, 0, k, _ptr(vals), _ptr(idx), _ptr(self._topk_work), _stream()))
        return idx.cpu().numpy()

    def predict_stored(self, ind=None, cov_mode=None):
        """Predictive mean / variance / covariance of samples of the data matrix (reference gp.py:203-232)."""
        if self.m == 0:
            raise RuntimeError("the GP has not been fitted: call fit()/update() first "
                               "(the reference fails with an AttributeError at gp.py:222)")
        mean = self.mean_host()
        if ind is not None:
            ind = np.asarray(ind, dtype=np.int64)
            mean = mean[ind]
        if cov_mode is None:
            return mean
        if cov_mode == "diag":
            var = np.maximum(0, self._full(self.s2))
            return mean, (var if ind is None else var[ind])
        if cov_mode == "full":
            if ind is None:
                raise ValueError("full covariance of all samples is an N x N matrix; pass `ind`")
            return mean, self._cov_full(ind)
        raise ValueError("cov_mode must be None, 'diag' or 'full'")

    def _cov_full(self, ind):
        """K[S,S] - V[:,S]^T V[:,S] for a short index list S: feature rows and whitened columns of S are replicated (owners
        contribute), every rank then forms the |S| x |S| block itself (ital_cov_block, FP64 MFMA)."""
        ind = [int(i) for i in ind]
        ns = len(ind)
        rows = self._gather_rows(ind)
        norms = torch.empty(ns, dtype=torch.float64, device=self.device)
        check(self._lib.ital_row_norms(_ptr(rows), ns, self.ldx, _ptr(norms), _stream()))
        lds = _pad16(ns)
        Vs = torch.zeros((max(self.m, 1), lds), dtype=torch.float64, device=self.device)
        Vs[: self.m, :ns] = self.gather_columns(self.V[: max(self.m, 1)], ind)[: self.m]
        out = torch.empty((ns, lds), dtype=torch.float64, device=self.device)
        check(self._lib.ital_cov_block(_ptr(rows), _ptr(norms), ns, _ptr(rows), _ptr(norms), ns, self.ldx, _ptr(Vs), lds,
                                       _ptr(Vs), lds, self.m, float(self.var), float(self.length_scale), _ptr(out), lds,
                                       _stream()))
        return out[:, :ns].cpu().numpy()

    def updated_prediction(self, ind, y, pred_ind, cov_mode=None):
        """Prediction for `pred_ind` after a simulated update with (ind, y), without updating (reference gp.py:295-344).
        The scorers do this inside their kernels (closed form on the fed-back block); this API call, which the reference
        exposes to callers, applies the same closed form on the host to the posterior block the device computes:
        W = (Sigma_FF + noise I)^-1, mu' = mu_P + Sigma_PF W (y - mu_F), Sigma' = Sigma_PP - Sigma_PF W Sigma_FP."""
        ind = [int(i) for i in ind]
        pred = [int(i) for i in pred_ind]
        y = np.asarray(y, dtype=np.float64).reshape(-1)
        mean, cov = self.predict_stored(pred + ind, cov_mode="full")
        n = len(pred)
        s_ff = cov[n:, n:] + self.noise * np.eye(len(ind))
        s_pf = cov[:n, n:]
        sol = np.linalg.solve(s_ff, np.column_stack((y - mean[n:], s_pf.T)))
        new_mean = mean[:n] + s_pf @ sol[:, 0]
        if cov_mode is None:
            return new_mean
        new_cov = cov[:n, :n] - s_pf @ sol[:, 1:]
        if cov_mode == "full":
            return new_mean, new_cov
        if cov_mode == "diag":
            return new_mean, np.maximum(0, np.diag(new_cov))
        raise ValueError("cov_mode must be None, 'diag' or 'full'")

    def rbf_cols(self, ind):
        """Kernel columns k(x_j, X) for a short index list (what replaces slicing K_all)."""
        rows = self._gather_rows([int(i) for i in ind])
        norms = torch.empty(len(ind), dtype=torch.float64, device=self.device)
        check(self._lib.ital_row_norms(_ptr(rows), len(ind), self.ldx, _ptr(norms), _stream()))
        out = torch.empty((len(ind), self.ldv), dtype=torch.float64, device=self.device)
        for c0 in range(0, len(ind), 16):
            c = min(16, len(ind) - c0)
            check(self._lib.ital_rbf_cols(_ptr(self.Xd), _ptr(self.xnorm), self.n, self.ldx, _ptr(rows[c0:c0 + c]),
                                          _ptr(norms[c0:c0 + c]), c, float(self.var), float(self.length_scale),
                                          _ptr(out[c0:c0 + c]), self.ldv, _stream()))
        return out[:, : self.n]

    def predict(self, X, cov_mode=None):
        """Predictive mean (and variance / full covariance) for external samples (reference gp.py:264-292).  Every rank
        computes it for all of X (the labelled-set state is replicated): no collective."""
        if self.m == 0:
            raise RuntimeError("the GP has not been fitted")
        if cov_mode not in (None, "diag", "full"):
            raise ValueError("cov_mode must be None, 'diag' or 'full'")
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        nt = X.shape[0]
        dev = self.device
        Xt = torch.zeros((nt, self.ldx), dtype=torch.float64, device=dev)
        Xt[:, : self.d] = torch.from_numpy(np.ascontiguousarray(X)).to(dev)
        ldvt = _pad16(nt)
        mean = torch.empty(nt, dtype=torch.float64, device=dev)
        pvar = torch.empty(nt, dtype=torch.float64, device=dev)
        xtn = torch.empty(nt, dtype=torch.float64, device=dev)
        Vt = torch.empty((self.m, ldvt), dtype=torch.float64, device=dev)
        check(self._lib.ital_predict(_ptr(Xt), nt, self.ldx, _ptr(self.XT), _ptr(self.XTn), self.m, _ptr(self.L), self.cap,
                                     _ptr(self.alpha), float(self.var), float(self.length_scale), _ptr(mean), _ptr(pvar),
                                     _ptr(xtn), _ptr(Vt), ldvt, 1, _stream()))
        if cov_mode == "diag":
            return mean.cpu().numpy(), pvar.cpu().numpy()
        if cov_mode == "full":
            cov = torch.empty((nt, ldvt), dtype=torch.float64, device=dev)
            check(self._lib.ital_cov_block(_ptr(Xt), _ptr(xtn), nt, _ptr(Xt), _ptr(xtn), nt, self.ldx, _ptr(Vt), ldvt,
                                           _ptr(Vt), ldvt, self.m, float(self.var), float(self.length_scale), _ptr(cov),
                                           ldvt, _stream()))
            return mean.cpu().numpy(), cov[:, :nt].cpu().numpy()
        return mean.cpu().numpy()
