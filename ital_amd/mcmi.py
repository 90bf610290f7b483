"""MCMI[min] on MI355X: host-side mirror of reference ital/mcmi.py `MCMI_min` (drop-in learner).

Guo & Greiner's optimistic active learner scores a candidate by the summed conditional entropy of ALL candidates
after a simulated update with the most favourable labelling of the batch (reference mcmi.py:101-124) and picks
the arg-min greedily (mcmi.py:69-79).  On the device:

    once per fetch:  gather the candidate block (features, whitened columns, mean, variance)   [replicated]
                     posterior covariance of the own candidates with the block  ital_cov_block (FP64 MFMA)
    per greedy step: ital_mcmi_score_step -> ital_select_local(arg-min) -> [record all-gather] ->
                     ital_select_resolve -> ital_cross_cov_cols (the picked member's covariance column)

With several ranks the scored candidates (rows i of the pairwise objective) are split across ranks, the block they
are scored against is replicated.
"""
import ctypes

import numpy as np
import torch

from . import _lib, sharding
from ._batch import make_batch_buffers
from ._lib import ITAL_MAX_T, ItalMcmiDesc, check
from .gp import _pad16, _ptr, _stream
from .retrieval_base import ActiveRetrievalBase


class MCMI_min(ActiveRetrievalBase):
    """Constructor arguments as reference ital/mcmi.py:21-45; `parallelized` is accepted and ignored."""

    #: refuse to materialise a covariance block larger than this many bytes per rank (use `subsample`, as the
    #: reference's configs do: usps.conf:25-26)
    max_cov_bytes = 96 << 30

    def __init__(self, data=None, queries=[], length_scale=0.1, var=1.0, noise=1e-6, subsample=None,
                 parallelized=True, *, device=None, rank=0, world=1, group=None):
        ActiveRetrievalBase.__init__(self, data, queries, length_scale, var, noise, device=device, rank=rank,
                                     world=world, group=group)
        self.subsample = subsample
        self.parallelized = parallelized
        self.eps = 1e-12  # reference ital/mcmi.py:88
        self.candidates = []
        self.keep_scores = False
        self.round_call = True        # one rank: the whole round as one call below the C ABI (False: step by step from here)
        self.last_scores = None
        self.profile = None
        self.event_pool = []
        self._block_bufs = None
        self._fetch_bufs = None

    @property
    def candidates(self):
        """The candidate list the last fetch left behind (reference mcmi.py:57-66, :79), as a list; formed on first read."""
        c = self._candidates
        if isinstance(c, tuple):
            self._candidates = c = np.delete(c[0], c[1]).tolist()
        return c

    @candidates.setter
    def candidates(self, value):
        self._candidates = value

    def _mark(self, stage=None, t=0, size=0, start=None):
        if self.profile is None:
            return None
        ev = self.event_pool.pop() if self.event_pool else torch.cuda.Event(enable_timing=True)
        ev.record()
        if start is not None:
            self.profile.append((stage, t, size, start, ev))
        return ev

    def _gather_block(self, cand):
        """Features, squared norms, whitened columns, mean and variance of the candidate block on every rank."""
        gp = self.gp
        dev = gp.device
        nc = len(cand)
        ldc = _pad16(nc)
        # block buffers are kept between fetches of the same shape (a round at the reference's subsample of 1000 is 0.2 ms of
        # kernels: allocations and fills are most of its host time); the padding stays zero, the rest is overwritten
        key = (nc, gp.ldx, gp.cap)
        if self._block_bufs is None or self._block_bufs[0] != key:
            self._block_bufs = (key, torch.zeros((nc, gp.ldx), dtype=torch.float64, device=dev),
                                torch.zeros((gp.cap, ldc), dtype=torch.float64, device=dev),
                                torch.zeros((3, nc), dtype=torch.float64, device=dev))
        _, Xc, Vc, vec = self._block_bufs
        # one launch: rows, whitened columns, norms, means and variances of the listed samples (zeros for samples of other
        # ranks); the list travels as one small upload
        cand_d = torch.from_numpy(np.ascontiguousarray(cand, dtype=np.int64)).to(dev)
        check(_lib.lib().ital_gather_block(_ptr(cand_d), nc, gp.row0, gp.n, _ptr(gp.Xd), _ptr(gp.xnorm), gp.ldx, _ptr(gp.V),
                                           gp.ldv, gp.m, _ptr(gp.mu), _ptr(gp.s2), _ptr(Xc), _ptr(Vc), ldc, _ptr(vec[0]),
                                           _ptr(vec[1]), _ptr(vec[2]), _stream()))
        if gp.collective:
            for buf in (Xc, Vc, vec):
                sharding.all_reduce_sum(buf, gp.group)
        return Xc, Vc, ldc, vec[0], vec[1], vec[2]

    def _round(self, k, cand, b, cov, pos_d, alive, ce, Xc, Vc, ldc, xnc, muc, s2c, st):
        """One rank: covariance block, k scoring / arg-min steps and k - 1 covariance columns enqueued by ONE call
        (ital_mcmi_round): at the reference's subsample of 1000 a step is 20 - 90 us of kernels and the launches of a Python
        host are 6 us apart."""
        lib = _lib.lib()
        gp = self.gp
        dev = gp.device
        nc = len(cand)
        r = b.get("mcmi_round")
        if r is None:
            b["mcmi_round"] = r = _lib.ItalMcmiRoundDesc()
        d = r.step
        r.k = k
        d.n_i, d.pos_offset, d.n_all = nc, 0, nc
        d.alive, d.mu, d.s2 = _ptr(alive), _ptr(muc), _ptr(s2c)
        d.cov, d.ld_cov, d.C, d.ldc = _ptr(cov), ldc, _ptr(b["C"]), ldc
        d.batch = b["batch"]
        d.noise, d.eps, d.ce = float(self.noise), float(self.eps), _ptr(ce)
        d.work, d.work_doubles = None, 0
        if k >= 5:
            want = int(lib.ital_mcmi_workspace(ITAL_MAX_T, nc))
            w = b.get("mcmi_work")
            if w is None or w.numel() < want:
                b["mcmi_work"] = w = torch.empty(want, dtype=torch.float64, device=dev)
            d.work, d.work_doubles = _ptr(w), w.numel()
        r.Xc, r.xnc, r.ldx, r.Vc, r.ldv, r.m, r.ldw = _ptr(Xc), _ptr(xnc), gp.ldx, _ptr(Vc), ldc, gp.m, gp.cap
        r.var, r.length_scale = float(self.var), float(self.length_scale)
        r.pos, r.status, r.record, r.ret, r.begin = _ptr(pos_d), _ptr(gp.status), _ptr(b["rec"]), _ptr(b["ret"]), 1
        check(lib.ital_mcmi_round(ctypes.byref(r), st))
        self.last_scores = []
        host = b["ret"].cpu().tolist()            # block positions + status word; the only synchronisation of the round
        picked = host[:k]
        gp.check_status(host[b["kmax"]])
        ret = [int(cand[p]) for p in picked]
        self._last_batch = (b, ret)
        self._candidates = (cand, picked)                    # as `del self.candidates[min_ind]` per pick (mcmi.py:79)
        return ret

    def fetch_unlabelled(self, k, show_progress=False):
        """Fetches a batch of unlabelled samples (reference ital/mcmi.py:48-81); list of python ints."""
        gp = self.gp
        if gp.m == 0:
            raise RuntimeError("fetch_unlabelled() needs a fitted relevance model: call update() first or pass queries")
        cand = self._unseen_array()
        if self.subsample and (self.subsample < len(cand)):
            # same call on the global numpy RNG as the reference (mcmi.py:61-63; an array draws as a list does)
            cand = np.random.choice(cand, self.subsample, replace=False)
        if len(cand) < k:
            k = len(cand)
        if k <= 0:
            self.candidates = cand.tolist()
            return []
        if k > ITAL_MAX_T:
            raise NotImplementedError("batches larger than %d are not enumerated on the device" % ITAL_MAX_T)
        lib = _lib.lib()
        dev = gp.device
        self._last_batch = None      # published after the round's successful download only
        cand = np.asarray(cand, dtype=np.int64)
        nc = len(cand)
        with torch.cuda.device(dev):
            st = _stream()
            Xc, Vc, ldc, xnc, muc, s2c = self._gather_block(cand)
            i0, i1 = sharding.row_range(nc, gp.world, gp.rank)
            n_i = i1 - i0
            if max(n_i, 1) * ldc * 8 > self.max_cov_bytes:
                raise MemoryError("MCMI_min: %d x %d covariance block; pass subsample= (reference configs use 1000)"
                                  % (n_i, nc))
            key = (k, gp.ldx, gp.cap, ldc, n_i, i0)
            if self._fetch_bufs is None or self._fetch_bufs[0] != key:
                self._fetch_bufs = (key, make_batch_buffers(dev, k, gp.ldx, gp.cap, ldc, gp.world),
                                    torch.empty((max(n_i, 1), ldc), dtype=torch.float64, device=dev),
                                    torch.arange(i0, max(i1, i0 + 1), dtype=torch.int32, device=dev),
                                    torch.empty(max(n_i, 1), dtype=torch.uint8, device=dev),
                                    torch.empty(max(n_i, 1), dtype=torch.float64, device=dev))
            _, b, cov, pos_d, alive, ce = self._fetch_bufs
            if (self.round_call and not gp.collective and self.profile is None and not self.keep_scores
                    and k <= nc <= (1 << 18)):
                return self._round(k, cand, b, cov, pos_d, alive, ce, Xc, Vc, ldc, xnc, muc, s2c, st)
            alive.fill_(1)
            b["ret"][b["kmax"]:].zero_()       # the selection steps OR the status word into this slot
            ev0 = self._mark()
            if n_i:
                check(lib.ital_cov_block(_ptr(Xc[i0:]), _ptr(xnc[i0:]), n_i, _ptr(Xc), _ptr(xnc), nc, gp.ldx,
                                         Vc.data_ptr() + 8 * i0, ldc, _ptr(Vc), ldc, gp.m, float(self.var),
                                         float(self.length_scale), _ptr(cov), ldc, st))
            self._mark("cov_block", 0, nc, ev0)
            self.last_scores = []
            for t in range(1, k + 1):
                desc = ItalMcmiDesc()
                desc.t, desc.n_i, desc.pos_offset, desc.n_all = t, n_i, i0, nc
                desc.alive, desc.mu, desc.s2 = _ptr(alive), _ptr(muc), _ptr(s2c)
                desc.cov, desc.ld_cov, desc.C, desc.ldc = _ptr(cov), ldc, _ptr(b["C"]), ldc
                desc.batch = b["batch"]
                desc.noise, desc.eps = float(self.noise), float(self.eps)
                desc.ce = _ptr(ce)
                if t >= 5 and n_i:
                    # batches of 5 .. 8: preparation kernel + one workgroup per (candidate, group of label patterns)
                    want = int(lib.ital_mcmi_workspace(t, n_i))
                    w = b.get("mcmi_work")
                    if w is None or w.numel() < want:
                        b["mcmi_work"] = w = torch.empty(int(lib.ital_mcmi_workspace(ITAL_MAX_T, n_i)), dtype=torch.float64,
                                                         device=dev)
                    desc.work, desc.work_doubles = _ptr(w), w.numel()
                ev0 = self._mark()
                check(lib.ital_mcmi_score_step(ctypes.byref(desc), st))
                self._mark("mcmi_score", t, nc - (t - 1), ev0)
                if self.keep_scores:
                    self.last_scores.append(ce.clone())
                if not gp.collective and n_i <= (1 << 18):
                    check(lib.ital_select_fused(_ptr(ce), _ptr(pos_d), _ptr(alive), n_i, i0, None, 0, gp.rank, 1, _ptr(muc),
                                                _ptr(s2c), _ptr(Xc), _ptr(xnc), gp.ldx, _ptr(Vc), ldc, gp.m, gp.cap,
                                                _ptr(b["C"]), ldc, t - 1, t - 1, b["batch"], _ptr(gp.status), _ptr(b["rec"]),
                                                _ptr(b["ret"]), st))
                else:
                    check(lib.ital_select_local(_ptr(ce), _ptr(pos_d), _ptr(alive), n_i, i0, None, 0, gp.rank, 1, _ptr(muc),
                                                _ptr(s2c), _ptr(Xc), _ptr(xnc), gp.ldx, _ptr(Vc), ldc, gp.m, gp.cap,
                                                _ptr(b["C"]), ldc, t - 1, b["kmax"], _ptr(gp.status), _ptr(b["work"]),
                                                _ptr(b["rec"]), st))
                    recs = sharding.gather_records(b["rec"], b["rec_all"], gp.group) if gp.collective else b["rec"]
                    check(lib.ital_select_resolve(_ptr(recs), gp.world, b["rec_len"], gp.rank, 1, t - 1, b["batch"],
                                                  _ptr(alive), _ptr(b["ret"]), st))
                if t < k:
                    slot = t - 1
                    check(lib.ital_cross_cov_cols(_ptr(Xc), _ptr(xnc), nc, gp.ldx, _ptr(b["XB"][slot]),
                                                  _ptr(b["XBn"][slot:]), 1, _ptr(b["VB"][slot]), gp.cap, _ptr(Vc), ldc,
                                                  gp.m, float(self.var), float(self.length_scale), _ptr(b["C"][slot]),
                                                  ldc, st))
            host = b["ret"].cpu().tolist()        # block positions + status word; the only synchronisation of the round
        picked = host[:k]
        gp.check_status(host[b["kmax"]])
        ret = [int(cand[p]) for p in picked]
        self._last_batch = (b, ret)
        self._candidates = (cand, picked)                    # as `del self.candidates[min_ind]` per pick (mcmi.py:79)
        return ret
