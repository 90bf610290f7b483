"""ITAL on MI355X: host-side mirror of reference ital/ital.py `ITAL` (drop-in learner).

`fetch_unlabelled(k)` keeps the reference's greedy batch construction (ital.py:84-134) but every stage is a HIP
kernel enqueued on one stream, with no host round trip inside a round:

    for step t = 1..k:   score all live candidates   ital_score_step      (replaces Pool.map of ital.py:124-128)
                         local arg-max + record      ital_select_local    (np.argmax, ital.py:130)
                         [all-gather of one record per rank over RCCL when world > 1]
                         winner -> batch state       ital_select_resolve  (append + del, ital.py:131-132)
                         next cross-covariance col.  ital_cross_cov_cols  (predict_cov_batch, ital.py:586)

Candidates are sharded by rows across ranks; the per-step exchange is ONE fixed-size record per rank.
"""
import ctypes

import numpy as np
import torch

from . import _lib, mvn_stream, sharding
from ._lib import ITAL_JUMP_BITS, ITAL_MAX_T, ITAL_REC_HEADER, ItalBatch, ItalScoreDesc, check
from ._batch import make_batch_buffers
from .gp import _pad16, _ptr, _stream
from .retrieval_base import ActiveRetrievalBase

_LABEL_MODES = {"mean": 0, "optimistic": 1, "pessimistic": 2}


class ITAL(ActiveRetrievalBase):
    """Information-theoretic Active Learning for information retrieval (reference ital/ital.py:12-134).

    Constructor arguments are the reference's (ital.py:15-81).  Options that this round's device scorer does not
    cover raise NotImplementedError at fetch time instead of silently taking another path: a non-perfect user
    model (label_prob < 1 or mistake_prob > 0), change_estimation_subset != 0, clip_cov and the Monte-Carlo
    switches.  `parallelized` is accepted and ignored (the GPU is the parallelism).
    """

    def __init__(self, data=None, queries=[], length_scale=0.1, var=1.0, noise=1e-6, label_prob=1.0,
                 mistake_prob=0.0, top_candidates=None, change_estimation_subset=0, clip_cov=0,
                 label_estimation='mean', monte_carlo_num_rel=None, monte_carlo_num_fb=None, parallelized=True, *,
                 device=None, rank=0, world=1, group=None):
        ActiveRetrievalBase.__init__(self, data, queries, length_scale, var, noise, device=device, rank=rank,
                                     world=world, group=group)
        self.label_prob = label_prob
        self.mistake_prob = mistake_prob
        self.top_candidates = top_candidates
        self.change_estimation_subset = change_estimation_subset
        self.clip_cov = clip_cov
        self.label_estimation = label_estimation
        self.monte_carlo_num_rel = monte_carlo_num_rel
        self.monte_carlo_num_fb = monte_carlo_num_fb
        self.parallelized = parallelized
        self.eps = 1e-12  # reference ital/ital.py:144
        self.last_scores = None  # per greedy step: device tensor of MI per list position (diagnostics/tests)
        self.keep_scores = False
        self.profile = None      # list to receive (stage, t, size, start_event, end_event) per launch (bench.py)
        self._fetch_bufs = None

    # ------------------------------------------------------------------ helpers
    def _unsupported(self):
        if not (self.label_prob >= 1 and self.mistake_prob <= 0):
            return "non-perfect user models (label_prob < 1 or mistake_prob > 0)"
        if self.change_estimation_subset is None or self.change_estimation_subset > 0:
            return "change_estimation_subset"
        if self.clip_cov:
            return "clip_cov"
        if self.monte_carlo_num_rel is not None or self.monte_carlo_num_fb is not None:
            return "monte-carlo enumeration"
        if self.label_estimation not in _LABEL_MODES:
            return "label_estimation=%r" % (self.label_estimation,)
        return None

    def _mark(self, stage=None, t=0, size=0, start=None):
        """HIP event on the launch stream (only when bench.py asked for per-kernel timings)."""
        if self.profile is None:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        if start is not None:
            self.profile.append((stage, t, size, start, ev))
        return ev

    def _buffers(self, kmax):
        gp = self.gp
        b = self._fetch_bufs
        if b is not None and b["kmax"] >= kmax and b["ldw"] == gp.cap:
            return b
        b = make_batch_buffers(gp.device, kmax, gp.ldx, gp.cap, gp.ldv, gp.world)
        self._fetch_bufs = b
        return b

    def _candidate_list(self):
        """Candidate list in the reference's order (ital.py:98, :111-117)."""
        candidates = self.get_unseen()
        if self.top_candidates is not None:
            top_candidates = self.top_candidates
            if isinstance(self.top_candidates, float):
                labelled = len(self.queries) + len(self.relevant_ids) + len(self.irrelevant_ids)
                top_candidates = min(len(candidates), int(self.top_candidates * labelled))
            if (top_candidates > 0) and (top_candidates < len(candidates)):
                cand_arr = np.asarray(candidates)
                top_ind = np.argpartition(self.rel_mean[cand_arr], -top_candidates)[-top_candidates:]
                candidates = cand_arr[top_ind].tolist()
        return candidates

    # ------------------------------------------------------------------ the hot path
    def fetch_unlabelled(self, k, show_progress=False):
        """Selects k unlabelled samples by greedy maximisation of mutual information (reference ital.py:84-134).

        Returns the list of selected sample indices (python ints, selection order)."""
        why = self._unsupported()
        if why is not None:
            raise NotImplementedError("ital_amd device scorer: %s is not implemented yet" % why)
        gp = self.gp
        if gp.m == 0:
            raise RuntimeError("fetch_unlabelled() needs a fitted relevance model: call update() first or pass queries "
                               "(the reference fails with an AttributeError at gp.py:222)")
        candidates = self._candidate_list()
        k = min(int(k), len(candidates))
        if k <= 0:
            return []
        if k > ITAL_MAX_T:
            raise NotImplementedError("batches larger than %d need the monte-carlo enumeration (reference ital.py:293-297)"
                                      % ITAL_MAX_T)
        lib = _lib.lib()
        dev = gp.device
        with torch.cuda.device(dev):
            b = self._buffers(k)
            st = _stream()
            # ---- candidate shard of this rank (list positions keep their global numbering)
            cand = np.asarray(candidates, dtype=np.int64)
            loc_rows, pos_offset = sharding.shard_candidates(cand, gp.row0, gp.row1)
            n_loc = len(loc_rows)
            cand_d = torch.from_numpy((loc_rows - gp.row0).astype(np.int32)).to(dev) if n_loc else \
                torch.zeros(1, dtype=torch.int32, device=dev)
            alive = torch.ones(max(n_loc, 1), dtype=torch.uint8, device=dev)
            mi = torch.zeros(max(n_loc, 1), dtype=torch.float64, device=dev)
            self.last_scores = []
            stream = mvn_stream.GLOBAL
            n_alive = len(candidates)
            for t in range(1, k + 1):
                desc = ItalScoreDesc()
                desc.t = t
                desc.n_cand = n_loc
                desc.cand, desc.alive, desc.mu, desc.s2 = _ptr(cand_d), _ptr(alive), _ptr(gp.mu), _ptr(gp.s2)
                desc.C, desc.ldc = _ptr(b["C"]), gp.ldv
                desc.row_offset, desc.pos_offset = gp.row0, pos_offset
                desc.batch = b["batch"]
                desc.noise, desc.eps = float(self.noise), float(self.eps)
                desc.label_mode = _LABEL_MODES[self.label_estimation]
                desc.mi = _ptr(mi)
                desc.status = _ptr(gp.status)
                if t >= 3:
                    if t not in b["jump"]:
                        b["jump"][t] = torch.from_numpy(mvn_stream.jump_table(t, ITAL_JUMP_BITS)).to(dev)
                        b["vk"][t] = torch.from_numpy(mvn_stream.korobov_vk(t)).to(dev)
                    desc.jump, desc.vk = _ptr(b["jump"][t]), _ptr(b["vk"][t])
                    for j in range(6):
                        desc.seed[j] = stream.state[j]
                ev0 = self._mark()
                check(lib.ital_score_step(ctypes.byref(desc), st))
                self._mark("score", t, n_alive, ev0)
                if self.keep_scores:
                    self.last_scores.append(mi.clone())
                check(lib.ital_select_local(_ptr(mi), _ptr(cand_d), _ptr(alive), n_loc, pos_offset, gp.row0, gp.rank, 0,
                                            _ptr(gp.mu), _ptr(gp.s2), _ptr(gp.Xd), _ptr(gp.xnorm), gp.ldx, _ptr(gp.V),
                                            gp.ldv, gp.m, gp.cap, _ptr(b["C"]), gp.ldv, t - 1, b["kmax"], _ptr(b["work"]),
                                            _ptr(b["rec"]), st))
                recs = sharding.gather_records(b["rec"], b["rec_all"], gp.group) if gp.world > 1 else b["rec"]
                check(lib.ital_select_resolve(_ptr(recs), gp.world, b["rec_len"], gp.rank, 0, t - 1, b["batch"],
                                              _ptr(alive), _ptr(b["ret"]), st))
                if t < k:
                    slot = t - 1
                    ev0 = self._mark()
                    check(lib.ital_cross_cov_cols(_ptr(gp.Xd), _ptr(gp.xnorm), gp.n, gp.ldx, _ptr(b["XB"][slot]),
                                                  _ptr(b["XBn"][slot:]), 1, _ptr(b["VB"][slot]), gp.cap, _ptr(gp.V),
                                                  gp.ldv, gp.m, float(self.var), float(self.length_scale),
                                                  _ptr(b["C"][slot]), gp.ldv, st))
                    self._mark("cross_cov", t, gp.m, ev0)
                # the reference's serial loop has now consumed this many uniforms of mvndst's stream
                stream.advance(n_alive * (2 << t) * mvn_stream.draws_per_call(t))
                n_alive -= 1
            ret = b["ret"][:k].cpu().tolist()  # the only synchronisation of the round
        gp.check_status()
        return [int(i) for i in ret]
