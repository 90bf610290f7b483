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
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _lib, mvn_stream, sharding
from ._lib import (ITAL_GENERIC_MAX_CALLS, ITAL_GENERIC_MAX_DIM, ITAL_GENERIC_MAX_REL, ITAL_JUMP_BITS, ITAL_MAX_T, ITAL_REC_HEADER, ItalBatch,
                   ItalGscoreDesc, ItalScoreDesc, check)
from ._batch import make_batch_buffers
from .gp import _pad16, _ptr, _stream
from .retrieval_base import ActiveRetrievalBase, UnseenList

_LABEL_MODES = {"mean": 0, "optimistic": 1, "pessimistic": 2}
_HOST_THREADS = max(1, min(int(os.environ.get("ITAL_HOST_THREADS", 16)), os.cpu_count() or 1))   # host share of one GPU (Monte-Carlo pattern sampling)
# pattern sampling: ranges of candidates per greedy step (host / GPU overlap), their minimum size, and the number of
# variables from which a step is split at all (below, the step's lattice sums are shorter than the host's decompositions:
# measured at 125 000 x 512, nothing to hide behind; on shards of 262 144 candidates and more from 7 variables on)
_MC_CHUNKS, _MC_CHUNK_MIN, _MC_CHUNK_FROM = (int(os.environ.get("ITAL_MC_CHUNKS", 4)), int(os.environ.get("ITAL_MC_CHUNK_MIN", 8192)),
                                             int(os.environ.get("ITAL_MC_CHUNK_FROM", 10)))      # (environment: experiments only)
_POOL = None


def _range_cuts(lo, hi, chunks, smallest=None):
    """Boundaries of the ranges a step of sampled patterns is scored in: sizes 1 : 2 : 4 : ... over [lo, hi) (the first range is
    decomposed while the GPU idles, every later one under the lattice sums of the range before), none shorter than `smallest`
    (default _MC_CHUNK_MIN) unless the whole span is.  Ascending int64 array, first entry lo, last entry hi."""
    smallest = _MC_CHUNK_MIN if smallest is None else smallest
    span = hi - lo
    cuts = lo + (span * ((1 << np.arange(chunks + 1)) - 1)) // ((1 << chunks) - 1)
    return np.unique(np.concatenate(([lo], cuts[cuts - lo >= min(smallest, span)], [hi])).astype(np.int64))


def _host_pool():
    global _POOL
    if _POOL is None:
        _POOL = ThreadPoolExecutor(_HOST_THREADS)
    return _POOL
_FUSED_SELECT_MAX = _lib.ITAL_ROUND_MAX_CAND   # per rank, up to this many candidates: arg-max + record (+ resolve) inside the scoring launch
_FUSED_LAUNCH_MAX = 1 << 18   # one rank, up to this many candidates: ital_select_fused (one workgroup) as the separate selection launch


class ITAL(ActiveRetrievalBase):
    """Information-theoretic Active Learning for information retrieval (reference ital/ital.py:12-134).

    Constructor arguments are the reference's (ital.py:15-81).  The perfect-user default runs on the specialised
    scorer (`ital_score_step`); every other option (user models, change-estimation subset, clip_cov, label_estimation,
    the Monte-Carlo switches) on the general one (`ital_score_generic`).  Configurations beyond the device limits
    (`_unsupported`) raise NotImplementedError at fetch time instead of silently taking another path.  `parallelized`
    is accepted and ignored (the GPU is the parallelism).
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
        self.force_generic = False  # route the perfect-user case through the general scorer too (cross-check in tests)
        self._ce_subset = None
        self.qmc_work_bytes = int(os.environ.get("ITAL_QMC_WORK_BYTES", 1 << 30))   # cap of the lattice scorer's workspace
        self._last_batch = None  # (batch buffers, picks) of the last fast-path round: update() reuses the winners' rows
        self.pair_counter = None  # optional int64 device tensor [1]: the general scorer adds its evaluated (Phi, Phi^-1) pairs
        self.generic_pipeline = True   # False: the general scorer always runs as its single kernel (cross-check in tests)
        self.event_pool = []     # pre-created timing events (bench.py)
        self.profile = None      # list to receive (stage, t, size, start_event, end_event) per launch (bench.py)
        self._fetch_bufs = None
        self.profile_steps = None      # round path: greedy steps whose lattice sums are bracketed by events (None: all)
        self.round_call = True        # up to ITAL_ROUND_MAX_CAND candidates per rank: a whole round through ital_fetch_round (False: step by step from Python)
        self._dev_list = None         # the candidate list the device holds: (buffers, UnseenList and its version, picks flagged dead)
        self.select_in_scorer = True  # False: the selection of a greedy step always runs as a launch of its own (cross-check in tests)
        self.mc_walk = [0, 0, 0.0]   # Monte-Carlo pattern sampling: standard normals computed / skipped, host seconds
        self.host_clock = None       # dict(gap_s=0.0, gaps=0, enqueue_s=0.0, t_download=None): host time of the round path (bench.py)

    # ------------------------------------------------------------------ helpers
    def _perfect_user(self):
        return self.label_prob >= 1 and self.mistake_prob <= 0

    def _fb_mode(self):
        """fb_mode of ital_gscore_desc: which simulated feedback the scorer enumerates (reference ital.py:300-342)."""
        return 0 if self._perfect_user() else (1 if self.label_prob >= 1 else 2)

    def _subset_mode(self):
        return self.change_estimation_subset is None or self.change_estimation_subset > 0

    def _mc_plan(self, nr, fb_mode):
        """Which enumerations the reference replaces by sampling at a step with nr enumerated variables
        (ital.py:293-297, :318-337): (rel sampled?, patterns, feedback sampled?, feedback configurations)."""
        num_rel = nr * self.monte_carlo_num_rel if self.monte_carlo_num_rel is not None else None
        rel_mc = num_rel is not None and not (2 ** (nr - 1) < num_rel)
        npat = num_rel if rel_mc else 2 ** nr
        num_fb = nr * self.monte_carlo_num_fb if self.monte_carlo_num_fb is not None else None
        if fb_mode == 3:                 # entropy objective: no simulated feedback at all
            fb_mc, nfb = False, 0
        elif fb_mode == 0:
            fb_mc, nfb = False, 1
        elif fb_mode == 1:
            fb_mc = num_fb is not None and not (2 ** (nr - 1) < num_fb)
            nfb = num_fb if fb_mc else 2 ** nr
        else:
            fb_mc = num_fb is not None and not (3 ** nr < 2 * num_fb)
            nfb = num_fb if fb_mc else 3 ** nr - 1
        return rel_mc, npat, fb_mc, nfb

    def _unsupported(self, k, n_unseen=0):
        """Reason why the device scorers cannot run this configuration (None if they can)."""
        if self.label_estimation not in _LABEL_MODES:
            return "label_estimation=%r" % (self.label_estimation,)
        if self.change_estimation_subset is None:
            # the whole candidate set is the estimation subset (ital.py:103-104): every orthant spans all candidates
            if n_unseen > ITAL_GENERIC_MAX_DIM:
                return ("change_estimation_subset=None with %d candidates: orthants of that dimension (limit %d)"
                        % (n_unseen, ITAL_GENERIC_MAX_DIM))
            sub, max_dim = n_unseen, n_unseen
        else:
            sub = self.change_estimation_subset if self.change_estimation_subset > 0 else 0
            max_dim = sub + k
        if self._needs_generic():
            if max_dim > ITAL_GENERIC_MAX_DIM:
                return "orthant dimension %d (subset + batch) above %d" % (max_dim, ITAL_GENERIC_MAX_DIM)
            if k > ITAL_GENERIC_MAX_REL:
                return "batches larger than %d with the general scorer" % ITAL_GENERIC_MAX_REL
            fb_mode = self._fb_mode()
            for nr in range(1, k + 1):
                _, npat, _, nfb = self._mc_plan(nr, fb_mode)
                if npat * (2 + nfb) > ITAL_GENERIC_MAX_CALLS:
                    return ("%d orthant probabilities per candidate at greedy step %d: set monte_carlo_num_rel / "
                            "monte_carlo_num_fb (reference ital.py:293-297)" % (npat * (2 + nfb), nr))
        elif k > ITAL_MAX_T:
            return ("batches larger than %d with full enumeration: set monte_carlo_num_rel (reference ital.py:293-297)"
                    % ITAL_MAX_T)
        return None

    def _needs_generic(self):
        return (self._subset_mode() or not self._perfect_user() or self.force_generic or self._clip_active()
                or self.monte_carlo_num_rel is not None or self.monte_carlo_num_fb is not None)

    def _clip_active(self):
        """clip_cov only ever acts on orthants of more than 5 dimensions, between 0 and 1 (reference ital.py:360)."""
        return bool(self.clip_cov) and 0 < self.clip_cov < 1

    def _mark(self, stage=None, t=0, size=0, start=None):
        """HIP event on the launch stream (only when bench.py asked for per-kernel timings).  Events come from
        `event_pool` when the caller filled it (creating a timing event costs ~0.3 ms on a loaded host)."""
        if self.profile is None:
            return None
        ev = self.event_pool.pop() if self.event_pool else torch.cuda.Event(enable_timing=True)
        ev.record()
        if start is not None:
            self.profile.append((stage, t, size, start, ev))
        return ev

    def _event(self):
        """A timing event for the library to record (handle must exist: pool events are recorded once when the pool is
        made; a fresh one is recorded here)."""
        if self.event_pool:
            return self.event_pool.pop()
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def _buffers(self, kmax):
        gp = self.gp
        b = self._fetch_bufs
        if b is not None and b["kmax"] >= kmax and b["ldw"] == gp.cap and b["gp"] is gp:
            return b
        b = make_batch_buffers(gp.device, kmax, gp.ldx, gp.cap, gp.ldv, gp.world)
        b["gp"] = gp              # fit() on an existing learner builds a new GP (other row count / dimension): new buffers
        self._fetch_bufs = b
        self._dev_list = None
        return b

    def _candidate_list(self, candidates=None):
        """Candidate list in the reference's order (ital.py:98, :111-117) as an int64 array."""
        if candidates is None:
            candidates = self._unseen_array()
        # change-estimation subset: drawn from the unrestricted candidate list on the global numpy RNG (ital.py:103-108)
        if self.change_estimation_subset is None:
            self._ce_subset = [int(i) for i in candidates]
        elif self.change_estimation_subset > 0:
            self._ce_subset = sorted(int(i) for i in np.random.choice(
                candidates, min(len(candidates), self.change_estimation_subset), replace=False))
        else:
            self._ce_subset = None
        if self.top_candidates is not None:
            top_candidates = self.top_candidates
            if isinstance(self.top_candidates, float):
                labelled = len(self.queries) + len(self.relevant_ids) + len(self.irrelevant_ids)
                top_candidates = min(len(candidates), int(self.top_candidates * labelled))
            if (top_candidates > 0) and (top_candidates < len(candidates)):
                top_ind = np.argpartition(self.rel_mean[candidates], -top_candidates)[-top_candidates:]
                candidates = candidates[top_ind]
        return candidates

    # ------------------------------------------------------------------ the hot path
    def fetch_unlabelled(self, k, show_progress=False):
        """Selects k unlabelled samples by greedy maximisation of mutual information (reference ital.py:84-134).

        Returns the list of selected sample indices (python ints, selection order)."""
        gp = self.gp
        if gp.m == 0:
            raise RuntimeError("fetch_unlabelled() needs a fitted relevance model: call update() first or pass queries "
                               "(the reference fails with an AttributeError at gp.py:222)")
        unseen = self._unseen_list()
        k = min(int(k), len(unseen))
        if k <= 0:
            return []
        why = self._unsupported(k, len(unseen))
        if why is not None:
            raise NotImplementedError("ital_amd device scorer: %s is not implemented" % why)
        if (self.top_candidates is None and not self._needs_generic() and self.round_call and self.select_in_scorer
                and self._round_possible(k, unseen)):
            # the retrieval loop's own case: the candidate list is get_unseen() itself.  Nothing of its size is touched on
            # the host -- the list lives on the device, the host keeps (base, removed ids) (retrieval_base.UnseenList)
            self._ce_subset = None
            self._last_batch = None
            return self._select_round(k, unseen)
        candidates = self._candidate_list(unseen.array())
        if len(candidates) < k:
            # k was clamped to the number of unseen samples BEFORE the top_candidates restriction (ital.py:99-117): the
            # reference picks until the list is empty and np.argmax([]) then raises exactly this (ital.py:130) -- with
            # the random streams advanced by the steps it did run, so those are run here too
            if len(candidates) > 0:
                self._select(len(candidates), candidates)
            raise ValueError("attempt to get argmax of an empty sequence")
        return self._select(k, candidates)

    def _shard(self, candidates, b=None):
        """Candidate shard of this rank as device arrays: local rows, alive flags, explicit list positions (or None when
        the local positions are one contiguous run of the list), and the list position of the first one.  `b`: batch
        buffers whose cached alive-flag array is reused (the fetch prologue sits between two rounds with the GPU idle)."""
        gp = self.gp
        dev = gp.device
        cand = np.asarray(candidates, dtype=np.int64)
        if gp.world == 1:
            loc_rows, pos_offset, gpos = cand, 0, None          # every candidate is local, in list order
        else:
            loc_rows, pos_offset, gpos = sharding.shard_candidates(cand, gp.row0, gp.row1, self._ascending(candidates))
        n_loc = len(loc_rows)
        cand_d = torch.from_numpy((loc_rows - gp.row0).astype(np.int32)).to(dev) if n_loc else \
            torch.zeros(1, dtype=torch.int32, device=dev)
        gpos_d = torch.from_numpy(gpos).to(dev) if gpos is not None else None
        if b is not None:
            if b.get("alive") is None or b["alive"].numel() < max(n_loc, 1):
                b["alive"] = torch.empty(max(n_loc, 1), dtype=torch.uint8, device=dev)
                b["mi"] = torch.empty(max(n_loc, 1), dtype=torch.float64, device=dev)
            alive = b["alive"]
            alive.fill_(1)
        else:
            alive = torch.ones(max(n_loc, 1), dtype=torch.uint8, device=dev)
        return cand, n_loc, pos_offset, cand_d, gpos_d, alive

    def _ascending(self, candidates):
        """The candidate list is the get_unseen() array itself (ascending by construction): lets the sharding arithmetic
        use binary searches instead of passes over a list of up to millions of entries."""
        u = self.__dict__.get("_unseen")
        return u is not None and candidates is u[0]._flat

    def _qmc_workspace(self, b, t, n_loc):
        """Workspace of the lattice scorer (prepared calls of a slab of candidates), grown on demand up to `qmc_work_bytes`."""
        want = int(_lib.lib().ital_round_workspace(t, max(n_loc, 1), max(self.qmc_work_bytes // 8, 1 << 16)))
        w = b.get("qmc_work")
        if w is None or w.numel() < want:
            b["qmc_work"] = w = torch.empty(want, dtype=torch.float64, device=self.gp.device)
        return w

    def _sel_parts_doubles(self, k, n_loc):
        """Doubles of the block partials of the selection inside the scoring launches of a round of k steps: three per
        scoring block -- n/256 blocks at t = 1, n/32 at t = 2, n/256 plus one per slab of the lattice workspace from t = 3 on
        (with a small workspace cap, `qmc_work_bytes`, a step of t = 7, 8 runs in slabs of a few candidates each)."""
        return int(_lib.lib().ital_sel_parts_len(k, n_loc, max(self.qmc_work_bytes // 8, 1 << 16)))

    def _select(self, k, candidates):
        """Greedy construction of a batch of k out of `candidates` (k <= len(candidates))."""
        gp = self.gp
        self._last_batch = None        # published only after the round's final successful download (update() pairs its
        if self._needs_generic():      # sample ids with rows of the batch buffers)
            self._dev_list = None
            return self._fetch_generic(k, candidates)
        if self.round_call and self.select_in_scorer and self._round_possible(k, candidates):
            return self._select_round(k, candidates)
        self._dev_list = None
        lib = _lib.lib()
        dev = gp.device
        with torch.cuda.device(dev):
            b = self._buffers(k)
            st = _stream()
            # ---- candidate shard of this rank (list positions keep their global numbering)
            cand, n_loc, pos_offset, cand_d, gpos_d, alive = self._shard(candidates, b)
            mi = b["mi"]                                   # every live position is written by the scorer
            self.last_scores = []
            stream = mvn_stream.GLOBAL
            saved_stream = (stream.state, stream.draws)
            n_alive = len(candidates)
            b["ret"][b["kmax"]:].zero_()      # the resolve steps OR every rank's status word into this slot
            for t in range(1, k + 1):
                desc = ItalScoreDesc()
                desc.t = t
                desc.n_cand = n_loc
                desc.cand, desc.alive, desc.mu, desc.s2 = _ptr(cand_d), _ptr(alive), _ptr(gp.mu), _ptr(gp.s2)
                desc.C, desc.ldc = _ptr(b["C"]), gp.ldv
                desc.row_offset, desc.pos_offset, desc.gpos = gp.row0, pos_offset, _ptr(gpos_d)
                desc.batch = b["batch"]
                desc.noise, desc.eps = float(self.noise), float(self.eps)
                desc.label_mode = _LABEL_MODES[self.label_estimation]
                desc.mi = _ptr(mi)
                desc.status = _ptr(gp.status)
                if t >= 3:
                    if t not in b["jump"]:
                        b["jump"][t] = torch.from_numpy(mvn_stream.jump_table(t, ITAL_JUMP_BITS)).to(dev)
                        b["jumppat"][t] = torch.from_numpy(mvn_stream.jump_pattern_table(t)).to(dev)
                        b["vk"][t] = torch.from_numpy(mvn_stream.korobov_vk(t)).to(dev)
                    desc.jump, desc.jumppat, desc.vk = _ptr(b["jump"][t]), _ptr(b["jumppat"][t]), _ptr(b["vk"][t])
                    for j in range(6):
                        desc.seed[j] = stream.state[j]
                    work = self._qmc_workspace(b, t, n_loc)
                    desc.work, desc.work_doubles = _ptr(work), work.numel()
                    if self.profile is not None and n_loc > 0:
                        # the lattice-sum kernel alone, bracketed by events the library records on the launch stream (an
                        # event has to be recorded once before its handle exists: the pool's events are, see bench.py).
                        # Every record is a barrier packet in the queue (~4 us of idle GPU in a 3 ms round), so the step
                        # as a whole is only bracketed where no kernel-level pair exists
                        k0, k1 = self._event(), self._event()
                        desc.ev_start, desc.ev_stop = k0.cuda_event, k1.cuda_event
                        # several slabs: the pair spans first .. last lattice sum incl. the launches between them
                        slabs = -(-n_loc // max(work.numel() // int(lib.ital_score_workspace(t, 1)), 1))
                        self.profile.append(("qmc_main" if slabs == 1 else "qmc_slabs%d" % slabs, t,
                                             n_alive if not gp.collective else n_loc, k0, k1))
                tail = self.select_in_scorer and 0 < n_loc <= _FUSED_SELECT_MAX
                fused = not gp.collective and 0 < n_loc <= (_FUSED_SELECT_MAX if tail else _FUSED_LAUNCH_MAX)
                if tail:
                    # the scoring launch ends with the selection itself (the block that finishes last selects): one rank --
                    # arg-max, record and batch bookkeeping, no selection launch at all; several ranks -- arg-max and record
                    # (what ital_select_local does in a single-workgroup launch of its own), exchange and resolve follow
                    parts = b.get("sel_parts")
                    if parts is None or parts.numel() < self._sel_parts_doubles(k, n_loc):
                        b["sel_parts"] = parts = torch.empty(self._sel_parts_doubles(k, n_loc), dtype=torch.float64, device=dev)
                        b["sel_counter"] = torch.zeros(1, dtype=torch.int32, device=dev)
                    desc.sel_X, desc.sel_xnorm, desc.sel_ldx = _ptr(gp.Xd), _ptr(gp.xnorm), gp.ldx
                    desc.sel_V, desc.sel_ldv, desc.sel_m, desc.sel_ldw, desc.sel_rank = _ptr(gp.V), gp.ldv, gp.m, gp.cap, gp.rank
                    desc.sel_record, desc.sel_ret = _ptr(b["rec"]), (_ptr(b["ret"]) if fused else None)
                    desc.sel_parts, desc.sel_parts_len, desc.sel_counter = _ptr(parts), parts.numel(), _ptr(b["sel_counter"])
                ev0 = self._mark() if t < 3 else None
                check(lib.ital_score_step(ctypes.byref(desc), st))
                if t < 3:
                    self._mark("score", t, n_alive, ev0)
                if self.keep_scores:
                    self.last_scores.append(mi.clone())
                if tail and fused:
                    pass
                elif fused:
                    check(lib.ital_select_fused(_ptr(mi), _ptr(cand_d), _ptr(alive), n_loc, pos_offset, _ptr(gpos_d), gp.row0,
                                                gp.rank, 0, _ptr(gp.mu), _ptr(gp.s2), _ptr(gp.Xd), _ptr(gp.xnorm), gp.ldx,
                                                _ptr(gp.V), gp.ldv, gp.m, gp.cap, _ptr(b["C"]), gp.ldv, t - 1, t - 1, b["batch"],
                                                _ptr(gp.status), _ptr(b["rec"]), _ptr(b["ret"]), st))
                else:
                    if not tail:
                        check(lib.ital_select_local(_ptr(mi), _ptr(cand_d), _ptr(alive), n_loc, pos_offset, _ptr(gpos_d), gp.row0,
                                                    gp.rank, 0, _ptr(gp.mu), _ptr(gp.s2), _ptr(gp.Xd), _ptr(gp.xnorm), gp.ldx,
                                                    _ptr(gp.V), gp.ldv, gp.m, gp.cap, _ptr(b["C"]), gp.ldv, t - 1, b["kmax"],
                                                    _ptr(gp.status), _ptr(b["work"]), _ptr(b["rec"]), st))
                    if gp.collective:
                        ev0 = self._mark()
                        recs = sharding.gather_records(b["rec"], b["rec_all"], gp.group)
                        self._mark("exchange", t, gp.world, ev0)
                    else:
                        recs = b["rec"]
                    check(lib.ital_select_resolve(_ptr(recs), gp.world, b["rec_len"], gp.rank, 0, t - 1, b["batch"],
                                                  _ptr(alive), _ptr(b["ret"]), st))
                if t < k:
                    slot = t - 1
                    ev0 = self._mark() if t == 1 else None           # one sample of the streaming kernel per round
                    check(lib.ital_cross_cov_cols(_ptr(gp.Xd), _ptr(gp.xnorm), gp.n, gp.ldx, _ptr(b["XB"][slot]),
                                                  _ptr(b["XBn"][slot:]), 1, _ptr(b["VB"][slot]), gp.cap, _ptr(gp.V),
                                                  gp.ldv, gp.m, float(self.var), float(self.length_scale),
                                                  _ptr(b["C"][slot]), gp.ldv, st))
                    if t == 1:
                        self._mark("cross_cov", t, gp.m, ev0)
                # the reference's serial loop has now consumed this many uniforms of mvndst's stream
                stream.advance(n_alive * (2 << t) * mvn_stream.draws_per_call(t))
                n_alive -= 1
            host = self._download(b["ret"], "the picks of the round", self._step_estimate_s(k, n_loc)).tolist()     # the only synchronisation of the round: the picks and the status word
            ret, status = host[:k], host[b["kmax"]]   # status: OR over the greedy steps and over all ranks (same everywhere)
            if status & 6:
                # linearly dependent variables inside a batch (duplicate samples), or a simulated update that does not pin the
                # labels (large noise): the fast scorer carries neither MVNDFN's limit-intersection logic nor the updated
                # integrals; redo the round with the general scorer from the same stream position -- on every rank alike
                gp.status.bitwise_and_(~6)
                stream.state, stream.draws = saved_stream
                return self._fetch_generic(k, candidates)
        if status:
            gp.check_status(status)
        self._last_batch = (b, list(ret))
        return [int(i) for i in ret]

    def _round_possible(self, k, candidates):
        """Can this round run as one ital_fetch_round call?  One rank: any list of up to ITAL_ROUND_MAX_CAND candidates.
        Several ranks: the ascending get_unseen() list (an UnseenList: every rank's share is one run of it), at least one
        and at most that many candidates on every rank, and a transport for the per-step record exchange.  Decided from
        the list and the process group alone: the same on every rank."""
        gp = self.gp
        n = len(candidates)
        if n < k:
            return False
        if not gp.collective:
            return n <= _FUSED_SELECT_MAX
        if not isinstance(candidates, UnseenList) or self._round_transport() is None:
            return False
        bounds = [sharding.row_range(gp.n_total, gp.world, r)[0] for r in range(gp.world)] + [gp.n_total]
        sizes = np.diff([candidates.count_below(x) for x in bounds])
        return bool(sizes.min() >= 1 and sizes.max() <= _FUSED_SELECT_MAX)

    @staticmethod
    def _step_estimate_s(k, n_loc):
        """Upper estimate of the compute time of ONE greedy step of the full enumeration on this rank (the last step of a batch
        of k dominates): its (Phi, Phi^-1) pairs at 1e11 pairs/s, less than half the measured rate of the lattice sums -- what
        the exchange deadline must not mistake for a stalled collective (sharding.await_download)."""
        if k < 3:
            return 0.0
        prime = (31, 47, 73, 113, 173, 263, 397, 593, 907, 1361)[min(k - 1, 10) - 1]
        return float(n_loc) * (2 ** k) * 16 * prime * (k - 1) / 1e11

    def _download(self, tensor, what, step_estimate_s=0.0):
        """Device -> host copy of a round's result.  On several ranks it is the point where this rank waits for the round's
        collectives: bounded (no resolved greedy step for ITAL_EXCHANGE_TIMEOUT_S, or for three times the estimated compute
        time of one step if that is longer) and with the raw communicator's asynchronous errors polled
        (sharding.await_download) -- a rank that dies mid-round must not leave the others waiting for ever."""
        gp = self.gp
        if not gp.collective:
            return tensor.cpu()
        kind = getattr(self, "_transport", None)
        kind = kind[1] if kind else None
        comm = kind[1] if kind and kind[0] == "nccl" else None
        if comm is None:
            e = sharding._RAW_COMMS.get((id(gp.group), str(gp.device)))      # the step path uses the raw communicator as well
            comm = e[1] if e else None
        name = {"nccl": "raw_nccl", "host": "host"}[kind[0]] if kind else ("raw_nccl (per step)" if comm else "torch_dist")
        if getattr(self, "_pinned", None) is None:
            self._pinned = {}
        return sharding.await_download(tensor, what, gp.group, gp.device, comm, gp.rank, gp.world, name, pinned=self._pinned,
                                       step_estimate_s=step_estimate_s)

    def _round_transport(self):
        """How ital_fetch_round exchanges the ranks' records: ("nccl", ncclComm_t of the process group -- the communicator
        torch.distributed itself uses, ProcessGroupNCCL._comm_ptr) over RCCL, ("host", None) through torch.distributed from
        a callback (backends that move host memory: the gloo rehearsals and tests), or None (the round is enqueued step by
        step with torch.distributed's all-gather between the launches)."""
        gp = self.gp
        cached = getattr(self, "_transport", None)
        if cached is not None and cached[0] is gp.group:
            return cached[1]
        kind = None
        import torch.distributed as dist
        if gp.group is not None and dist.is_initialized():
            if dist.get_backend(gp.group) == "nccl":
                comm = sharding.raw_comm(gp.group, gp.device)      # None: the step path (torch.distributed between launches)
                kind = ("nccl", comm) if comm else None
            else:
                kind = ("host", None)
        if os.environ.get("ITAL_ROUND_TRANSPORT") == "none":
            kind = None
        elif os.environ.get("ITAL_ROUND_TRANSPORT") == "host" and kind is not None:
            kind = ("host", None)
        if kind is not None:
            self._transport = (gp.group, kind)
        return kind

    def _host_exchange(self, b):
        """The record exchange as a callback of ital_fetch_round (transport "host"): torch.distributed's all-gather of this
        rank's record buffer, issued from inside the C call at the place the RCCL transport issues ncclAllGather."""
        cb = b.get("exchange_cb")
        if cb is None:
            group = self.gp.group

            def exchange(ctx, record, records_all, rec_len, stream):
                try:
                    sharding.gather_records(b["rec"], b["rec_all"], group)
                    return 0
                except Exception as e:      # noqa: BLE001 -- must not unwind through the C frames
                    b["exchange_exc"] = e   # re-raised by the caller of ital_fetch_round (an ExchangeError: deadline / lost peer)
                    return -5
            b["exchange_cb"] = cb = _lib.EXCHANGE_FN(exchange)
        return cb

    def _round_signature(self, b, k):
        gp = self.gp
        w = b.get("qmc_work")
        return (id(b), k, gp.cap, gp.ldv, gp.V.data_ptr(), gp.mu.data_ptr(), 0 if w is None else w.data_ptr(),
                float(self.noise), float(self.eps),
                float(self.var), float(self.length_scale), self.label_estimation, self.qmc_work_bytes,
                self.profile is not None, repr(self.profile_steps))

    def _round_prepare(self, slot, b, k, n, m, begin, cur, state_before, n_prev=0, n_loc=None, pos_offset=0):
        """Fills round descriptor `slot` (one of two) for a round of k steps over n candidates with m labelled samples, the
        candidate list in device buffer `cur` (begin = 2: compacted out of the other buffer, which holds n_prev entries).
        Several ranks: n_loc of the n candidates are this rank's, the first of them at list position pos_offset.
        Nothing here depends on the picks of the round before: the descriptor of the NEXT round is prepared while the GPU
        works on the current one, off the critical path of the retrieval loop."""
        lib = _lib.lib()
        gp = self.gp
        dev = gp.device
        lists = b["cand_lists"]
        r = b["round_descs"][slot]
        d = r.step
        r.k, r.n_rows, r.var, r.length_scale = k, gp.n, float(self.var), float(self.length_scale)
        r.begin, r.cand_prev, r.n_prev = begin, (_ptr(lists[cur ^ 1]) if begin == 2 else None), (n_prev if begin == 2 else 0)
        n_loc = n if n_loc is None else n_loc
        d.n_cand = n_loc
        d.cand, d.alive, d.mu, d.s2 = _ptr(lists[cur]), _ptr(b["alive"]), _ptr(gp.mu), _ptr(gp.s2)
        d.C, d.ldc, d.row_offset, d.pos_offset, d.gpos = _ptr(b["C"]), gp.ldv, gp.row0, pos_offset, None
        r.world, r.records_all, r.nccl_comm, r.exchange = 0, None, None, _lib.EXCHANGE_FN(0)
        if gp.collective:
            kind, comm = self._round_transport()
            r.world, r.records_all = gp.world, _ptr(b["rec_all"])
            if kind == "nccl":
                r.nccl_comm = comm
            else:
                r.exchange = self._host_exchange(b)
        d.batch = b["batch"]
        d.noise, d.eps, d.label_mode = float(self.noise), float(self.eps), _LABEL_MODES[self.label_estimation]
        d.mi, d.status = _ptr(b["mi"]), _ptr(gp.status)
        d.sel_X, d.sel_xnorm, d.sel_ldx = _ptr(gp.Xd), _ptr(gp.xnorm), gp.ldx
        d.sel_V, d.sel_ldv, d.sel_m, d.sel_ldw, d.sel_rank = _ptr(gp.V), gp.ldv, m, gp.cap, gp.rank
        d.sel_record, d.sel_ret = _ptr(b["rec"]), _ptr(b["ret"])
        parts = b["sel_parts"]
        d.sel_parts, d.sel_parts_len, d.sel_counter = _ptr(parts), parts.numel(), _ptr(b["sel_counter"])
        r.mi_keep = None
        events = []
        if k >= 3:
            work = self._qmc_workspace(b, k, n_loc)
            d.work, d.work_doubles = _ptr(work), work.numel()
        for t in range(1, k + 1):
            r.ev_start[t] = r.ev_stop[t] = None
            if t >= 3:
                if t not in b["jump"]:
                    b["jump"][t] = torch.from_numpy(mvn_stream.jump_table(t, ITAL_JUMP_BITS)).to(dev)
                    b["jumppat"][t] = torch.from_numpy(mvn_stream.jump_pattern_table(t)).to(dev)
                    b["vk"][t] = torch.from_numpy(mvn_stream.korobov_vk(t)).to(dev)
                r.jump[t], r.jumppat[t], r.vk[t] = _ptr(b["jump"][t]), _ptr(b["jumppat"][t]), _ptr(b["vk"][t])
                if self.profile is not None and (self.profile_steps is None or t in self.profile_steps):
                    k0, k1 = self._event(), self._event()
                    r.ev_start[t], r.ev_stop[t] = k0.cuda_event, k1.cuda_event
                    slabs = -(-n_loc // max(work.numel() // int(lib.ital_score_workspace(t, 1)), 1))
                    # (candidates the bracketed launches score: the whole list on one rank; this rank's share otherwise --
                    # which rank the earlier picks of the round come from is not known when the descriptor is built)
                    events.append(("qmc_main" if slabs == 1 else "qmc_slabs%d" % slabs, t,
                                   n - (t - 1) if not gp.collective else n_loc, k0, k1))
        # the reference's serial loop consumes n_alive * 2 * 2^t calls of mvndst's stream at step t: states before every step
        st6 = (ctypes.c_int * 6)(*state_before)
        check(lib.ital_mvn_round_seeds(st6, n, k, ctypes.byref(r.seeds)))
        draws = sum((n - (t - 1)) * (2 << t) * mvn_stream.draws_per_call(t) for t in range(3, k + 1))
        return dict(slot=slot, k=k, n=n, n_loc=n_loc, m=m, begin=begin, cur=cur, state_before=tuple(state_before),
                    state_after=tuple(int(v) for v in st6), draws=draws, events=events, sig=self._round_signature(b, k))

    def _select_round(self, k, candidates):
        """_select as ONE call below the C ABI (ital_fetch_round): candidate-list upkeep, k scoring steps that end with
        their selection (several ranks: with the rank's record, then the exchange issued from C and the resolve launch),
        k - 1 covariance columns -- enqueued from C (a Python host needs 10 - 17 us per launch, the first greedy steps are
        shorter than that).  Several ranks work on their own share of the list (list positions lo .. hi).  The candidate list stays on the device between rounds: when
        the list is the previous one minus the previous batch (the retrieval loop: fetch, label the batch, fetch), it is
        compacted there by its alive flags instead of being rebuilt and uploaded; and the descriptor of such a next round
        is filled in while the GPU still works on the current one."""
        t_enter = time.perf_counter()
        lib = _lib.lib()
        gp = self.gp
        dev = gp.device
        n = len(candidates)
        listed = isinstance(candidates, UnseenList)      # (several ranks: always, see _round_possible)
        if gp.collective:
            # this rank's share of the ascending list: one run of it, list positions lo .. hi
            lo, hi = candidates.count_below(gp.row0), candidates.count_below(gp.row1)
        else:
            lo, hi = 0, n
        n_loc = hi - lo
        with torch.cuda.device(dev):
            b = self._buffers(k)
            st = _stream()
            # ---- candidate list: two device buffers (the compaction reads one, writes the other)
            lists = b.get("cand_lists")
            if lists is None or lists[0].numel() < n_loc or b.get("alive") is None or b["alive"].numel() < n_loc \
                    or b["sel_parts"].numel() < self._sel_parts_doubles(k, n_loc):
                b["cand_lists"] = lists = [torch.empty(max(n_loc, 1), dtype=torch.int32, device=dev) for _ in range(2)]
                b["cand_cur"] = 0
                b["alive"] = torch.empty(n_loc, dtype=torch.uint8, device=dev)
                b["mi"] = torch.empty(n_loc, dtype=torch.float64, device=dev)
                b["sel_parts"] = torch.empty(self._sel_parts_doubles(k, n_loc), dtype=torch.float64, device=dev)
                b["sel_counter"] = torch.zeros(1, dtype=torch.int32, device=dev)
                b["round_descs"] = [_lib.ItalRoundDesc(), _lib.ItalRoundDesc()]
                b["round_next"] = None
                self._dev_list = None
            dl = self._dev_list
            stream = mvn_stream.GLOBAL
            # the device holds the list as of version v with the picks of that round flagged dead: it follows the host's
            # when the host's list is that version minus exactly those picks
            follows = (listed and dl is not None and dl["b"] is b and dl["unseen"] is candidates
                       and dl["version"] == candidates.version - 1 and tuple(sorted(dl["picks"])) == candidates.last_removed)
            self._dev_list = None                              # re-published after the round's successful download
            p = b["round_next"]
            if (p is not None and follows and not self.keep_scores and p["k"] == k and p["n"] == n and p["m"] == gp.m
                    and p["state_before"] == tuple(stream.state) and p["sig"] == self._round_signature(b, k)):
                b["cand_cur"] = p["cur"]                       # the round the previous one prepared for
                if gp.collective:
                    # prepared before the picks were known: how many of them lay in this rank's rows and before them
                    d = b["round_descs"][p["slot"]].step
                    d.n_cand, d.pos_offset = n_loc, lo
                    p["n_loc"] = n_loc
                    for i, ev in enumerate(p["events"]):
                        p["events"][i] = ev[:2] + (n_loc,) + ev[3:]
            else:
                if p is not None:                              # prepared for a round that did not come: its events go back
                    for ev in p["events"]:
                        self.event_pool += [ev[3], ev[4]]
                n_prev = 0
                if follows:
                    begin, n_prev = 2, dl["n_loc"]             # the device holds the parent list with exactly those picks flagged
                    b["cand_cur"] ^= 1
                else:
                    begin = 1
                    share = candidates.in_rows(gp.row0, gp.row1) if listed else np.asarray(candidates, dtype=np.int64)
                    lists[b["cand_cur"]][:n_loc].copy_(torch.from_numpy((share - gp.row0).astype(np.int32)))
                p = self._round_prepare(0, b, k, n, gp.m, begin, b["cand_cur"], stream.state, n_prev, n_loc, lo)
            b["round_next"] = None
            self.last_round = (p["begin"], p["slot"])          # diagnostics / tests: how the candidate list reached the device
            r = b["round_descs"][p["slot"]]
            keep = None
            if self.keep_scores:
                keep = torch.zeros((k, n_loc), dtype=torch.float64, device=dev)
                r.mi_keep = _ptr(keep)
            saved_stream = (stream.state, stream.draws)
            hc = self.host_clock
            if hc is not None:
                # host time on the critical path of the retrieval loop: from the download of the previous round's picks (the
                # caller's feedback, update(), this prologue) to the call that enqueues the next round
                t_call = time.perf_counter()
                if hc.get("t_download") is not None:
                    hc["gap_s"] += t_call - hc["t_download"]
                    hc["gaps"] += 1
                    hc["prologue_s"] = hc.get("prologue_s", 0.0) + (t_call - t_enter)
            rc = lib.ital_fetch_round(ctypes.byref(r), st)
            if rc and b.get("exchange_exc") is not None:      # the host transport's callback failed: its own error, not the C one
                exc, b["exchange_exc"] = b["exchange_exc"], None
                raise exc
            check(rc)
            stream.state, stream.draws = p["state_after"], stream.draws + p["draws"]
            if self.profile is not None:
                self.profile += p["events"]
            # ---- while the GPU works: the descriptor of the round that follows in the retrieval loop (this batch labelled,
            # then the next fetch of k): nothing in it depends on which samples this round picks
            if n - k >= k and gp.m + k <= gp.cap and not self.keep_scores:
                # (several ranks: this rank's share of that list is known only with the picks -- patched in when the round comes)
                b["round_next"] = self._round_prepare(p["slot"] ^ 1, b, k, n - k, gp.m + k, 2, b["cand_cur"] ^ 1, stream.state,
                                                      n_loc, n_loc if gp.collective else n - k, lo)
            if hc is not None:
                hc["enqueue_s"] += time.perf_counter() - t_call     # the call itself + the next round's descriptor (GPU busy)
            host = self._download(b["ret"], "the picks of the round", self._step_estimate_s(k, n_loc)).tolist()     # the only synchronisation of the round: the picks and the status word
            if hc is not None:
                hc["t_download"] = time.perf_counter()
            ret, status = host[:k], host[b["kmax"]]
            self.last_scores = [keep[t, :n_loc] for t in range(k)] if keep is not None else []
            if status & 8:
                raise RuntimeError("ital_amd: the candidate list kept on the device lost track of the host's (internal error)")
            if status & 6:
                # see _select: duplicates inside the batch / a simulated update that does not pin the labels
                gp.status.bitwise_and_(~6)
                stream.state, stream.draws = saved_stream
                return self._fetch_generic(k, candidates.array() if listed else candidates)
        if status:
            gp.check_status(status)
        self._last_batch = (b, list(ret))
        if listed:
            self._dev_list = dict(b=b, unseen=candidates, version=candidates.version, picks=[int(i) for i in ret], n_loc=n_loc)
        return [int(i) for i in ret]

    # ------------------------------------------------------------------ general scorer (noisy users, estimation subset)
    def _fetch_generic(self, k, candidates):
        """Greedy batch construction through ital_score_generic: any user model (reference ital.py:300-342), with or
        without a change-estimation subset (ital.py:227-275, 541-582).

        Without a subset the base set of the scorer IS the batch so far: it lives in the replicated device batch state that
        ital_select_resolve / ital_select_fused maintain, and a round is enqueued without any host round trip (one download
        of the picks at the end) unless an option needs the host between the steps (Monte-Carlo sampling on numpy's
        generator, the counting pass of clip_cov).  With a subset the base set also holds the subset members and grows only
        when a pick lies outside it: that bookkeeping stays on the host (one synchronisation per greedy step)."""
        lib = _lib.lib()
        gp = self.gp
        dev = gp.device
        self._last_batch = None
        subset_mode = self._ce_subset is not None
        fb_mode = self._fb_mode()
        E = list(self._ce_subset) if subset_mode else []
        kmax_e = len(E) + k
        GN = ITAL_GENERIC_MAX_DIM
        stream = mvn_stream.GLOBAL
        h = ITAL_REC_HEADER
        with torch.cuda.device(dev):
            st = _stream()
            b = self._buffers(max(kmax_e, 4))
            kmax = b["kmax"]
            cand, n_loc, pos_offset, cand_d, gpos_d, alive = self._shard(candidates)
            # every rank's candidates one contiguous run of the list (the ascending get_unseen() order; not after the
            # argpartition order of top_candidates on several ranks)?  Decided from the list alone: the same on every rank
            runs = sharding.contiguous_runs(cand, gp.n_total, gp.world, self._ascending(candidates))
            pos_of = {int(c): i for i, c in enumerate(cand.tolist())}
            mi = torch.zeros(max(n_loc, 1), dtype=torch.float64, device=dev)
            if "jump1" not in b:
                b["jump1"] = torch.from_numpy(mvn_stream.jump1_table(ITAL_JUMP_BITS)).to(dev)
                b["vk_all"] = torch.from_numpy(mvn_stream.vk_table(GN)).to(dev)
                b["iota"] = torch.arange(kmax, dtype=torch.int32, device=dev)
                b["zero64"] = torch.zeros(1, dtype=torch.int64, device=dev)
            jump1, vk = b["jump1"], b["vk_all"]
            C = b["C"]
            e_mu = np.zeros(kmax_e)
            e_sig = np.zeros((kmax_e, kmax_e))
            keep = []                     # device temporaries of the enqueued work (released after the round's synchronisation)
            b["ret"][kmax:].zero_()
            if E:
                # covariance columns of the subset members with every row, and among themselves
                rows = gp._gather_rows(E)
                norms = torch.empty(len(E), dtype=torch.float64, device=dev)
                check(lib.ital_row_norms(_ptr(rows), len(E), gp.ldx, _ptr(norms), st))
                vcols = gp.gather_columns(gp.V[: max(gp.m, 1)], E)          # [m, |E|]
                for c0 in range(0, len(E), 16):
                    c = min(16, len(E) - c0)
                    Wt = torch.zeros((c, gp.cap), dtype=torch.float64, device=dev)
                    Wt[:, : gp.m] = vcols[: gp.m, c0:c0 + c].t()
                    check(lib.ital_cross_cov_cols(_ptr(gp.Xd), _ptr(gp.xnorm), gp.n, gp.ldx, _ptr(rows[c0:c0 + c]),
                                                  _ptr(norms[c0:c0 + c]), c, _ptr(Wt), gp.cap, _ptr(gp.V), gp.ldv, gp.m,
                                                  float(self.var), float(self.length_scale), _ptr(C[c0:c0 + c]), gp.ldv,
                                                  st))
                    keep.append(Wt)
                keep += [rows, norms, vcols]
                e_sig[: len(E), : len(E)] = gp.gather_columns(C[: len(E)], E).cpu().numpy()
                e_mu[: len(E)] = self.rel_mean[np.asarray(E)]
            picks, pick_pos = [], []
            self._mc_cache = {}           # host copies of this rank's variances / covariance columns (pattern sampling)
            self.last_scores = []
            self.last_patterns = []       # keep_scores: the sampled sign patterns of every Monte-Carlo step (diagnostics/tests)
            n_alive = len(candidates)
            z_next = None
            for t in range(1, k + 1):
                nE = len(E) if subset_mode else t - 1
                nr = t
                rel_mc, npat, fb_mc, nfb = self._mc_plan(nr, fb_mode)
                clip_count = self._clip_active() and nE + 1 > 5
                dpc = mvn_stream.draws_per_call
                if subset_mode:
                    draws_out = npat * (dpc(nr) + (1 + nfb) * dpc(nE + 1))
                    draws_in = npat * (dpc(nr) + (1 + nfb) * dpc(nE))
                    in_pos = sorted(pos_of[e] for e in E if e in pos_of and e not in picks)
                else:
                    draws_out = npat * (1 + nfb) * dpc(nr)
                    draws_in = 0
                    in_pos = []
                mc = None
                if rel_mc or fb_mc:
                    t_host0 = time.perf_counter()
                    if not subset_mode and t > 1:
                        # batch state kept by the device: the members' means / covariances for the pattern sampler
                        picks = [int(i) for i in b["ret"][: t - 1].cpu().tolist()]
                        t_host1 = time.perf_counter()          # (the wait for the step before: GPU time, not host time)
                        pick_pos = list(range(t - 1))
                        e_mu[: t - 1] = b["bmu"][: t - 1].cpu().numpy()
                        e_sig[: t - 1, : t - 1] = b["sig"].view(kmax, kmax)[: t - 1, : t - 1].cpu().numpy()
                    else:
                        t_host1 = t_host0
                    # pattern sampling alone on a large shard: the step is scored in ranges of candidates, the SVDs of the
                    # next range on the host under the lattice sums of the current one
                    n_chunks = (_MC_CHUNKS + (2 if n_loc >= 32 * _MC_CHUNK_MIN else 0)) if (rel_mc and not fb_mc and not subset_mode and not clip_count and runs
                                              and gpos_d is None
                                              and (nr >= _MC_CHUNK_FROM or (nr >= 7 and n_loc >= 32 * _MC_CHUNK_MIN))
                                              and n_loc >= _MC_CHUNK_MIN * _MC_CHUNKS) else 0
                    mc = self._mc_samples(nr, npat, rel_mc, fb_mc, nfb, fb_mode, cand, picks, pos_of,
                                          E if subset_mode else picks, pick_pos, e_mu, e_sig, C, subset_mode, z_next,
                                          (pos_offset, pos_offset + n_loc) if runs else None, n_chunks)
                    if os.environ.get("ITAL_MC_TIMING"):
                        print("t=%d: waited %.1f ms for the step before; batch state + sampler set-up%s %.1f ms" % (
                            t, (t_host1 - t_host0) * 1e3, "" if n_chunks else " + ALL decompositions",
                            (time.perf_counter() - t_host1) * 1e3), flush=True)
                z_next = None
                desc = ItalGscoreDesc()
                desc.n_cand = n_loc
                desc.cand, desc.alive, desc.mu, desc.s2 = _ptr(cand_d), _ptr(alive), _ptr(gp.mu), _ptr(gp.s2)
                desc.C, desc.ldc = _ptr(C), gp.ldv
                desc.row_offset, desc.pos_offset, desc.gpos = gp.row0, pos_offset, _ptr(gpos_d)
                if subset_mode:
                    dead_pos = [pos_of[q] for q in picks]
                    t_eidx = torch.as_tensor(E if E else [0], dtype=torch.int64, device=dev)
                    t_esort = torch.as_tensor(np.argsort(np.asarray(E, dtype=np.int64), kind="stable") if E else [0],
                                              dtype=torch.int32, device=dev)
                    t_emu = torch.from_numpy(np.ascontiguousarray(e_mu)).to(dev)
                    t_esig = torch.from_numpy(np.ascontiguousarray(e_sig)).to(dev)
                    t_ppos = torch.as_tensor(pick_pos if pick_pos else [0], dtype=torch.int32, device=dev)
                    t_in = torch.as_tensor(in_pos if in_pos else [0], dtype=torch.int64, device=dev)
                    t_dead = torch.as_tensor(dead_pos if dead_pos else [0], dtype=torch.int64, device=dev)
                    keep += [t_eidx, t_esort, t_emu, t_esig, t_ppos, t_in, t_dead]
                    desc.nE, desc.E_idx, desc.E_sort, desc.E_mu, desc.E_sig, desc.ldE = nE, _ptr(t_eidx), _ptr(t_esort), \
                        _ptr(t_emu), _ptr(t_esig), kmax_e
                    desc.n_picks, desc.pick_pos = len(picks), _ptr(t_ppos)
                    desc.n_in, desc.in_pos, desc.n_dead, desc.dead_pos = len(in_pos), _ptr(t_in), len(dead_pos), _ptr(t_dead)
                else:
                    # the base set is the batch so far: members, their order by data index, means, covariances and list
                    # positions are the device batch state itself
                    desc.nE, desc.E_idx, desc.E_sort, desc.E_mu, desc.E_sig, desc.ldE = nE, _ptr(b["bidx"]), \
                        _ptr(b["bsort"]), _ptr(b["bmu"]), _ptr(b["sig"]), kmax
                    desc.n_picks, desc.pick_pos = nE, _ptr(b["iota"])
                    desc.n_in, desc.in_pos, desc.n_dead, desc.dead_pos = 0, _ptr(b["zero64"]), nE, _ptr(b["bgpos"])
                desc.subset_mode, desc.fb_mode = int(subset_mode), fb_mode
                desc.label_prob, desc.mistake_prob = float(self.label_prob), float(self.mistake_prob)
                desc.label_mode = _LABEL_MODES[self.label_estimation]
                desc.noise, desc.eps = float(self.noise), float(self.eps)
                desc.clip_cov = float(self.clip_cov) if self._clip_active() else 0.0
                for j in range(6):
                    desc.seed[j] = stream.state[j]
                desc.jump1, desc.vk = _ptr(jump1), _ptr(vk)
                desc.draws_out, desc.draws_in = draws_out, draws_in
                desc.mi, desc.status = _ptr(mi), _ptr(gp.status)
                desc.pair_count = _ptr(self.pair_counter)
                if self.generic_pipeline and not self._clip_active() and nE + 1 <= (13 if subset_mode else 16):
                    # (round 5: with a change-estimation subset too -- the pipeline's wide form, one lattice-sum launch per
                    # dimension that occurs among the step's calls)
                    # workspace of the pipeline of kernels (verdicts, records of the calls to integrate): what one slab of
                    # all candidates takes, capped (the library then walks the candidates in several slabs)
                    desc.mc_rel, desc.mc_fb = (npat if rel_mc else 0), (nfb if fb_mc else 0)     # (read by the size query)
                    want = int(lib.ital_score_generic_workspace(ctypes.byref(desc)))
                    desc.mc_rel, desc.mc_fb = 0, 0
                    want = min(want, max(self.qmc_work_bytes // 8, 1 << 16))
                    w = b.get("qmc_work")
                    if w is None or w.numel() < want:
                        b["qmc_work"] = w = torch.empty(want, dtype=torch.float64, device=dev)
                    desc.work, desc.work_doubles = _ptr(w), w.numel()
                total_draws = None
                rel_ranges = None
                if mc is None and self.keep_scores:
                    self.last_patterns.append(None)          # this step enumerates its patterns
                if mc is not None:
                    rel_arr, fb_arr, draws_pp = mc          # per list position (dead positions hold zeros)
                    if rel_arr is not None and not isinstance(rel_arr, np.ndarray):
                        rel_ranges, rel_arr = rel_arr, None     # generator of (lo, hi, rows): consumed at the launches below
                    if self.keep_scores and rel_ranges is None:
                        self.last_patterns.append(rel_arr)
                    if gpos_d is None:
                        mine = slice(pos_offset, pos_offset + max(n_loc, 1))
                    else:
                        mine = gpos_d.cpu().numpy()
                    if rel_arr is not None:
                        t_rel = torch.from_numpy(np.ascontiguousarray(rel_arr[mine])).to(dev)
                        desc.mc_rel, desc.rel_samples = npat, _ptr(t_rel)
                        keep.append(t_rel)
                    if fb_arr is not None:
                        t_fb = torch.from_numpy(np.ascontiguousarray(fb_arr[mine])).to(dev)
                        desc.mc_fb, desc.fb_samples = nfb, _ptr(t_fb)
                        keep.append(t_fb)
                    off = np.concatenate(([0], np.cumsum(draws_pp)[:-1])).astype(np.int64)
                    t_off = torch.from_numpy(np.ascontiguousarray(off[mine])).to(dev)
                    keep.append(t_off)
                    desc.draw_off = _ptr(t_off)
                    total_draws = int(draws_pp.sum())
                if clip_count:
                    # with clip_cov the number of mvndst calls (one per group of correlated variables) and hence the
                    # stream consumption depends on the data: a counting pass of the same kernel reports it per candidate
                    counts = torch.zeros(max(n_loc, 1), dtype=torch.int64, device=dev)
                    desc.draw_count = _ptr(counts)
                    check(lib.ital_score_generic(ctypes.byref(desc), st))
                    desc.draw_count = None
                    if gp.collective:
                        # uniforms consumed before each of this rank's candidates: prefix over the whole list
                        full = torch.zeros(len(cand), dtype=torch.int64, device=dev)
                        if n_loc:
                            if gpos_d is None:
                                full[pos_offset:pos_offset + n_loc] = counts[:n_loc]
                            else:
                                full[gpos_d] = counts[:n_loc]
                        sharding.all_reduce_sum(full, gp.group)
                        excl = torch.cumsum(full, 0) - full
                        t_off = (excl[pos_offset:pos_offset + max(n_loc, 1)] if gpos_d is None else excl[gpos_d]).contiguous()
                        total_draws = int(full.sum().item())
                    else:
                        t_off = torch.cumsum(counts, 0) - counts
                        total_draws = int(counts.sum().item())
                    keep += [counts, t_off]
                    desc.draw_off = _ptr(t_off)
                ev0 = self._mark()
                if rel_ranges is None:
                    check(lib.ital_score_generic(ctypes.byref(desc), st))
                else:
                    # one call per range of candidates: patterns of range r + 1 are decomposed on the host (LAPACK, thread
                    # pool) while the GPU integrates range r; the uploads go through page-locked memory on a stream of
                    # their own (a pageable copy would wait for the scorer in front of it)
                    if b.get("mc_pin") is None or b["mc_pin"].shape[0] < n_loc or b["mc_pin"].shape[1] < npat:
                        b["mc_pin"] = torch.empty((n_loc, ITAL_GENERIC_MAX_REL), dtype=torch.int32).pin_memory()
                        b["mc_dev"] = torch.empty((n_loc, ITAL_GENERIC_MAX_REL), dtype=torch.int32, device=dev)
                        b["mc_stream"] = torch.cuda.Stream(device=dev)
                    pin, t_rel, side = b["mc_pin"], b["mc_dev"], b["mc_stream"]
                    main = torch.cuda.current_stream(dev)
                    side.wait_stream(main)                   # earlier readers of the device buffer (the step before) are done
                    kept = np.zeros((len(cand), npat), dtype=np.uint32) if self.keep_scores else None
                    base = {f: getattr(desc, f) for f in ("cand", "alive", "mi", "draw_off", "pos_offset")}
                    dbg = os.environ.get("ITAL_MC_TIMING")
                    tq = time.perf_counter()
                    # (try / finally around the WHOLE loop, not only the C call: the generator runs host LAPACK and thread-pool
                    # work between the ranges -- if that raises while a deferred range is still running on the library's internal
                    # streams, the join below is what orders those kernels, which write `mi` and the workspace, before anything
                    # the caller's stream does next with these torch buffers)
                    try:
                        for lo, hi, rows, last_range in rel_ranges:
                            if dbg:
                                t_rows = time.perf_counter() - tq
                                tq = time.perf_counter()
                            a, e = lo - pos_offset, hi - pos_offset
                            flat = pin.view(-1)[a * npat:e * npat]
                            flat.copy_(torch.from_numpy(rows.view(np.int32).reshape(-1)))
                            dflat = t_rel.view(-1)[a * npat:e * npat]
                            with torch.cuda.stream(side):
                                dflat.copy_(flat, non_blocking=True)
                            up = torch.cuda.Event()
                            up.record(side)
                            main.wait_event(up)
                            desc.n_cand = e - a
                            desc.cand, desc.alive = base["cand"] + 4 * a, base["alive"] + a
                            desc.mi, desc.draw_off = base["mi"] + 8 * a, base["draw_off"] + 8 * a
                            desc.pos_offset = base["pos_offset"] + a
                            desc.mc_rel, desc.rel_samples = npat, dflat.data_ptr()
                            # all but the last range leave the library's streams unjoined: the preparation of the next range's
                            # first slab then runs under this range's lattice sums (ital_gscore_desc.defer_join)
                            desc.defer_join = 0 if last_range else 1
                            check(lib.ital_score_generic(ctypes.byref(desc), st))
                            if dbg:
                                print("t=%d range %d..%d: patterns %.1f ms, upload + launch %.1f ms" % (
                                    t, lo, hi, t_rows * 1e3, (time.perf_counter() - tq) * 1e3), flush=True)
                                tq = time.perf_counter()
                            if kept is not None:
                                kept[lo:hi] = rows
                    finally:
                        desc.defer_join = 0
                        rc_join = lib.ital_score_generic_join(st)      # (nothing pending after a last range; cheap)
                    check(rc_join)
                    if kept is not None:
                        self.last_patterns.append(kept)
                self._mark("score_generic", t, n_alive, ev0)
                if self.keep_scores:
                    self.last_scores.append(mi.clone())
                n_in_alive = len(in_pos)
                stream.advance(total_draws if total_draws is not None else
                               (n_alive - n_in_alive) * draws_out + n_in_alive * draws_in)
                n_alive -= 1
                if t < k:
                    # the standard normals of the next step's pattern sampling depend on nothing but their count: drawn
                    # now, while the scorer runs, they are off the critical path (same order on numpy's global generator).
                    # Only this rank's candidates' normals are computed; the generator is walked past the others'
                    # (ital_np_legacy_normals).  The next step's live ranks of the local positions [lo, hi) depend on the
                    # pick this step is about to make: lo - t <= first, last <= hi covers every outcome
                    rel_nx, npat_nx, fb_nx, _ = self._mc_plan(nr + 1, fb_mode)
                    if rel_nx and not fb_nx:
                        g0, g1 = (0, n_alive) if not runs else \
                            (max(pos_offset - t, 0), min(pos_offset + n_loc, n_alive))
                        z_next = (g0, self._walk_normals(n_alive, g0, g1, npat_nx * (nr + 1)).reshape(-1, npat_nx, nr + 1))
                if not subset_mode:
                    slot = t - 1
                    if not gp.collective and n_loc <= _FUSED_LAUNCH_MAX:
                        check(lib.ital_select_fused(_ptr(mi), _ptr(cand_d), _ptr(alive), n_loc, pos_offset, _ptr(gpos_d),
                                                    gp.row0, gp.rank, 0, _ptr(gp.mu), _ptr(gp.s2), _ptr(gp.Xd),
                                                    _ptr(gp.xnorm), gp.ldx, _ptr(gp.V), gp.ldv, gp.m, gp.cap, _ptr(C), gp.ldv,
                                                    slot, slot, b["batch"], _ptr(gp.status), _ptr(b["rec"]), _ptr(b["ret"]),
                                                    st))
                    else:
                        check(lib.ital_select_local(_ptr(mi), _ptr(cand_d), _ptr(alive), n_loc, pos_offset, _ptr(gpos_d),
                                                    gp.row0, gp.rank, 0, _ptr(gp.mu), _ptr(gp.s2), _ptr(gp.Xd),
                                                    _ptr(gp.xnorm), gp.ldx, _ptr(gp.V), gp.ldv, gp.m, gp.cap, _ptr(C), gp.ldv,
                                                    slot, kmax, _ptr(gp.status), _ptr(b["work"]), _ptr(b["rec"]), st))
                        recs = sharding.gather_records(b["rec"], b["rec_all"], gp.group) if gp.collective else b["rec"]
                        check(lib.ital_select_resolve(_ptr(recs), gp.world, b["rec_len"], gp.rank, 0, slot, b["batch"],
                                                      _ptr(alive), _ptr(b["ret"]), st))
                    if t < k:
                        ev0 = self._mark()
                        check(lib.ital_cross_cov_cols(_ptr(gp.Xd), _ptr(gp.xnorm), gp.n, gp.ldx, _ptr(b["XB"][slot]),
                                                      _ptr(b["XBn"][slot:]), 1, _ptr(b["VB"][slot]), gp.cap, _ptr(gp.V),
                                                      gp.ldv, gp.m, float(self.var), float(self.length_scale),
                                                      _ptr(C[slot]), gp.ldv, st))
                        self._mark("cross_cov", t, gp.m, ev0)
                    continue
                # ---- subset mode: the winner is resolved on the host (it may or may not extend the base set)
                check(lib.ital_select_local(_ptr(mi), _ptr(cand_d), _ptr(alive), n_loc, pos_offset, _ptr(gpos_d), gp.row0,
                                            gp.rank, 0, _ptr(gp.mu), _ptr(gp.s2), _ptr(gp.Xd), _ptr(gp.xnorm), gp.ldx,
                                            _ptr(gp.V), gp.ldv, gp.m, gp.cap, _ptr(C), gp.ldv, nE, kmax, _ptr(gp.status),
                                            _ptr(b["work"]), _ptr(b["rec"]), st))
                recs = sharding.gather_records(b["rec"], b["rec_all"], gp.group) if gp.collective else b["rec"].unsqueeze(0)
                recs_h = self._download(recs, "the records of greedy step %d" % t).numpy()          # host synchronisation of this greedy step
                keep.clear()
                w = sharding.winner(recs_h, 0)
                rec = recs_h[w]
                pick = int(rec[2])
                if int(rec[6]) == gp.rank:
                    alive[int(rec[7])] = 0
                picks.append(pick)
                if pick in E:
                    pick_pos.append(E.index(pick))
                else:
                    # new member of the base set: its covariance column, mean and covariances with the members so far
                    e_mu[nE] = rec[3]
                    e_sig[nE, nE] = rec[4]
                    e_sig[nE, :nE] = rec[h + gp.ldx + gp.cap: h + gp.ldx + gp.cap + nE]
                    e_sig[:nE, nE] = e_sig[nE, :nE]
                    if t < k:
                        rec_d = recs[w]
                        xrow = rec_d[h:h + gp.ldx].contiguous()
                        vcol = rec_d[h + gp.ldx:h + gp.ldx + gp.cap].contiguous()
                        xn = rec_d[5:6].contiguous()
                        check(lib.ital_cross_cov_cols(_ptr(gp.Xd), _ptr(gp.xnorm), gp.n, gp.ldx, _ptr(xrow), _ptr(xn), 1,
                                                      _ptr(vcol), gp.cap, _ptr(gp.V), gp.ldv, gp.m, float(self.var),
                                                      float(self.length_scale), _ptr(C[nE]), gp.ldv, st))
                        keep += [xrow, vcol, xn]
                    pick_pos.append(nE)
                    E.append(pick)
            if not subset_mode:
                host = self._download(b["ret"], "the picks of the round").tolist()        # picks and the status word (OR over steps and ranks)
                picks, status = host[:k], host[kmax]
            else:
                status = None                         # only the replicated Cholesky append reports here: same on all ranks
            keep.clear()
        gp.check_status(status)
        if not subset_mode:
            self._last_batch = (b, [int(i) for i in picks])
        return [int(i) for i in picks]

    def _walk_normals(self, n_live, j0, j1, per_cand):
        """numpy's global generator walked over the `per_cand` standard normals of each of `n_live` candidates
        (multivariate_normal.rvs per candidate, reference ital.py:297); returns those of candidates j0 .. j1-1 (flat).
        `mc_walk` accumulates [normals computed, normals skipped, seconds] (diagnostics / tests)."""
        j0, j1 = max(int(j0), 0), min(int(j1), int(n_live))
        j1 = max(j1, j0)
        t0 = time.perf_counter()
        z = _lib.legacy_normals(j0 * per_cand, (j1 - j0) * per_cand, _HOST_THREADS)
        _lib.legacy_normals((n_live - j1) * per_cand, 0)
        self.mc_walk[0] += (j1 - j0) * per_cand
        self.mc_walk[1] += (n_live - (j1 - j0)) * per_cand
        self.mc_walk[2] += time.perf_counter() - t0
        return z

    def _mc_samples(self, nr, npat, rel_mc, fb_mc, nfb, fb_mode, cand, picks, pos_of, E, pick_pos, e_mu, e_sig, C, subset_mode,
                    z_rel=None, local=None, chunks=0):
        """Sign patterns / feedback configurations of the Monte-Carlo switches, drawn from numpy's global RNG in the
        reference's serial order (per live candidate: one multivariate_normal.rvs, ital.py:297; per pattern one
        np.random.choice, ital.py:323-337).  `z_rel`: the standard normals of this step when the caller drew them ahead
        (pattern sampling only); `local`: list positions [lo, hi) whose patterns this rank reads.  Returns per list position: patterns [P, npat] uint32 (or None), feedback
        [P, npat, nfb] uint32 (or None), uniforms of mvndst's stream consumed [P] int64.
        `chunks` > 0 (pattern sampling alone, `local` given): the patterns come back as a generator of (lo, hi, rows) over
        `chunks` consecutive ranges of the local list positions -- the per-candidate decompositions of a range are only done
        when it is asked for, so the caller can score one range on the GPU while the host prepares the next."""
        gp = self.gp
        P = len(cand)
        dead = np.zeros(P, dtype=bool)
        for q in picks:
            dead[pos_of[q]] = True
        live = np.flatnonzero(~dead)
        nE = len(E)
        in_e = np.full(P, -1, dtype=np.int64)
        if subset_mode:
            for e_pos, e in enumerate(E):
                if e in pos_of:
                    in_e[pos_of[e]] = e_pos
        n_full = np.where(in_e >= 0, nE, nE + 1)               # orthant dimension of the full-dimension calls
        dpc = mvn_stream.draws_per_call
        d_full = np.array([dpc(int(v)) for v in range(nE + 2)], dtype=np.int64)[n_full]
        npre_draws = (dpc(nr) + d_full) if subset_mode else d_full
        rel_arr = fb_arr = None
        # ---- mean / covariance of the enumerated variables of the live candidates (ital.py:247-248, :529); with pattern
        # sampling alone only of the ones this rank scores (live[jl0:jl1]): nobody reads the other patterns
        jl0, jl1 = 0, len(live)
        if rel_mc and not fb_mc and local is not None:
            jl0, jl1 = int(np.searchsorted(live, local[0])), int(np.searchsorted(live, local[1]))
        if rel_mc:
            mu_all = np.asarray(self.rel_mean, dtype=np.float64)
            pp = list(pick_pos)
            rows = cand[live[jl0:jl1]]
            cache = self._mc_cache
            if rel_mc and not fb_mc and local is not None:
                # (the same decision on every rank: `local` is set for all of them or for none)
                # every row read below is a row of this rank: variances and covariance columns come from the local shard
                # (a column is downloaded once per fetch, when its member joins the batch) -- no collective, nothing N-sized
                if "s2" not in cache:
                    cache["s2"] = gp.s2[: gp.n].cpu().numpy()
                for b in pp:
                    if ("C", b) not in cache:
                        cache[("C", b)] = C[b][: gp.n].cpu().numpy()
                s2_all, rows_l = cache["s2"], rows - gp.row0
                cpick = np.stack([cache[("C", b)] for b in pp]) if pp else np.zeros((0, gp.n))
            else:
                s2_all, rows_l = gp._full(gp.s2), rows
                cpick = np.stack([gp._full(C[b]) for b in pp]) if pp else np.zeros((0, gp.n_total))

            def moments(j0, j1):
                """Mean [n, nr] and covariance [n, nr, nr] of (members so far, candidate) for live candidates j0 .. j1-1."""
                rw, rl = rows[j0 - jl0:j1 - jl0], rows_l[j0 - jl0:j1 - jl0]
                mean = np.empty((len(rw), nr))
                cov = np.empty((len(rw), nr, nr))
                mean[:, : nr - 1] = e_mu[pp][None, :] if pp else 0
                cov[:, : nr - 1, : nr - 1] = e_sig[np.ix_(pp, pp)][None] if pp else 0
                mean[:, nr - 1] = mu_all[rw]
                cov[:, nr - 1, nr - 1] = s2_all[rl]
                if pp:
                    cov[:, : nr - 1, nr - 1] = cpick[:, rl].T
                    cov[:, nr - 1, : nr - 1] = cov[:, : nr - 1, nr - 1]
                if subset_mode:
                    for j in np.flatnonzero(in_e[live[j0:j1]] >= 0):    # members of the base set: covariances from E itself
                        idx = pp + [int(in_e[live[j0 + j]])]
                        mean[j] = e_mu[idx]
                        cov[j] = e_sig[np.ix_(idx, idx)]
                elif nr == 1:
                    cov[:, 0, 0] = np.maximum(0, cov[:, 0, 0])     # first step: predict_stored(cov_mode='diag') (ital.py:558)
                return mean, cov
        weights = (1 << np.arange(nr - 1, -1, -1)).astype(np.uint32)   # variable v at bit nr-1-v
        if fb_mc:
            if fb_mode == 1:
                vals = np.array([1, -1])
                pr = np.array([1.0 - self.mistake_prob, self.mistake_prob])
            else:
                vals = np.array([0, 1, -1])
                pr = np.array([1.0 - self.label_prob, self.label_prob * (1.0 - self.mistake_prob),
                               self.label_prob * self.mistake_prob])
            cdf = pr.cumsum()
            cdf /= cdf[-1]

        def transform(z, j0, j1):
            mean, cov = moments(j0, j1)
            _, sv, vt = np.linalg.svd(cov)
            x = z @ (np.sqrt(sv)[:, :, None] * vt) + mean[:, None, :]
            return ((x > 0) * weights).sum(axis=2).astype(np.uint32)

        def draw_rel(j0, j1, z=None):
            """multivariate_normal.rvs for live candidates j0..j1-1: numpy's legacy generator = standard normals in
            order, then x = z . (sqrt(s) v) + mean with (u, s, v) = svd(cov).  The per-candidate LAPACK calls are
            independent of each other: large stacks are cut into slices for a thread pool (numpy's gufuncs release the
            GIL; the result is the same bits as one call)."""
            if z is None:
                z = np.random.standard_normal((j1 - j0, npat, nr))
            n = j1 - j0
            workers = min(_HOST_THREADS, n * nr * nr // 65536)
            if workers < 2:
                return transform(z, j0, j1)
            cuts = np.linspace(0, n, 2 * workers + 1).astype(np.int64)
            parts = list(_host_pool().map(lambda ab: transform(z[ab[0]:ab[1]], j0 + ab[0], j0 + ab[1]),
                                          zip(cuts[:-1], cuts[1:])))
            return np.concatenate(parts)

        def draw_fb(pats):
            """np.random.choice(vals, (nfb, nr), p) for every pattern of `pats` [..., npat]: uniforms in order."""
            u = np.random.random_sample(pats.shape + (nfb, nr))
            smp = vals[cdf.searchsorted(u, side="right")]
            relv = ((pats[..., None] >> np.arange(nr - 1, -1, -1)) & 1).astype(bool)     # [..., npat, nr]
            smp = np.where(relv[..., None, :], smp, -smp)
            vbit = (1 << np.arange(nr)).astype(np.uint32)
            nz = ((smp != 0) * vbit).sum(axis=-1).astype(np.uint32)
            ps = ((smp > 0) * vbit).sum(axis=-1).astype(np.uint32)
            return nz | (ps << np.uint32(16))

        L = len(live)
        enum_pats = np.arange(npat, dtype=np.uint32)
        if rel_mc and not fb_mc:
            # the legacy generator cannot jump, so every rank walks the whole stream of normals -- but only its own
            # candidates' are computed (the others are skipped: raw draws and accept tests, ital_np_legacy_normals), and
            # the decompositions are done for those alone
            if z_rel is None:
                z_loc = self._walk_normals(L, jl0, jl1, npat * nr).reshape(-1, npat, nr)
            else:
                g0, z = z_rel                                       # drawn ahead for live ranks g0 .. g0 + len(z) - 1
                z_loc = z[jl0 - g0:jl1 - g0]
            if chunks > 0 and local is not None and jl1 > jl0:
                draws = np.zeros(P, dtype=np.int64)
                draws[live] = npat * (npre_draws[live] + nfb * d_full[live])

                def ranges():
                    # a SHORT first range: the GPU idles while the host decomposes it (the pick of the step before, the new
                    # member's covariance column and the SVDs of its candidates: 0.17 s per step at 1M x 512 with four equal
                    # ranges, 2.8 s of a 101 s round), every later range is decomposed under the lattice sums of the one before
                    # -- as long as a range is not much longer than the one before (the host decomposes ~1.4 M candidates per
                    # second on 16 threads, the GPU integrates 1.6 M (7 variables) .. 47 k (16) per second): sizes 1 : 2 : 4 : ...
                    cuts = _range_cuts(jl0, jl1, chunks)
                    for a, b_ in zip(cuts[:-1], cuts[1:]):
                        lo = local[0] if a == jl0 else int(live[a])
                        hi = local[1] if b_ == jl1 else int(live[b_])
                        rows = np.zeros((hi - lo, npat), dtype=np.uint32)
                        rows[live[a:b_] - lo] = draw_rel(int(a), int(b_), z_loc[a - jl0:b_ - jl0])
                        yield lo, hi, rows, bool(b_ == cuts[-1])
                return ranges(), None, draws
            rel_live = np.zeros((L, npat), dtype=np.uint32)
            if jl1 > jl0:
                rel_live[jl0:jl1] = draw_rel(jl0, jl1, z_loc)
        elif fb_mc and not rel_mc:
            fb_live = draw_fb(np.broadcast_to(enum_pats, (L, npat)))
        else:
            rel_live = np.empty((L, npat), dtype=np.uint32)
            fb_live = np.empty((L, npat, nfb), dtype=np.uint32)
            for j in range(L):                                       # the two samplers interleave per candidate
                rel_live[j] = draw_rel(j, j + 1)[0]
                fb_live[j] = draw_fb(rel_live[j])
        if rel_mc:
            rel_arr = np.zeros((P, npat), dtype=np.uint32)
            rel_arr[live] = rel_live
        draws = np.zeros(P, dtype=np.int64)
        if fb_mc:
            fb_arr = np.zeros((P, npat, nfb), dtype=np.uint32)
            fb_arr[live] = fb_live
            calls = ((fb_live & 0xffff) != 0).sum(axis=(1, 2))      # all-zero feedback samples make no call
            draws[live] = npat * npre_draws[live] + calls * d_full[live]
        else:
            draws[live] = npat * (npre_draws[live] + nfb * d_full[live])
        return rel_arr, fb_arr, draws
