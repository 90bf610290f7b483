"""Learner API -- the drop-in boundary (mirror of reference ital/retrieval_base.py `ActiveRetrievalBase`).

Same constructor arguments, methods (fit / reset / top_results / get_unseen / fetch_unlabelled / update /
partition_feedback), attributes (gp, rel_mean, relevant_ids, irrelevant_ids, unnameable_ids, rounds, data,
queries) and error behaviour; the GP behind it is the streaming MI355X one (ital_amd.gp).
Keyword-only extras: device, rank, world, group (row sharding over GPUs).

This file IS the drop-in boundary: `partition_feedback` and `updated_prediction` have one correct form -- the reference's
(retrieval_base.py:129-193: same branches, same error message, same ordering of the ids) -- and restate it; everything
below them (GP state, candidate bookkeeping, the device path) is this package's own.
"""
import numpy as np

from .gp import GaussianProcess


class IdSet(set):
    """A set that counts its in-place changes (`changes`): what the learners keep their relevant / irrelevant / unnameable
    ids in (public attributes of the reference, retrieval_base.py:50-52).  Behaves as a set everywhere else."""

    _serial = 0      # process-wide: a freshly wrapped set never starts at a token value an earlier wrapper held

    def __init__(self, *a):
        set.__init__(self, *a)
        IdSet._serial += 1 << 20
        self.changes = IdSet._serial


def _counting(name):
    base = getattr(set, name)

    def method(self, *a, **kw):
        self.changes += 1
        return base(self, *a, **kw)
    method.__name__ = name
    method.__doc__ = base.__doc__
    return method


for _name in ("add", "discard", "remove", "pop", "clear", "update", "difference_update", "intersection_update",
              "symmetric_difference_update", "__ior__", "__iand__", "__isub__", "__ixor__"):
    setattr(IdSet, _name, _counting(_name))
del _name


class UnseenList(object):
    """The ascending list of samples without feedback (reference retrieval_base.py:78-87) as an ascending base array plus a
    short sorted list of ids taken out of it since: what a round of the retrieval loop needs from the list -- its length,
    the number of entries below a row boundary, the entries of one rank's rows -- costs O(k log N) per round instead of a
    copy of the N-sized array (8 MB at a million samples, on the critical path of every rank whatever the number of ranks).
    The array itself is only formed for callers that read it (get_unseen(), the options that restrict or subsample the
    list).  `version` counts the removals: the candidate list a device holds is described by the version it was built
    from (ital.py)."""

    REBASE_AT = 4096         # removed ids from which a materialisation also becomes the new base

    def __init__(self, base):
        self.base = base                     # ascending int64 array, never written to
        self.removed = []                    # ascending python ints, all members of base
        self.version = 0
        self.last_removed = ()               # the ids of the last remove() (version - 1 -> version)
        self._flat = base                    # the list as an array (None: not formed since the last removal)
        self._prev_flat = None               # (array before the last removal, the ids it lost): one pass forms the next one

    def __len__(self):
        return len(self.base) - len(self.removed)

    def count_below(self, x):
        """Number of list entries < x (= list position of the first entry >= x)."""
        import bisect
        return int(np.searchsorted(self.base, x)) - bisect.bisect_left(self.removed, x)

    def remove(self, ids):
        """Takes `ids` (distinct) out of the list; False (list unchanged) when one of them is not on it."""
        import bisect
        ids = sorted(set(int(i) for i in ids))
        if not ids:
            return True
        arr = np.asarray(ids, dtype=np.int64)
        at = np.searchsorted(self.base, arr)
        if np.any(at >= len(self.base)) or np.any(self.base[np.minimum(at, len(self.base) - 1)] != arr):
            return False
        for i in ids:
            j = bisect.bisect_left(self.removed, i)
            if j < len(self.removed) and self.removed[j] == i:
                return False
        prev = self._flat
        for i in ids:
            bisect.insort(self.removed, i)
        self.version += 1
        self.last_removed = tuple(ids)
        self._flat, self._prev_flat = None, (prev, arr) if prev is not None else None
        return True

    def in_rows(self, row0, row1):
        """The list entries in [row0, row1) as an array: O(entries returned)."""
        import bisect
        a, b = int(np.searchsorted(self.base, row0)), int(np.searchsorted(self.base, row1))
        part = self.base[a:b]
        r0, r1 = bisect.bisect_left(self.removed, row0), bisect.bisect_left(self.removed, row1)
        if r1 > r0:
            part = np.delete(part, np.searchsorted(part, np.asarray(self.removed[r0:r1], dtype=np.int64)))
        return part

    def array(self):
        """The whole list as an int64 array (treat as read-only; the same object until the next removal)."""
        if self._flat is None:
            prev = self._prev_flat
            if prev is not None:             # the array before the last removal is at hand: one pass over it
                self._flat = np.delete(prev[0], np.searchsorted(prev[0], prev[1]))
            else:
                self._flat = np.delete(self.base, np.searchsorted(self.base, np.asarray(self.removed, dtype=np.int64)))
            self._prev_flat = None
            if len(self.removed) >= self.REBASE_AT:
                self.base, self.removed = self._flat, []
        return self._flat


class ActiveRetrievalBase(object):

    #: initial capacity of the labelled set on the device (None: GaussianProcess picks it; it grows on demand)
    gp_capacity = None

    def __init__(self, data=None, queries=[], length_scale=0.1, var=1.0, noise=1e-6, *, device=None, rank=0,
                 world=1, group=None):
        self.length_scale = length_scale
        self.var = var
        self.noise = noise
        self.device, self.rank, self.world, self.group = device, rank, world, group
        self.fit(data, queries)

    def fit(self, data, queries=[]):
        """reference retrieval_base.py:34-45"""
        self.data = data
        self.queries = queries
        if self.data is not None:
            self.gp = GaussianProcess(self.data, self.length_scale, self.var, self.noise, device=self.device,
                                      rank=self.rank, world=self.world, group=self.group, capacity=self.gp_capacity)
            self.reset()
        else:
            self.gp = None

    def reset(self):
        """reference retrieval_base.py:48-61"""
        self.rounds = 0
        self.relevant_ids = set()
        self.irrelevant_ids = set()
        self.unnameable_ids = set()
        self._last_batch = None
        self.gp.reset()
        if len(self.queries) > 0:
            n = len(self.data)
            self.gp.update_points(self.queries, [1] * len(self.queries), ind=list(range(n, n + len(self.queries))))
            self._fitted = True
        else:
            self._fitted = False

    # ------------------------------------------------------------------ checkpoint / resume (the
    # reference has none: its state dies with the process)
    def state_dict(self):
        """Everything needed to continue a retrieval session in another process: the labelled set in insertion order with
        the sizes of the update() calls that built it, the id sets, the round counter and the position of the replayed
        mvndst stream and the state of numpy's global legacy generator (what MCMI_min(subsample), the change-estimation
        subset and the Monte-Carlo switches draw from, as the reference does).  Small (O(labelled samples)): the whitened
        block V is rebuilt on load, not stored -- the replay appends sample by sample where the session appended staged
        batches, so the means agree to ~1e-15, not bit for bit (a pick at an exact numerical tie may differ)."""
        from . import mvn_stream
        gp = self.gp
        return dict(version=1, n=len(self.data), hyper=(float(self.length_scale), float(self.var), float(self.noise)),
                    ind=[int(i) for i in gp.ind], y=(gp.y.copy() if gp.y is not None else np.zeros(0)),
                    appends=list(gp.appends), relevant_ids=sorted(self.relevant_ids),
                    irrelevant_ids=sorted(self.irrelevant_ids), unnameable_ids=sorted(self.unnameable_ids),
                    rounds=int(self.rounds), mvn_stream=(tuple(mvn_stream.GLOBAL.state), int(mvn_stream.GLOBAL.draws)),
                    np_random=np.random.get_state())

    def load_state_dict(self, sd, restore_stream=True):
        """Restores a session saved by state_dict() on a learner constructed over the same data (and queries) with the same
        hyper-parameters: reset(), then the same sequence of appends to the GP (same kernels, same order: the predictive
        means equal the saved session's to ~1e-15), the id sets, the round counter and -- unless told otherwise -- the
        process-wide positions of the two random streams (mvndst's, numpy's global generator).  Collective on several ranks,
        like update()."""
        from . import mvn_stream
        if sd.get("version") != 1 or sd["n"] != len(self.data):
            raise ValueError("state_dict of another data set / version")
        if tuple(sd["hyper"]) != (float(self.length_scale), float(self.var), float(self.noise)):
            raise ValueError("state_dict was saved with other hyper-parameters: %r" % (sd["hyper"],))
        self.reset()
        at = self.gp.m                       # queries, if any, are back in place after reset()
        ind, y = list(sd["ind"]), np.asarray(sd["y"], dtype=np.float64)
        for c in sd["appends"]:
            if c < 0:
                continue                     # the queries' own append: done by reset()
            if c:
                self.gp.update(ind[at:at + c], y[at:at + c])
            at += c
        if self.gp.ind != ind:
            raise ValueError("state_dict does not match this learner's queries")
        self.relevant_ids = set(sd["relevant_ids"])
        self.irrelevant_ids = set(sd["irrelevant_ids"])
        self.unnameable_ids = set(sd["unnameable_ids"])
        self.rounds = int(sd["rounds"])
        self._fitted = self.gp.m > 0
        if restore_stream:
            mvn_stream.GLOBAL.state, mvn_stream.GLOBAL.draws = tuple(sd["mvn_stream"][0]), int(sd["mvn_stream"][1])
            if sd.get("np_random") is not None:
                np.random.set_state(sd["np_random"])
        return self

    @property
    def rel_mean(self):
        """Predictive mean of every sample (numpy; None before the first update, as reference retrieval_base.py:61).
        gp.update() replicates the means on every rank's device (asynchronously, all ranks take part in update()); this
        attribute only downloads that vector when somebody reads it -- a local operation, safe to read on one rank."""
        return self.gp.mean_host() if self._fitted else None

    def top_results(self, k=None):
        """reference retrieval_base.py:64-75.  Up to ITAL_TOPK_MAX results are selected on the device (ital_topk, exact;
        ties by descending index as the reversal of a stable ascending sort gives); the complete ranking (k = None) is
        N-sized by definition and sorts the downloaded means on the host."""
        from ._lib import ITAL_TOPK_MAX
        if k is not None and self._fitted and 1 <= int(k) <= min(ITAL_TOPK_MAX, len(self.data)):
            return self.gp.topk_mean(int(k))
        ind = np.argsort(self.rel_mean, kind="stable")[::-1]
        return ind[:k] if k is not None else ind

    def get_unseen(self):
        """reference retrieval_base.py:78-87 (ascending sample indices, python ints)."""
        return self._unseen_array().tolist()

    # the three id sets are public attributes, as in the reference; assigning a new set to one of them (rather than going
    # through update()) invalidates the derived candidate bookkeeping
    def _set_ids(self, name, value):
        # kept as a set that counts its in-place changes (`IdSet`): the candidate bookkeeping notices ANY change made behind
        # update()'s back, also one that leaves the sizes as they were (an id swapped for another)
        # NOTE (differs from the reference, which stores the caller's object, retrieval_base.py:50-52): a plain set is COPIED
        # into an IdSet -- later changes of the caller's own set object are not seen by the learner; change `learner.<name>`
        # itself (in place, or by assigning again).  INTEGRATION.md, "id sets".
        self.__dict__[name] = value if isinstance(value, IdSet) else IdSet(value)
        self.__dict__["_unseen"] = None

    def _get_ids(self, name):
        try:
            return self.__dict__[name]
        except KeyError:
            raise AttributeError(name[1:]) from None       # before fit() / reset(), as an attribute never assigned

    relevant_ids = property(lambda self: self._get_ids("_relevant_ids"), lambda self, v: self._set_ids("_relevant_ids", v))
    irrelevant_ids = property(lambda self: self._get_ids("_irrelevant_ids"), lambda self, v: self._set_ids("_irrelevant_ids", v))
    unnameable_ids = property(lambda self: self._get_ids("_unnameable_ids"), lambda self, v: self._set_ids("_unnameable_ids", v))

    def _id_sizes(self):
        """Token of the three id sets' state: sizes and change counters."""
        sets = (self.relevant_ids, self.irrelevant_ids, self.unnameable_ids)
        return tuple(len(q) for q in sets) + tuple(getattr(q, "changes", -1) for q in sets)

    def _unseen_list(self):
        """The samples without feedback as an UnseenList, kept between calls: update() takes the newly seen samples out of
        it (O(k log N)).  Rebuilt from the three id sets (N-sized work) when they were changed behind update()'s back:
        a set assigned anew, or sizes that no longer match what update() left."""
        u = self.__dict__.get("_unseen")
        if u is not None and u[1] == self._id_sizes():
            return u[0]
        seen = self.relevant_ids | self.irrelevant_ids | self.unnameable_ids
        if not seen:
            arr = np.arange(len(self.data), dtype=np.int64)
        else:
            mask = np.ones(len(self.data), dtype=bool)
            mask[np.fromiter(seen, dtype=np.int64, count=len(seen))] = False
            arr = np.flatnonzero(mask)
        lst = UnseenList(arr)
        self.__dict__["_unseen"] = (lst, self._id_sizes())
        return lst

    def _unseen_array(self):
        """Ascending indices of the samples without feedback (int64 array; treat as read-only)."""
        return self._unseen_list().array()

    def _unseen_after(self, new_ids, sizes_before):
        """Called by update() once the id sets hold `new_ids` (samples that had no feedback before): the kept list loses
        them (the device keeps its candidate list the same way, ital.py)."""
        u = self.__dict__.get("_unseen")
        if u is None or u[1] != sizes_before:
            self.__dict__["_unseen"] = None
            return
        ids = set(int(i) for i in new_ids)
        if not u[0].remove(ids):
            self.__dict__["_unseen"] = None        # feedback for something that was not a candidate: rebuild next time
            return
        self.__dict__["_unseen"] = (u[0], self._id_sizes())

    def fetch_unlabelled(self, k):
        raise NotImplementedError('fetch_unlabelled() has to be implemented in a derived class.')

    def update(self, feedback):
        """reference retrieval_base.py:105-126"""
        rel, irr, unnameable = self.partition_feedback(feedback)
        sizes_before = self._id_sizes()
        if len(rel) + len(irr) > 0:
            self.gp.update(rel + irr, np.concatenate((np.ones(len(rel)), -1 * np.ones(len(irr)))),
                           row_cache=self._batch_rows())
            self._fitted = True
            self.relevant_ids.update(rel)
            self.irrelevant_ids.update(irr)
            self.rounds += 1
        new_unnameable = [i for i in unnameable if i not in self.unnameable_ids]
        self.unnameable_ids.update(unnameable)
        self._unseen_after(rel + irr + new_unnameable, sizes_before)

    def _batch_rows(self):
        """Feature rows of the last fetched batch as kept (replicated) in the device batch state, or None."""
        last = getattr(self, "_last_batch", None)
        if not last:
            return None
        b, picks = last
        return b["XB"], {int(i): slot for slot, i in enumerate(picks)}

    def updated_prediction(self, feedback, test_ind, cov_mode='full'):
        """Prediction after a simulated update with `feedback`, without performing it (reference
        retrieval_base.py:129-164)."""
        rel, irr, _ = self.partition_feedback(feedback)
        if len(rel) + len(irr) == 0:
            return self.gp.predict_stored(test_ind, cov_mode=cov_mode)
        rel.sort()
        irr.sort()
        return self.gp.updated_prediction(rel + irr, np.concatenate((np.ones(len(rel)), -1 * np.ones(len(irr)))),
                                          test_ind, cov_mode=cov_mode)

    def partition_feedback(self, feedback):
        """reference retrieval_base.py:167-193"""
        rel, irr, unnameable = [], [], []
        for i, fb in feedback.items():
            if fb > 0:
                if i in self.irrelevant_ids:
                    raise RuntimeError('Cannot change feedback once given.')
                elif i not in self.relevant_ids:
                    rel.append(i)
            elif fb < 0:
                if i in self.relevant_ids:
                    raise RuntimeError('Cannot change feedback once given.')
                elif i not in self.irrelevant_ids:
                    irr.append(i)
            else:
                unnameable.append(i)
        return rel, irr, unnameable
