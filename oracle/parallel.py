"""ORACLE (test infrastructure): the reference's multiprocessing evaluation (ital/ital.py:124-126, :619-624) on
top of the serial restatement -- a new fork pool per greedy step, candidates mapped over all host cores.
Used only by bench.py's `cpu_baseline` leg (kind "port").  Like the reference's parallel mode its picks for
t >= 3 depend on the worker count (every worker inherits the same mvndst stream state)."""
import multiprocessing as mp
import os

import numpy as np

from .ital import _Appended

_state = None


def _init(state):
    global _state
    _state = state


def _score(i):
    return _state.score(i)


def fetch_unlabelled_parallel(learner, k, processes=None):
    """ITAL.fetch_unlabelled with `parallelized=True` semantics.  Returns (picks, number of scored candidates)."""
    processes = processes or os.cpu_count()
    cand = learner.get_unseen()
    k = min(k, len(cand))
    learner._ce_subset = None
    state = _Appended(learner)
    scored = 0
    ctx = mp.get_context("fork")
    for _ in range(k):
        with ctx.Pool(processes, initializer=_init, initargs=(state,)) as pool:
            vals = pool.map(_score, cand)
        scored += len(cand)
        best = int(np.argmax(vals))
        state.append(cand[best])
        del cand[best]
    return state.ret, scored
