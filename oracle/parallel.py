"""ORACLE (test infrastructure): the reference's multiprocessing evaluation (ital/ital.py:124-126, :619-624) on
top of the serial restatement -- a new fork pool per greedy step, candidates mapped over all host cores.
Used only by bench.py's `cpu_baseline` legs (kind "port"): the headline workload whole (fetch_unlabelled_parallel) and the
larger configurations on a bounded sample, extrapolated (fetch_unlabelled_sampled).  Like the reference's parallel mode its picks for
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


def effective_cores():
    """Host cores this process may really use: the smallest of os.cpu_count(), the scheduler affinity mask and the cgroup's
    CPU quota (a container on a 256-core host with a 16-CPU share runs 256 workers no faster than 16)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(float(parts[0]) / float(parts[1]) + 0.5)))
            else:
                quota = int(parts[0])
                if quota > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, int(quota / float(f.read()) + 0.5)))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def fetch_unlabelled_sampled(learner, k, processes=None, budget_s=15.0, n_max=2048):
    """The same scheme on a BOUNDED sample, for the configurations whose full round would take a CPU hours ("timed on
    a candidate subsample ... flagged extrapolated" -- the per-candidate cost does not depend on N, reference
    ital/ital.py:183-224): per greedy step a new fork pool, a pilot of one candidate per worker, then as many more candidates as
    the step's share of `budget_s` allows (at most n_max in all); the step's pick is the arg-max of the scored sample.
    Returns per step t: dict(t, scored, fork_s (pool start-up), map_s (scoring wall time), per_cand_s = map_s / scored)."""
    import time
    processes = processes or effective_cores()
    cand = learner.get_unseen()
    k = min(k, len(cand))
    learner._ce_subset = None
    state = _Appended(learner)
    ctx = mp.get_context("fork")
    steps = []
    t_begin = time.perf_counter()
    for step in range(k):
        share = max((budget_s - (time.perf_counter() - t_begin)) / (k - step), 0.0)
        t0 = time.perf_counter()
        with ctx.Pool(processes, initializer=_init, initargs=(state,)) as pool:
            pool.map(_score, [])                                  # (workers up)
            t1 = time.perf_counter()
            s0 = min(processes, len(cand), n_max)
            vals = pool.map(_score, cand[:s0], chunksize=1)
            t2 = time.perf_counter()
            wave = max(t2 - t1, 1e-6)                             # one candidate per worker: wall time of one "wave"
            more = int((share - (t2 - t0)) / wave) * processes
            s1 = max(0, min(more, n_max - s0, len(cand) - s0))
            if s1:
                vals += pool.map(_score, cand[s0:s0 + s1], chunksize=1)
            t3 = time.perf_counter()
        scored = s0 + s1
        steps.append(dict(t=step + 1, scored=scored, fork_s=t1 - t0, map_s=t3 - t1, per_cand_s=(t3 - t1) / scored))
        best = int(np.argmax(vals))
        state.append(cand[best])
        del cand[best]
    return steps
