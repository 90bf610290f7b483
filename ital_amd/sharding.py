"""Row / candidate sharding across ranks and the per-greedy-step record exchange.

Pure host logic (no GPU needed): used by ital_amd.gp / ital_amd.ital on the GPU box over RCCL ("nccl"
backend) and exercised on CPU with gloo in tests/test_dist_gloo.py.
"""
import numpy as np


def row_range(n_total, world, rank):
    """Contiguous row block [row0, row1) of `rank`; blocks differ by at most one row and tile [0, n_total)."""
    return n_total * rank // world, n_total * (rank + 1) // world


class ShardedRows(object):
    """Stand-in for the n x d data matrix on a rank that holds only its own row block (the reference keeps the whole
    matrix in every worker; at 1M x 512 x 8 ranks that is 32 GB of host memory for rows nobody reads).  `local` holds
    rows [row0, row1) of an n_total x d matrix, (row0, row1) = row_range(n_total, world, rank)."""

    def __init__(self, local, n_total, row0):
        self.local = np.asarray(local, dtype=np.float64)
        self.n_total, self.row0, self.row1 = int(n_total), int(row0), int(row0) + len(self.local)
        self.shape = (self.n_total, self.local.shape[1])
        self.ndim = 2

    def __len__(self):
        return self.n_total


def shard_candidates(candidates, row0, row1, ascending=False):
    """Positions of the global candidate list that fall into this rank's rows.
    `ascending`: the list is known to be sorted (the plain get_unseen() order): two binary searches instead of a pass over it.

    Returns (global data indices of the local positions, list position of the first one, explicit list positions or
    None).  With the ascending `get_unseen()` order the local positions are one contiguous run of the list and
    `global position = pos_offset + local position`; after the `top_candidates` restriction (np.argpartition order,
    reference ital/ital.py:111-117) they are not, and the third value carries the list position of every local one.
    The arg-max tie-break and the replay of the mvndst stream are keyed on list positions either way."""
    cand = np.asarray(candidates, dtype=np.int64)
    if ascending:
        lo, hi = int(np.searchsorted(cand, row0)), int(np.searchsorted(cand, row1))
        return cand[lo:hi], (lo if hi > lo else 0), None
    mine = np.flatnonzero((cand >= row0) & (cand < row1))
    first = int(mine[0]) if len(mine) else 0
    if len(mine) and not np.array_equal(mine, np.arange(first, first + len(mine))):
        return cand[mine], first, mine.astype(np.int64)
    return cand[mine], first, None


def contiguous_runs(candidates, n_total, world, ascending=False):
    """True when the list positions of EVERY rank's candidates form one contiguous run of the list (then
    `shard_candidates` returns no explicit positions on any rank).  A function of the list alone: all ranks agree."""
    if world == 1 or ascending:
        return True
    cand = np.asarray(candidates, dtype=np.int64)
    bounds = np.array([row_range(n_total, world, r)[1] for r in range(world)], dtype=np.int64)
    owner = np.searchsorted(bounds, cand, side="right")
    return bool(np.all(np.diff(owner) >= 0))


def _host_staged(t, group):
    """gloo moves host memory only: device tensors are staged through the host for it (CPU tests, and the
    two-ranks-on-one-GPU parity test); RCCL ("nccl") takes device pointers directly."""
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend(group) == "gloo"


_RAW_COMMS = {}          # (id(group), device) -> (group, ncclComm_t as int or None, why the group stays on torch.distributed)
_log = None


def _logger():
    global _log
    if _log is None:
        import logging
        _log = logging.getLogger("ital_amd.sharding")
    return _log


def _agree(flag, group, device):
    """min over the ranks of a 0/1 flag, through torch.distributed (the decision every rank must take alike)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


def raw_comm(group, device):
    """This rank's ncclComm_t of an "nccl" (RCCL) process group as an integer -- the communicator torch.distributed itself
    uses (ProcessGroupNCCL._comm_ptr) -- or None: other backends, a torch without that accessor, ITAL_RAW_COMM=0, a
    communicator that fails the checks below ON ANY RANK.  With it the library issues ncclAllGather itself, on the
    caller's stream (ital_select_exchange, ital_fetch_round): no trip through torch.distributed's Python and stream
    hand-over per exchange.  Operations on the communicator stay ordered: torch's own collectives wait for the current
    stream and the current stream waits for them (async_op=False everywhere in this package).

    COLLECTIVE on first use per group (every rank of the group must call it at the same point -- the first exchange of a
    learner is such a point): whether the ranks leave torch.distributed is decided TOGETHER, never by a rank alone.  (1) a
    torch all-reduce (MIN) of "this rank wants to and may" (backend, ITAL_RAW_COMM) -- it also connects the communicator;
    (2) every rank reads its communicator and asks the library what RCCL says about it (ital_exchange_info: size and rank
    must be the group's), all-reduce (MIN) of the outcome; (3) only if every rank passed: one small all-gather through
    ital_select_exchange, checked against what every rank must receive, all-reduce (MIN) of that.  A rank that fails a
    step says why (logging, `raw_comm_reason`), and ALL ranks stay on torch.distributed: nobody is left waiting in a raw
    all-gather the others never enter."""
    import os
    import torch
    import torch.distributed as dist
    if group is None or not dist.is_initialized():
        return None
    key = (id(group), str(device))
    if key in _RAW_COMMS:
        return _RAW_COMMS[key][1]
    if dist.get_backend(group) != "nccl":          # (a property of the group: the same on every rank)
        _RAW_COMMS[key] = (group, None, "backend %s moves the records through torch.distributed" % dist.get_backend(group))
        return None
    dev = torch.device(device)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    comm, why = None, None
    if os.environ.get("ITAL_RAW_COMM") == "0":
        why = "ITAL_RAW_COMM=0 on rank %d" % rank
    backend = None
    if why is None:
        try:
            backend = group._get_backend(dev)
            if not hasattr(backend, "_comm_ptr"):
                why = "this torch's ProcessGroupNCCL has no _comm_ptr()"
        except Exception as e:      # noqa: BLE001
            why = "no NCCL backend object for %s: %r" % (dev, e)
    ok = _agree(why is None, group, dev)           # step 1 (the all-reduce itself connects the communicator)
    if ok:
        try:
            comm = int(backend._comm_ptr()) or None
            if comm is None:
                why = "rank %d: the communicator is not connected (null _comm_ptr)" % rank
            else:
                from . import _lib
                import ctypes
                w, r = ctypes.c_int(-1), ctypes.c_int(-1)
                how = ctypes.create_string_buffer(600)
                _lib.check(_lib.lib().ital_exchange_info(comm, ctypes.byref(w), ctypes.byref(r), how, 600))
                if (w.value, r.value) != (world, rank):
                    why = ("rank %d: RCCL numbers this communicator rank %d of %d, the process group rank %d of %d"
                           % (rank, r.value, w.value, rank, world))
        except Exception as e:      # noqa: BLE001
            why = "rank %d: %s" % (rank, e)
        ok = _agree(why is None, group, dev)       # step 2
    if ok:
        try:
            probe = _raw_comm_probe(dev, world, rank, comm)
            if not probe:
                why = "rank %d: the probing all-gather on the raw communicator returned other data than expected" % rank
        except Exception as e:      # noqa: BLE001
            why = "rank %d: the probing all-gather on the raw communicator failed: %s" % (rank, e)
        ok = _agree(why is None, group, dev)       # step 3
    if not ok:
        why = why or "another rank of the group cannot use its raw communicator (its log says why)"
        _logger().warning("ital_amd: record exchange of this process group stays on torch.distributed: %s", why)
        comm = None
    _RAW_COMMS[key] = (group, comm, why)           # the group is kept alive with its entry: ids are not reused under it
    return comm


def raw_comm_reason(group, device):
    """Why `raw_comm` keeps this group on torch.distributed (None: it does not, or was not asked yet)."""
    e = _RAW_COMMS.get((id(group), str(device)))
    return e[2] if e else None


def _raw_comm_probe(dev, world, rank, comm):
    """One small all-gather through ital_select_exchange on the communicator, checked against what every rank must
    receive (rank r contributes r + 1).  Entered by all ranks or by none (raw_comm agrees on that first)."""
    import torch
    from . import _lib
    send = torch.full((2,), float(rank + 1), dtype=torch.float64, device=dev)
    recv = torch.zeros((world, 2), dtype=torch.float64, device=dev)
    _lib.check(_lib.lib().ital_select_exchange(send.data_ptr(), recv.data_ptr(), 2, comm,
                                               torch.cuda.current_stream(dev).cuda_stream))
    want = torch.arange(1, world + 1, dtype=torch.float64).repeat_interleave(2).reshape(world, 2)
    return bool(torch.equal(recv.cpu(), want))


class ExchangeError(RuntimeError):
    """The record exchange of a greedy step did not complete: a deadline passed with the collective still pending (a rank
    that died or never entered it), or RCCL reported an asynchronous error on the communicator.  The process should end
    (a process that has touched the GPU is not restarted in place); the message names rank, world, transport and what the
    rank was waiting for."""


def exchange_timeout_s():
    """Seconds a rank waits for a round's collectives before it gives up (ITAL_EXCHANGE_TIMEOUT_S, default 120; <= 0: no
    deadline -- the behaviour before round 5)."""
    import os
    try:
        return float(os.environ.get("ITAL_EXCHANGE_TIMEOUT_S", "120"))
    except ValueError:
        return 120.0


def comm_error(comm):
    """None, or what RCCL's progress thread reports on the raw communicator (ital_exchange_error -> ncclCommGetAsyncError)."""
    if not comm:
        return None
    import ctypes
    from . import _lib
    err = ctypes.c_int(-1)
    rc = _lib.lib().ital_exchange_error(comm, ctypes.byref(err))
    if rc == -5:
        return _lib.lib().ital_last_error().decode()
    return None          # 0: healthy; -38: this RCCL has no such entry point (nothing to poll)


def await_download(tensor, what, group=None, device=None, comm=None, rank=0, world=1, transport=None, timeout_s=None, pinned=None,
                   step_estimate_s=0.0):
    """`tensor.cpu()` with a deadline on LACK OF PROGRESS: the device-to-host copy of a round's picks is enqueued behind the
    round's kernels and collectives; the host waits for it with an event it QUERIES (never an unbounded synchronize), polls
    the communicator's asynchronous error state meanwhile and raises ExchangeError when RCCL reports a failure or when
    NOTHING HAS MOVED for the deadline.  "Moved": every 0.25 s the buffer itself (the picks of the greedy steps resolved so
    far + the status word) is read through a side stream -- a copy engine, not ordered behind the round's kernels -- and the
    clock restarts whenever its content has changed, i.e. whenever another greedy step has been resolved.  (Round 5 bounded the
    TOTAL wait, which covers the GPU compute of all k steps: a healthy round of k = 8 over ~2 M rows per rank takes longer than
    120 s and was told that "another rank died".)  The deadline is ITAL_EXCHANGE_TIMEOUT_S (120 s), or 3 x `step_estimate_s`
    -- the caller's upper estimate of ONE greedy step's compute time on this rank -- if that is longer.  One rank without a
    collective: a plain synchronous copy (nothing can stall it but the device itself).
    Counterpart of the parent noticing a dead worker of the reference's Pool (ital/ital.py:124-126)."""
    import time
    import torch
    if world <= 1 and not comm:
        return tensor.cpu()
    limit = exchange_timeout_s() if timeout_s is None else timeout_s
    if limit <= 0:
        return tensor.cpu()
    limit = max(limit, 3.0 * float(step_estimate_s or 0.0))
    # (`pinned`: a dict the caller keeps -- the page-locked landing buffers and the side stream are made once, not per round)
    key = (tuple(tensor.shape), tensor.dtype)
    host = pinned.get(key) if pinned is not None else None
    if host is None:
        host = torch.empty(tensor.shape, dtype=tensor.dtype, pin_memory=True)
        if pinned is not None:
            pinned[key] = host
    host.copy_(tensor, non_blocking=True)
    done = torch.cuda.Event()
    done.record(torch.cuda.current_stream(tensor.device))
    t0 = time.perf_counter()
    t_moved = t0                    # when the buffer's content was last seen to change
    last_seen = None
    next_poll = 0.05
    spins = 0
    while not done.query():
        spins += 1
        now = time.perf_counter()
        waited = now - t0
        if waited >= next_poll:
            next_poll = waited + 0.25
            err = comm_error(comm)
            if err:
                raise ExchangeError("ital_amd: rank %d of %d: RCCL reports an error on the communicator while waiting for %s "
                                    "(transport %s): %s" % (rank, world, what, transport, err))
            if waited >= min(1.0, 0.25 * limit):       # (short rounds never pay for the snapshots)
                seen = _peek(tensor, pinned if pinned is not None else {})
                if seen is not None and seen != last_seen:
                    if last_seen is not None:
                        t_moved = now
                    last_seen = seen
        if now - t_moved > limit:
            raise ExchangeError("ital_amd: rank %d of %d: %s did not arrive and no greedy step has been resolved for %.1f s "
                                "(transport %s, %.1f s since the round was enqueued): a collective of the round is still "
                                "pending -- another rank died or never entered it -- or one greedy step's own kernels take "
                                "longer than the deadline on this rank (ITAL_EXCHANGE_TIMEOUT_S sets it; <= 0 waits for ever).  "
                                "This process should exit now." % (rank, world, what, limit, transport, waited))
        if spins > 2000:
            time.sleep(0.0002)      # a long round: stop burning the core (the first ~ms are polled back to back)
    return host.clone() if pinned is not None else host      # (the landing buffer is reused by the next round)


def _peek(tensor, cache):
    """The CURRENT content of a small device buffer (bytes), read through a side stream while the buffer's own stream is busy:
    the progress probe of await_download.  None when the copy does not come back within a second (nothing is concluded)."""
    import time
    import torch
    side = cache.get("peek_stream")
    if side is None:
        side = cache["peek_stream"] = torch.cuda.Stream(device=tensor.device)
    key = ("peek", tuple(tensor.shape), tensor.dtype)
    land = cache.get(key)
    if land is None:
        land = cache[key] = torch.empty(tensor.shape, dtype=tensor.dtype, pin_memory=True)
    with torch.cuda.stream(side):
        land.copy_(tensor, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(side)
    t0 = time.perf_counter()
    while not ev.query():
        if time.perf_counter() - t0 > 1.0:
            return None
        time.sleep(0.0005)
    return land.numpy().tobytes()


def gather_records(record, out, group=None):
    """ONE collective per greedy step: every rank contributes its fixed-size record, all ranks receive all of them.
    `record` [R], `out` [world, R] (same dtype/device)."""
    import torch.distributed as dist
    world = out.shape[0]
    if world == 1 and (group is None or not dist.is_initialized()):
        out[0].copy_(record)
        return out
    if record.is_cuda and record.dtype.itemsize == 8 and record.is_contiguous() and out.is_contiguous() \
            and record.numel() < (1 << 31):
        comm = raw_comm(group if group is not None else dist.group.WORLD, record.device)
        if comm is not None:
            import torch
            from . import _lib
            _lib.check(_lib.lib().ital_select_exchange(record.data_ptr(), out.data_ptr(), record.numel(), comm,
                                                       torch.cuda.current_stream(record.device).cuda_stream))
            return out
    if _host_staged(record, group) or not record.is_cuda:
        host = [torch_like_cpu(record) for _ in range(world)]
        _bounded(dist.all_gather(host, record.cpu(), group=group, async_op=True), "the all-gather of the ranks' records", group)
        for w in range(world):
            out[w].copy_(host[w])
        return out
    dist.all_gather_into_tensor(out.view(-1), record, group=group)   # one contiguous [world * R] receive buffer
    return out


def _bounded(work, what, group=None):
    """Waits for a host-side (gloo) collective with the exchange deadline; ExchangeError when it passes or the backend fails."""
    import datetime
    import torch.distributed as dist
    limit = exchange_timeout_s()
    try:
        if limit > 0:
            ok = work.wait(datetime.timedelta(seconds=limit))
        else:
            ok = work.wait()
    except Exception as e:      # noqa: BLE001 -- gloo raises RuntimeError / DistBackendError on a time-out or a lost peer
        raise ExchangeError("ital_amd: rank %d of %d: %s did not complete within %.1f s (%s: %s)"
                            % (dist.get_rank(group), dist.get_world_size(group), what, limit, type(e).__name__,
                               str(e).splitlines()[0] if str(e) else "")) from e
    if ok is False:
        raise ExchangeError("ital_amd: rank %d of %d: %s did not complete within %.1f s"
                            % (dist.get_rank(group), dist.get_world_size(group), what, limit))


def torch_like_cpu(t):
    import torch
    return torch.empty(t.shape, dtype=t.dtype)


def all_reduce_sum(t, group=None):
    """In-place sum over ranks (replication of rows owned by one rank: owners contribute, the others add zeros)."""
    import torch.distributed as dist
    if _host_staged(t, group):
        h = t.cpu()
        _bounded(dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group, async_op=True), "an all-reduce of replicated rows", group)
        t.copy_(h)
        return t
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def broadcast(t, src, group=None):
    """In-place broadcast of rank `src`'s tensor (group rank)."""
    import torch.distributed as dist
    src_global = dist.get_global_rank(group, src) if group is not None else src
    if _host_staged(t, group):
        h = t.cpu()
        _bounded(dist.broadcast(h, src=src_global, group=group, async_op=True), "a broadcast", group)
        t.copy_(h)
        return t
    dist.broadcast(t, src=src_global, group=group)
    return t


def all_gather_parts(loc, sizes, group=None):
    """Concatenation over ranks of 1-D shards of the given sizes."""
    import torch
    import torch.distributed as dist
    if _host_staged(loc, group):
        parts = [torch.empty(s, dtype=loc.dtype) for s in sizes]
        _bounded(dist.all_gather(parts, loc.cpu().contiguous(), group=group, async_op=True), "the all-gather of the means", group)
        return torch.cat(parts).to(loc.device)
    parts = [torch.empty(s, dtype=loc.dtype, device=loc.device) for s in sizes]
    dist.all_gather(parts, loc.contiguous(), group=group)
    return torch.cat(parts)


def winner(records, mode=0):
    """Host statement of the rule `ital_select_resolve` applies on the device (used by tests and documentation):
    records[:, 0] = score, records[:, 1] = global list position (< 0: rank had no live candidate).  mode 0: first
    maximum with NaN beating every number (np.argmax), mode 1: first minimum (np.argmin)."""
    best = -1
    for w in range(len(records)):
        v, p = records[w][0], records[w][1]
        if p < 0:
            continue
        if best < 0:
            best = w
            continue
        bv, bp = records[best][0], records[best][1]
        vn, bn = np.isnan(v), np.isnan(bv)
        if vn != bn:
            better = vn
        elif vn or v == bv:
            better = p < bp
        else:
            better = v > bv if mode == 0 else v < bv
        if better:
            best = w
    return best
