"""N > 1 path on CPU: world size 2 over gloo.  Exercises the product's sharding arithmetic and the one-record-per-rank
exchange of a greedy step (ital_amd/sharding.py), and checks the winner rule against np.argmax / np.argmin."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from ital_amd import sharding  # noqa: E402

REC = 10 + 16 + 16 + 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, seen, scores, mode, ret, order=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cand = [i for i in range(n_total) if i not in seen] if order is None else list(order)
        row0, row1 = sharding.row_range(n_total, world, rank)
        loc, pos_offset, gpos = sharding.shard_candidates(cand, row0, row1)
        rec = torch.zeros(REC, dtype=torch.float64)
        if len(loc):
            vals = scores[loc]
            # local arg-extreme with the reference's rule (first extreme, NaN wins); local positions keep list order
            lp = int(np.argmax(vals) if mode == 0 else np.argmin(vals))
            gp = pos_offset + lp if gpos is None else int(gpos[lp])
            rec[0], rec[1], rec[2], rec[6], rec[7] = float(vals[lp]), gp, int(loc[lp]), rank, lp
        else:
            rec[1] = -1
        out = torch.zeros((world, REC), dtype=torch.float64)
        sharding.gather_records(rec, out, None)
        w = sharding.winner(out.numpy(), mode)
        ret[rank] = (int(out[w, 2]), out.numpy().copy(), (row0, row1, pos_offset, len(loc)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,case", [(0, "plain"), (0, "ties"), (0, "nan"), (1, "plain"), (0, "empty_rank")])
def test_two_rank_exchange(mode, case):
    rng = np.random.default_rng(11)
    n_total = 41
    scores = rng.normal(size=n_total)
    seen = {3, 20, 21}
    if case == "ties":
        scores[[5, 30, 35]] = scores.max() + 1.0       # equal maxima on both ranks: lowest list position wins
    if case == "nan":
        scores[33] = np.nan                              # a NaN beats every number (np.argmax)
        scores[7] = scores[np.isfinite(scores)].max() + 5
    if case == "empty_rank":
        seen = set(range(20, 41))                        # rank 1 has no live candidate
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, n_total, seen, scores, mode, ret), nprocs=world, join=True)
        r0, r1 = ret[0], ret[1]
    cand = [i for i in range(n_total) if i not in seen]
    vals = scores[cand]
    expect = cand[int(np.argmax(vals) if mode == 0 else np.argmin(vals))]
    assert r0[0] == expect and r1[0] == expect                      # both ranks pick the reference's sample
    assert np.array_equal(r0[1], r1[1], equal_nan=True)             # and hold identical gathered records
    (a0, b0, p0, n0), (a1, b1, p1, n1) = r0[2], r1[2]
    assert a0 == 0 and b0 == a1 and b1 == n_total                   # row blocks tile the data
    assert n0 + n1 == len(cand) and (n1 == 0 or p1 == n0)           # list positions are continuous across ranks


def _raw_comm_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # a backend that is not RCCL has no communicator to hand below the C ABI: the exchanges stay on torch.distributed
        ret[rank] = (sharding.raw_comm(dist.group.WORLD, "cpu"), sharding.raw_comm(None, "cpu"))
    finally:
        dist.destroy_process_group()


def test_no_raw_communicator_outside_rccl():
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_raw_comm_worker, args=(2, port, ret), nprocs=2, join=True)
        assert dict(ret) == {0: (None, None), 1: (None, None)}
    assert sharding.raw_comm(None, "cpu") is None      # no process group at all


class _FakeNcclBackend(object):
    """Stands in for ProcessGroupNCCL on a CPU box: hands out a 'communicator' per the test's plan."""

    def __init__(self, ptr):
        self.ptr = ptr

    def _comm_ptr(self):
        return self.ptr


def _raw_decision_worker(rank, world, port, plan, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes
        from ital_amd import _lib
        group = dist.group.WORLD
        # the protocol of sharding.raw_comm with the pieces below the C ABI replaced (no RCCL on a CPU box): gloo carries
        # the agreeing all-reduces, the backend / communicator checks follow the plan of this rank
        real_backend = dist.get_backend
        dist.get_backend = lambda g=None: "nccl"
        if plan[rank] == "env":
            os.environ["ITAL_RAW_COMM"] = "0"
        type(group)._get_backend = lambda self, dev: _FakeNcclBackend(0 if plan[rank] == "null" else 1234 + rank)

        class FakeLib(object):
            def ital_exchange_info(self, comm, w, r, how, n):
                if plan[rank] == "norccl":
                    return -38
                ctypes.cast(w, ctypes.POINTER(ctypes.c_int))[0] = world
                ctypes.cast(r, ctypes.POINTER(ctypes.c_int))[0] = (rank + 1) % world if plan[rank] == "order" else rank
                return 0

            def ital_last_error(self):
                return b"no RCCL (test)"

        real_lib = _lib.lib
        _lib.lib = lambda: FakeLib()
        probes = []
        sharding._raw_comm_probe = lambda dev, w, r, comm: probes.append(comm) or plan[rank] != "probe"
        try:
            got = sharding.raw_comm(group, "cpu")
            again = sharding.raw_comm(group, "cpu")              # cached: no further collective
        finally:
            dist.get_backend, _lib.lib = real_backend, real_lib
        ret[rank] = (got, again, len(probes), sharding.raw_comm_reason(group, "cpu"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("plan", [("ok", "ok"), ("env", "ok"), ("ok", "null"), ("norccl", "ok"), ("ok", "order"), ("probe", "ok")])
def test_leaving_torch_distributed_is_decided_by_all_ranks_together(plan):
    """sharding.raw_comm: a rank without a usable communicator (ITAL_RAW_COMM=0 on that rank only, a null _comm_ptr, no RCCL
    found, another rank order, a failed probe) keeps EVERY rank on torch.distributed -- nobody enters a raw all-gather
    alone -- and says why; with all ranks fine all of them get their communicator.  (Gloo carries the agreement here; on a
    GPU node the same code runs over RCCL, tests/test_gpu_multidevice.py.)"""
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_raw_decision_worker, args=(2, port, plan, ret), nprocs=2, join=True)
        res = dict(ret)
    if plan == ("ok", "ok"):
        assert [res[r][0] for r in (0, 1)] == [1234, 1235] and res[0][2] == res[1][2] == 1
        assert res[0][3] is None and res[1][3] is None
    else:
        bad = [r for r in (0, 1) if plan[r] != "ok"][0]
        for r in (0, 1):
            assert res[r][0] is None and res[r][1] is None and res[r][3]
            assert res[r][2] == (1 if "probe" in plan else 0)        # the probe is entered by both ranks or by neither
        assert ("rank %d" % bad) in res[bad][3] and "another rank" in res[1 - bad][3]


def test_row_range_tiles():
    for n in (1, 7, 64, 9298, 1000003):
        for world in (1, 2, 3, 8):
            edges = [sharding.row_range(n, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


def test_top_candidates_order_carries_explicit_list_positions():
    loc, first, gpos = sharding.shard_candidates([5, 30, 6, 31], 0, 20)
    assert loc.tolist() == [5, 6] and first == 0 and gpos.tolist() == [0, 2]
    loc, first, gpos = sharding.shard_candidates([5, 30, 6, 31], 20, 40)
    assert loc.tolist() == [30, 31] and first == 1 and gpos.tolist() == [1, 3]
    loc, first, gpos = sharding.shard_candidates([5, 6, 30, 31], 20, 40)
    assert loc.tolist() == [30, 31] and first == 2 and gpos is None


def test_two_rank_exchange_in_argpartition_order():
    """Candidate list in an arbitrary order (np.argpartition after top_candidates, reference ital/ital.py:116-117): ties
    are broken by list position, not by data index, on whichever rank the samples live."""
    rng = np.random.default_rng(3)
    n_total = 40
    scores = rng.normal(size=n_total)
    order = rng.permutation(n_total)[:25].tolist()
    scores[[order[4], order[17], order[9]]] = scores.max() + 1.0
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, n_total, set(), scores, 0, ret, order), nprocs=world, join=True)
        r0, r1 = ret[0], ret[1]
    expect = order[int(np.argmax(scores[order]))]
    assert expect == order[4]
    assert r0[0] == expect and r1[0] == expect


def _bench_helpers_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        same_sha, same = bench.picks_digest([[5, 9, 2, 7], [1, 3, 8, 4]])
        own_sha, own = bench.picks_digest([[5, 9, 2, 7], [1, 3, 8, 4 + rank]])
        ret[rank] = (same_sha, bench.ranks_agree(same, "cpu", world), bench.ranks_agree(own, "cpu", world),
                     bench.gather_floats(1.5 + rank, "cpu", world))
    finally:
        dist.destroy_process_group()


def test_bench_compares_the_ranks_picks():
    """bench.py at N > 1: every rank must have picked the same batches (a digest of the rounds' picks, all-gathered); one
    differing pick on one rank is seen by all of them.  (World size 2 over gloo; the GPU rehearsal of the whole line:
    profiles/r4_bench_2rank_gloo_*.json.)"""
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_bench_helpers_worker, args=(2, port, ret), nprocs=2, join=True)
        res = dict(ret)
    assert res[0][0] == res[1][0] and len(res[0][0]) == 16
    for r in (0, 1):
        assert res[r][1] is True and res[r][2] is False
        assert res[r][3] == [1.5, 2.5]


# ---------------------------------------------------------------------------------------------- failure detection (round 5)
def _stall_worker(rank, world, port, ret):
    """Rank 1 never enters the exchange (it 'died' mid-round): rank 0 must raise within the deadline, not wait for ever."""
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["ITAL_EXCHANGE_TIMEOUT_S"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rec = torch.full((REC,), float(rank), dtype=torch.float64)
    out = torch.zeros((world, REC), dtype=torch.float64)
    sharding.gather_records(rec, out, None)                      # a healthy exchange first
    assert out[:, 0].tolist() == [0.0, 1.0]
    if rank == 1:
        time.sleep(8)                                            # stalls "inside the round"
        ret[rank] = "stalled"
        os._exit(0)                                              # (no orderly shutdown of the group: the peer has given up)
    t0 = time.time()
    try:
        sharding.gather_records(rec, out, None)
        ret[rank] = "no error"
    except sharding.ExchangeError as e:
        ret[rank] = ("ExchangeError", time.time() - t0, str(e))
    os._exit(0)


def test_stalled_rank_raises_within_the_deadline():
    """SURVEY section 5 'failure detection': the record exchange is bounded (ITAL_EXCHANGE_TIMEOUT_S) -- the counterpart of
    multiprocessing.Pool raising in the parent when a worker dies (reference ital/ital.py:124-126)."""
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        ctx = mp.spawn(_stall_worker, args=(2, port, ret), nprocs=2, join=False)
        import time
        deadline = time.time() + 60
        while time.time() < deadline and not all(not p.is_alive() for p in ctx.processes):
            time.sleep(0.2)
        for p in ctx.processes:
            if p.is_alive():
                p.kill()
        got = dict(ret)
    assert got.get(0) is not None and got[0][0] == "ExchangeError", got
    assert got[0][1] < 6.0, got            # within the 2 s deadline (plus slack), not gloo's 30 minutes
    assert "rank 0 of 2" in got[0][2] and "all-gather" in got[0][2]


def test_exchange_timeout_setting(monkeypatch):
    monkeypatch.setenv("ITAL_EXCHANGE_TIMEOUT_S", "7.5")
    assert sharding.exchange_timeout_s() == 7.5
    monkeypatch.setenv("ITAL_EXCHANGE_TIMEOUT_S", "nonsense")
    assert sharding.exchange_timeout_s() == 120.0
    monkeypatch.delenv("ITAL_EXCHANGE_TIMEOUT_S")
    assert sharding.exchange_timeout_s() == 120.0
