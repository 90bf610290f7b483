"""The sharded path end to end on the device: two ranks (both on cuda:0 -- the GPU box has one GPU -- exchanging
their records over gloo, host-staged) must reproduce the single-rank picks and the reference's golden picks.
The RCCL transport itself is the driver's multi-GPU bench; everything else of the N > 1 path runs here."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
import _ranks  # noqa: E402
import make_golden  # noqa: E402  (fixture table only)
import make_golden_baselines  # noqa: E402  (fixture table only)

FIXTURES = dict(make_golden.FIXTURES, **make_golden_baselines.FIXTURES)


def _worker(rank, world, port, name, mode, out):
    """A golden fixture's session on `world` ranks (placement: _ranks.join).  Everything the assertions need goes to out[rank]."""
    dev, group = _ranks.join(rank, world, port, mode)
    try:
        from ital_amd import ITAL, MCMI_min, mvn_stream
        from ital_amd.baselines import EMOC, EntropySampling
        z = np.load(os.path.join(HERE, "golden", name + ".npz"))
        spec = FIXTURES[name]
        cls = {"ITAL": ITAL, "MCMI_min": MCMI_min, "EMOC": EMOC, "EntropySampling": EntropySampling}[spec["learner"]]
        np.random.seed(0)
        mvn_stream.GLOBAL.reset()
        L = cls(z["X"], length_scale=float(z["length_scale"]), device=dev, rank=rank, world=world, group=group, **spec["kw"])
        assert L.gp.collective
        L.update({int(z["query"]): 1})
        rel = z["rel"]
        picks = []
        for r in range(int(z["rounds"])):
            ret = L.fetch_unlabelled(int(z["k"]))
            picks.append(ret)
            L.update({int(i): float(rel[i]) for i in ret})
        top = None
        if rank == 0:
            # reading the means / the ranking on ONE rank only must not need the others (they are replicated by update())
            top = (np.asarray(L.top_results(10)).tolist(), np.asarray(L.rel_mean).copy())
        dist.barrier()
        upd = L.updated_prediction({int(rel.argmax()): 1, int(rel.argmin()): -1}, [0, len(rel) // 2, len(rel) - 1])
        transport = L._round_transport() if hasattr(L, "_round_transport") else None
        out[rank] = (picks, np.asarray(L.rel_mean).copy(), (L.gp.row0, L.gp.row1), top, upd, transport and transport[0],
                     (mvn_stream.GLOBAL.draws, tuple(mvn_stream.GLOBAL.state)))
    finally:
        _ranks.leave(group)


def check_golden(name, res):
    """What every rank of a sharded golden session must have produced (`res`: out[0 .. world - 1] of _worker)."""
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    world = len(res)
    want = [z[f"r{r}_ret"].tolist() for r in range(int(z["rounds"]))]
    for r in res:
        assert r[0] == want                                         # every rank returns the reference's picks
        np.testing.assert_array_equal(r[1], res[0][1])
        assert r[6] == res[0][6]                                    # the replayed mvndst stream stands at the same place
    np.testing.assert_allclose(res[0][1], z["final_rel_mean"], rtol=0, atol=1e-9)
    n = len(z["X"])
    assert res[0][2][0] == 0 and res[-1][2][1] == n                 # the ranks' row blocks tile the data
    assert all(res[w][2][1] == res[w + 1][2][0] for w in range(world - 1))
    top10, mean0 = res[0][3]
    assert top10 == np.argsort(mean0, kind="stable")[::-1][:10].tolist()
    if "top_results_10" in z:
        assert top10 == z["top_results_10"].tolist()
    # full covariance blocks across the shards: all ranks hold the same simulated update (reference retrieval_base.py:129-164)
    for r in res[1:]:
        np.testing.assert_array_equal(res[0][4][0], r[4][0])
        np.testing.assert_array_equal(res[0][4][1], r[4][1])
    assert res[0][4][1].shape == (3, 3) and np.all(np.isfinite(res[0][4][1]))


GOLDEN_SHARDED = ["usps500", "synth96_k6", "synth300_mcmi", "usps500_mcmi", "emoc_synth150", "entropy_synth80", "synth80_mcrel",
                  "synth50_mcboth", "synth200_topcand", "synth200_topcand_float", "synth200_noisy", "synth50_clip"]


@pytest.mark.parametrize("name", GOLDEN_SHARDED)
def test_two_ranks_match_golden(name):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    check_golden(name, _ranks.spawn(_worker, 2, name, "gloo"))


def _dup_worker(rank, world, port, X, mode, out):
    dev, group = _ranks.join(rank, world, port, mode)
    try:
        from ital_amd import ITAL, mvn_stream
        mvn_stream.GLOBAL.reset()
        L = ITAL(X, length_scale=0.9, device=dev, rank=rank, world=world, group=group)
        L.keep_scores = True
        L.update({58: 1, 59: -1})
        picks = [L.fetch_unlabelled(4)]
        L.update({i: 1.0 if X[i, 0] > 0.5 else -1.0 for i in picks[0]})
        picks.append(L.fetch_unlabelled(3))
        out[rank] = (picks, mvn_stream.GLOBAL.draws, int(L.gp.status.item()))
    finally:
        _ranks.leave(group)


def duplicate_rows_case():
    X = duplicate_rows_case()
    one = _ranks.spawn(_dup_worker, 1, X, None)[0]
    two = _ranks.spawn(_dup_worker, 2, X, "gloo")
    assert two[0][0] == two[1][0] == one[0]
    assert two[0][1] == two[1][1] == one[1]
    assert two[0][2] == 0 and two[1][2] == 0      # the fall-back bits were cleared on both ranks


@pytest.mark.parametrize("name", ["usps500", "synth300_mcmi"])
def test_rccl_code_path_on_a_one_rank_group(name):
    """The collectives of the sharded path (record all-gather, row all-reduce, vector all-gather) through RCCL itself --
    on a one-rank "nccl" process group, which is all a one-GPU box can host (several devices: test_gpu_multidevice.py)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    res = _ranks.spawn(_worker, 1, name, "rccl1")
    check_golden(name, res)
    if name == "usps500":
        assert res[0][5] == "nccl"           # the rounds ran as single calls with ncclAllGather issued from C


def _round_worker(rank, world, port, mode, round_call, X, k, rounds, out):
    dev, group = _ranks.join(rank, world, port, mode)
    try:
        from ital_amd import ITAL, mvn_stream
        mvn_stream.GLOBAL.reset()
        L = ITAL(X, length_scale=float(np.sqrt(X.shape[1] / 12.0)), device=dev, rank=rank, world=world, group=group)
        L.round_call = round_call
        L.update({0: 1, len(X) - 1: -1})
        picks, how, scores = [], [], []
        for r in range(rounds):
            L.keep_scores = r == rounds - 1            # the last round also hands back its score vectors
            L.last_round = None
            ret = L.fetch_unlabelled(k)
            picks.append(ret)
            how.append(L.last_round)
            L.update({int(i): (1.0 if X[i, 0] > 0.5 else -1.0) for i in ret})
        scores = [s_.cpu().numpy().copy() for s_ in L.last_scores]
        transport = L._round_transport() if L.gp.collective else None
        out[rank] = (picks, how, scores, (L.gp.row0, L.gp.row1), transport and transport[0], mvn_stream.GLOBAL.draws,
                     np.asarray(L.rel_mean).copy())
    finally:
        _ranks.leave(group)


def _run_round_workers(world, mode, round_call, X, k, rounds):
    return dict(enumerate(_ranks.spawn(_round_worker, world, mode, round_call, X, k, rounds)))


def check_rounds_against_one_rank(one, ranks_round, ranks_steps, k, rounds, transport):
    """`ranks_round` / `ranks_steps`: the sharded run through ital_fetch_round / step by step (dicts rank -> result of
    _round_worker); `one`: the one-rank run.  Picks, stream position, means, score vectors."""
    world = len(ranks_round)
    for res in (ranks_round, ranks_steps):
        assert all(res[r][0] == one[0] for r in range(world))
        assert all(res[r][5] == one[5] for r in range(world))
        np.testing.assert_allclose(res[0][6], one[6], rtol=0, atol=1e-12)
    for rank in range(world):
        assert ranks_round[rank][4] == transport
        # how the candidate list reached the device: uploaded once, then compacted there (speculative descriptor: slot flips)
        assert [h[0] for h in ranks_round[rank][1]] == [1] + [2] * (rounds - 1)
        assert all(h is None for h in ranks_steps[rank][1])
        for a, b_ in zip(ranks_round[rank][2], ranks_steps[rank][2]):
            live = a != 0                 # (members picked earlier in the round: zero here, the stale score there)
            assert live.sum() >= len(a) - k
            np.testing.assert_array_equal(a[live], b_[: len(a)][live])
    # the shares' score vectors are the one-rank vector cut at the share boundaries (last round; dead entries excepted)
    lo_hi = [ranks_round[r][3] for r in range(world)]
    assert all(lo_hi[r][1] == lo_hi[r + 1][0] for r in range(world - 1))
    for t in range(k):
        whole = np.concatenate([ranks_round[r][2][t] for r in range(world)])
        assert whole.shape == one[2][t].shape
        np.testing.assert_allclose(whole, one[2][t], rtol=1e-12, atol=0)


@pytest.mark.parametrize("n,d,k", [(700, 16, 4), (90, 6, 5)])
def test_round_as_one_call_on_two_ranks(n, d, k):
    """ital_fetch_round with several ranks: every rank enqueues its whole round -- list upkeep on its own share, scoring
    launches ending with the rank's record, exchange, resolve, covariance column -- in one call; the exchange is the host's
    callback here (gloo moves host memory; RCCL's ncclAllGather sits at the same place, next test).  Picks, stream
    position, score vectors and means equal the step-by-step path's and the one-rank run's over five rounds -- first
    round from an uploaded list, the following ones compacted on the device out of the previous share, with the
    descriptor prepared before the picks were known."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    X = np.random.default_rng(77).random((n, d))
    rounds = 5
    one = _run_round_workers(1, None, True, X, k, rounds)[0]
    two_round = _run_round_workers(2, "gloo", True, X, k, rounds)
    two_steps = _run_round_workers(2, "gloo", False, X, k, rounds)
    check_rounds_against_one_rank(one, two_round, two_steps, k, rounds, "host")


def test_round_as_one_call_through_rccl():
    """The same call with the transport of a real multi-GPU run: ncclAllGather on the process group's own communicator
    (ProcessGroupNCCL._comm_ptr) from inside ital_fetch_round -- on a one-rank group, all a one-GPU box can host."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    X = np.random.default_rng(78).random((600, 12))
    one = _run_round_workers(1, None, True, X, 4, 4)[0]
    res = _run_round_workers(1, "rccl1", True, X, 4, 4)[0]
    assert res[4] == "nccl"
    assert [h[0] for h in res[1]] == [1, 2, 2, 2]
    assert res[0] == one[0] and res[5] == one[5]
    for a, b_ in zip(res[2], one[2]):
        np.testing.assert_array_equal(a, b_)
    np.testing.assert_allclose(res[6], one[6], rtol=0, atol=1e-12)


def test_record_exchange_below_the_c_abi_through_rccl():
    """ital_select_exchange: the per-step all-gather as a non-Python host would drive it -- a raw ncclComm_t (here a
    one-rank communicator created through RCCL's own C API) and a HIP stream; no torch.distributed involved."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ctypes
    from ital_amd import _lib
    lib = _lib.lib()
    # torch's own RCCL, opened the way a Python host opens it: RTLD_LOCAL -- its symbols are NOT in the global scope, the
    # library has to find the loaded object itself (and must not bring a second RCCL into the process)
    rccl = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

    def rccl_objects():
        with open("/proc/self/maps") as f:
            return {line.split()[-1] for line in f if "librccl" in line}
    before = rccl_objects()

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    torch.cuda.set_device(0)
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        rec_len = _lib.ITAL_REC_HEADER + 32 + 64 + 4
        rec = torch.arange(rec_len, dtype=torch.float64, device="cuda:0") * 0.5
        out = torch.zeros((1, rec_len), dtype=torch.float64, device="cuda:0")
        _lib.check(lib.ital_select_exchange(rec.data_ptr(), out.data_ptr(), rec_len, comm,
                                            torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert torch.equal(out[0], rec)
        # what RCCL says about the communicator (sharding.raw_comm compares it with the process group's numbering)
        w, r = ctypes.c_int(-1), ctypes.c_int(-1)
        how = ctypes.create_string_buffer(600)
        _lib.check(lib.ital_exchange_info(comm, ctypes.byref(w), ctypes.byref(r), how, 600))
        assert (w.value, r.value) == (1, 0) and b"librccl" in how.value
        assert rccl_objects() == before and len(before) == 1      # still the one RCCL the communicator came from
        assert lib.ital_select_exchange(rec.data_ptr(), out.data_ptr(), rec_len, None,
                                        torch.cuda.current_stream().cuda_stream) != 0     # no communicator: refused
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)
