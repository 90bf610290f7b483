"""The sharded path over RCCL with ONE DEVICE PER RANK -- what bench.py runs under torch.distributed.run on a multi-GPU
node.  Every test here skips on a box with fewer GPUs than ranks (the pool's test boxes have one); on a node with 2 or
more, `pytest -m gpu` proves by itself that the ranks' picks, scores and stream positions over ncclAllGather / xGMI are the
one-rank run's and the reference's golden ones.  The worker bodies are those of test_gpu_multirank*.py (two ranks on one
GPU over gloo, which runs everywhere): only the placement differs (_ranks.join, mode "rccl").

Reference path under test: ital/ital.py:124-130 (Pool.map over the candidates + np.argmax -> row shards + one record
all-gather per greedy step)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ranks  # noqa: E402
import test_gpu_multirank as mr  # noqa: E402
import test_gpu_multirank_scale as mrs  # noqa: E402

NDEV = _ranks.device_count()
# (the pool's one-GPU boxes allow few processes on a card at once; here every rank has a card of its own)
def _worlds(sizes):
    return [pytest.param(w, marks=pytest.mark.skipif(NDEV < w, reason="%d GPUs visible, %d needed (one device per rank)" % (NDEV, w)))
            for w in sizes]


WORLDS = _worlds((2, 4))
# the communicator size of the scaling run (bench.py --gpus 8): one cheap golden session and the scaling workload itself --
# one process per device (each card hosts one rank; the runner itself never initialises a GPU)
WORLDS_8 = _worlds((2, 4, 8))


@pytest.mark.skipif(NDEV < 8, reason="%d GPUs visible, 8 needed (one device per rank)" % NDEV)
def test_eight_ranks_on_their_own_devices_match_golden():
    """The golden USPS session with the rows over EIGHT devices: the world size of the scaling run."""
    res = _ranks.spawn(mr._worker, 8, "usps500", "rccl")
    mr.check_golden("usps500", res)
    assert all(r[5] == "nccl" for r in res)


@pytest.mark.parametrize("world", WORLDS)
@pytest.mark.parametrize("name", ["usps500", "synth96_k6", "synth300_mcmi", "synth200_topcand", "synth200_topcand_float",
                                  "synth200_noisy", "synth80_mcrel", "emoc_synth150"])
def test_ranks_on_their_own_devices_match_golden(name, world):
    """Golden sessions of the reference (picks of every round, final means, top_results) with the rows sharded over
    `world` devices: perfect user through ital_fetch_round with ncclAllGather issued from C, batches of 6, MCMI_min, the
    top_candidates restriction (explicit list positions), a noisy user and sampled patterns through the general scorer."""
    res = _ranks.spawn(mr._worker, world, name, "rccl")
    mr.check_golden(name, res)
    if name in ("usps500", "synth96_k6"):
        assert all(r[5] == "nccl" for r in res)        # the raw communicator of the process group, agreed by all ranks


@pytest.mark.parametrize("world", WORLDS)
def test_round_as_one_call_over_rccl(world):
    """ital_fetch_round on `world` devices against the step-by-step path and the one-rank run: picks, stream position,
    score vectors, means; candidate list uploaded once, then compacted on each device out of its previous share."""
    n, d, k, rounds = 700, 16, 4, 5
    X = np.random.default_rng(77).random((n, d))
    one = mr._run_round_workers(1, None, True, X, k, rounds)[0]
    by_round = mr._run_round_workers(world, "rccl", True, X, k, rounds)
    by_steps = mr._run_round_workers(world, "rccl", False, X, k, rounds)
    mr.check_rounds_against_one_rank(one, by_round, by_steps, k, rounds, "nccl")


@pytest.mark.parametrize("world", WORLDS)
def test_duplicate_rows_on_different_devices_fall_back_on_every_rank(world):
    """The status word of the rank that meets a singular batch travels in the records: all ranks redo the round through
    the general scorer (and ITS collectives, torch.distributed or raw) -- same picks and stream position as one rank."""
    X = mr.duplicate_rows_case()
    one = _ranks.spawn(mr._dup_worker, 1, X, None)[0]
    res = _ranks.spawn(mr._dup_worker, world, X, "rccl")
    assert all(r[0] == one[0] and r[1] == one[1] and r[2] == 0 for r in res)


def _env_worker(rank, world, port, X, out):
    if rank == 1:
        os.environ["ITAL_RAW_COMM"] = "0"      # ONE rank declines the raw communicator: all must stay on torch.distributed
    dev, group = _ranks.join(rank, world, port, "rccl")
    try:
        from ital_amd import ITAL, mvn_stream, sharding
        mvn_stream.GLOBAL.reset()
        L = ITAL(X, length_scale=float(np.sqrt(X.shape[1] / 12.0)), device=dev, rank=rank, world=world, group=group)
        L.update({0: 1, len(X) - 1: -1})
        picks = [L.fetch_unlabelled(4) for _ in range(1)]
        out[rank] = (picks, L._round_transport(), sharding.raw_comm_reason(group, dev))
    finally:
        _ranks.leave(group)


@pytest.mark.skipif(NDEV < 2, reason="needs 2 GPUs")
def test_one_rank_declining_the_raw_communicator_keeps_all_on_torch_distributed():
    X = np.random.default_rng(78).random((600, 12))
    one = mr._run_round_workers(1, None, True, X, 4, 1)[0]
    res = _ranks.spawn(_env_worker, 2, X)
    assert res[0][0] == res[1][0] == one[0]
    assert res[0][1] is None and res[1][1] is None            # no transport below the C ABI: the step path on both
    assert "rank 1" in res[1][2] and "another rank" in res[0][2]


@pytest.mark.parametrize("world", WORLDS_8)
def test_one_million_x512_k4_over_rccl(world):
    """The scaling curve's workload (bench.py `scaling_workload`) on `world` devices against the one-rank run: identical
    picks, sampled MI to 1e-12, both random streams at the same position."""
    n = 1_000_000
    cfg = dict(n=n, d=512, k=4, rounds=2, kw={})
    one = mrs._run(1, cfg)
    many = mrs._run(world, cfg, "rccl")
    assert mrs._compare(one, many, n) >= 300
    assert all(r["transport"] == "nccl" for r in many)
    print("1M x 512, k = 4: fetch %.3f s on one device, %.3f s on %d devices"
          % (one[0]["secs"][-1], max(r["secs"][-1] for r in many), world))


@pytest.mark.skipif(NDEV < 2, reason="needs 2 GPUs")
def test_monte_carlo_k16_over_rccl():
    """BASELINE config 5's switch (monte_carlo_num_rel = 1, k = 16) at 100 000 x 64 on two devices."""
    n = 100_000
    cfg = dict(n=n, d=64, k=16, rounds=1, kw=dict(monte_carlo_num_rel=1))
    one = mrs._run(1, cfg)
    two = mrs._run(2, cfg, "rccl")
    assert mrs._compare(one, two, n) >= 100


def _ctx_worker(rank, world, port, golden, out):
    """The context-style layer of the C ABI (ital_ctx_*, csrc/ctx.hip) on `world` devices: rows sharded, the query's row
    replicated through the exchange, one ncclAllGather of a record per greedy step -- two golden rounds."""
    import ctypes
    dev, group = _ranks.join(rank, world, port, "rccl")
    try:
        from ital_amd import _lib, sharding
        lib, chk = _lib.load(), _lib.check
        comm = sharding.raw_comm(group, dev)
        if comm is None:
            out[rank] = ("no raw communicator", sharding.raw_comm_reason(group, dev))
            return
        z = np.load(golden)
        X = np.ascontiguousarray(z["X"], dtype=np.float64)
        n, d = X.shape
        k = int(z["k"])
        ctx = ctypes.c_void_p()
        chk(lib.ital_ctx_create(n, d, float(z["length_scale"]), float(z["var"]), float(z["noise"]), 64, rank, world, comm,
                                ctypes.byref(ctx)))
        row0 = ctypes.c_int64(-1)
        n_loc = lib.ital_ctx_local_rows(ctx, ctypes.byref(row0))
        mine = np.ascontiguousarray(X[row0.value:row0.value + n_loc])
        chk(lib.ital_ctx_fit(ctx, mine.ctypes.data, 0, None))
        picks_all, means = [], []
        picks = np.zeros(8, dtype=np.int64)
        for r in range(int(z["rounds"])):
            ind = np.ascontiguousarray(z["r%d_ind" % r], dtype=np.int64)
            y = np.ascontiguousarray(z["r%d_y" % r], dtype=np.float64)
            new = slice(0, len(ind)) if r == 0 else slice(len(z["r%d_ind" % (r - 1)]), len(ind))
            chk(lib.ital_ctx_update(ctx, ind[new].ctypes.data, y[new].ctypes.data, len(ind[new]), None))
            mean = np.empty(n_loc)
            chk(lib.ital_ctx_predict_stored(ctx, mean.ctypes.data, None, None))
            means.append((row0.value, mean))
            assert lib.ital_ctx_fetch(ctx, k, picks.ctypes.data, None) == k, lib.ital_last_error()
            picks_all.append(picks[:k].tolist())
        chk(lib.ital_ctx_destroy(ctx))
        out[rank] = ("ok", picks_all, means)
    finally:
        _ranks.leave(group)


@pytest.mark.parametrize("world", WORLDS)
def test_context_api_on_several_devices(world, golden_dir):
    golden = os.path.join(golden_dir, "usps500.npz")
    z = np.load(golden)
    res = _ranks.spawn(_ctx_worker, world, golden)
    for r in res:
        assert r[0] == "ok", r
        assert r[1] == [z["r%d_ret" % q].tolist() for q in range(int(z["rounds"]))]
        for q, (row0, mean) in enumerate(r[2]):
            np.testing.assert_allclose(mean, z["r%d_rel_mean" % q][row0:row0 + len(mean)], rtol=0, atol=2e-9)
