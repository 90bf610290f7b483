"""The sharded path at the sizes BASELINE.json names, two ranks against one: both ranks on cuda:0 (the GPU box has one GPU),
records exchanged over gloo.  What a first real 8-GPU run could trip over is covered here: 64-bit row / position offsets,
slabs of the lattice workspace on a shard that does not start at position 0, the stream position of the replay, and -- with
the Monte-Carlo pattern switch (BASELINE config 5: k = 16) -- a host side whose cost scales with the rank's OWN
candidates (the other ranks' standard normals are skipped, not computed: ital_np_legacy_normals).

Reference path under test: ital/ital.py:124-130 (Pool.map over candidates -> row shards), :293-297 (pattern sampling)."""
import os
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import _ranks  # noqa: E402


def _rows(row0, row1, d, seed, block=65536):
    """Rows [row0, row1) of the synthetic matrix whose 65536-row block b is default_rng([seed, b]).random() (as bench.py):
    every rank generates exactly its own rows."""
    out = np.empty((row1 - row0, d))
    b = row0 // block
    while b * block < row1:
        lo, hi = max(row0, b * block), min(row1, (b + 1) * block)
        blk = np.random.default_rng([seed, b]).random((block, d))
        out[lo - row0:hi - row0] = blk[lo - b * block:hi - b * block]
        b += 1
    return out


def _label(i):
    return 1.0 if (i * 2654435761) % (1 << 32) < (1 << 31) else -1.0


def _worker(rank, world, port, cfg, mode, out):
    dev, group = _ranks.join(rank, world, port, mode)
    try:
        from ital_amd import ITAL, mvn_stream, sharding
        n, d, k, rounds = cfg["n"], cfg["d"], cfg["k"], cfg["rounds"]
        row0, row1 = sharding.row_range(n, world, rank)
        data = sharding.ShardedRows(_rows(row0, row1, d, seed=11), n, row0)
        np.random.seed(7)
        mvn_stream.GLOBAL.reset()
        L = ITAL(data, length_scale=float(np.sqrt(d / 12.0)), device=dev, rank=rank, world=world, group=group, **cfg["kw"])
        L.keep_scores = True
        L.update({3: 1, n - 5: -1})                 # one labelled sample on each shard
        sample = np.random.default_rng(5).choice(n, 400, replace=False)
        picks, scores, secs = [], [], []
        for _ in range(rounds):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ret = L.fetch_unlabelled(k)
            torch.cuda.synchronize()
            secs.append(time.perf_counter() - t0)
            picks.append(ret)
            # MI of the sampled data rows this rank scored, per greedy step (list position = data index minus the labelled
            # samples in front: the candidate list is the ascending get_unseen() order)
            cand = np.asarray(L._unseen_array())
            mine = cand[(cand >= row0) & (cand < row1)]
            sel = np.flatnonzero(np.isin(mine, sample))
            scores.append({int(mine[j]): [float(s[j].item()) for s in L.last_scores] for j in sel})
            L.update({int(i): _label(int(i)) for i in ret})
        out[rank] = dict(picks=picks, scores=scores, draws=mvn_stream.GLOBAL.draws, state=tuple(mvn_stream.GLOBAL.state),
                         secs=secs, mc_walk=list(L.mc_walk), np_tail=np.random.random_sample(3).tolist(),
                         rows=(row0, row1), transport=(L._round_transport() or (None,))[0] if world > 1 else None)
    finally:
        _ranks.leave(group)


def _run(world, cfg, mode="gloo"):
    return _ranks.spawn(_worker, world, cfg, mode if world > 1 else None)


def _compare(one, many, n):
    """`many`: the results of the ranks of a sharded run; `one`: the one-rank run of the same session."""
    a, world = one[0], len(many)
    for r in many:
        assert r["picks"] == a["picks"]                                       # same batches on every rank as on one
        assert r["draws"] == a["draws"] and r["state"] == a["state"]          # the replayed mvndst stream stands where it would
        assert r["np_tail"] == a["np_tail"]                                   # numpy's global generator too
    assert [r["rows"] for r in many] == [(n * w // world, n * (w + 1) // world) for w in range(world)]
    checked = 0
    for rnd, ref in enumerate(a["scores"]):
        for part in (r["scores"][rnd] for r in many):
            for idx, vals in part.items():
                if idx in ref:
                    np.testing.assert_allclose(vals, ref[idx], rtol=1e-12, atol=0)    # MI does not depend on the sharding
                    checked += 1
    return checked


def test_two_ranks_one_million_x512_k4():
    """The workload of the scaling curve (bench.py `scaling_workload`): 1 000 000 x 512, k = 4, full enumeration; rows split
    over two ranks (500 000 each: the second shard starts at list position ~500 000, the t = 3, 4 steps run in slabs)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    n = 1_000_000
    cfg = dict(n=n, d=512, k=4, rounds=2, kw={})
    one = _run(1, cfg)
    two = _run(2, cfg)
    checked = _compare(one, two, n)
    assert checked >= 300
    want = sum((n - 2 - 4 * r - t) * (2 << (t + 1)) * (0 if t + 1 < 3 else 8 * (2 * t - 1)) for r in range(2) for t in range(4))
    assert one[0]["draws"] == want
    print("1M x 512, k = 4: fetch %.3f s on one rank, %.3f s per rank with two ranks sharing the GPU"
          % (one[0]["secs"][-1], max(two[0]["secs"][-1], two[1]["secs"][-1])))


def test_two_ranks_monte_carlo_k16_host_work_is_rank_local():
    """BASELINE config 5's switch at a size one GPU handles quickly: 100 000 x 64, k = 16, monte_carlo_num_rel = 1.  Picks,
    sampled MI and both random streams equal the one-rank run; each rank COMPUTES the standard normals of its own
    candidates only (half of them) and skips the rest."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    n = 100_000
    cfg = dict(n=n, d=64, k=16, rounds=1, kw=dict(monte_carlo_num_rel=1))
    one = _run(1, cfg)
    two = _run(2, cfg)
    checked = _compare(one, two, n)
    assert checked >= 100
    total = one[0]["mc_walk"][0]
    assert one[0]["mc_walk"][1] == 0 and total > 1.3e8                     # sum_t (N - t + 1) t^2 standard normals, t = 1..16
    for r in two:
        made, skipped, sec = r["mc_walk"]
        assert made + skipped == total
        assert abs(made - total / 2) < 0.02 * total                          # its own half (+ t candidates of slack per step)
    print("k = 16 Monte-Carlo patterns, 100 000 rows: %d normals per round; host walk %.2f s on one rank, %.2f / %.2f s on two; "
          "fetch %.2f s / %.2f s" % (total, one[0]["mc_walk"][2], two[0]["mc_walk"][2], two[1]["mc_walk"][2],
                                     one[0]["secs"][0], max(two[0]["secs"][0], two[1]["secs"][0])))
