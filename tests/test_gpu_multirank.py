"""The sharded path end to end on the device: two ranks (both on cuda:0 -- the GPU box has one GPU -- exchanging
their records over gloo, host-staged) must reproduce the single-rank picks and the reference's golden picks.
The RCCL transport itself is the driver's multi-GPU bench; everything else of the N > 1 path runs here."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402  (fixture table only)
import make_golden_baselines  # noqa: E402  (fixture table only)

FIXTURES = dict(make_golden.FIXTURES, **make_golden_baselines.FIXTURES)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ital_amd import ITAL, MCMI_min, mvn_stream
        from ital_amd.baselines import EMOC, EntropySampling
        z = np.load(os.path.join(HERE, "golden", name + ".npz"))
        spec = FIXTURES[name]
        cls = {"ITAL": ITAL, "MCMI_min": MCMI_min, "EMOC": EMOC, "EntropySampling": EntropySampling}[spec["learner"]]
        np.random.seed(0)
        mvn_stream.GLOBAL.reset()
        L = cls(z["X"], length_scale=float(z["length_scale"]), device="cuda:0", rank=rank, world=world,
                group=dist.group.WORLD, **spec["kw"])
        L.update({int(z["query"]): 1})
        rel = z["rel"]
        picks = []
        for r in range(int(z["rounds"])):
            ret = L.fetch_unlabelled(int(z["k"]))
            picks.append(ret)
            L.update({int(i): float(rel[i]) for i in ret})
        out[rank] = (picks, np.asarray(L.rel_mean).copy(), (L.gp.row0, L.gp.row1))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["usps500", "synth96_k6", "synth300_mcmi", "usps500_mcmi", "emoc_synth150",
                                  "entropy_synth80", "synth80_mcrel", "synth50_mcboth"])
def test_two_ranks_match_golden(name):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, name, out), nprocs=world, join=True)
        r0, r1 = out[0], out[1]
    want = [z[f"r{r}_ret"].tolist() for r in range(int(z["rounds"]))]
    assert r0[0] == want and r1[0] == want                          # both ranks return the reference's picks
    np.testing.assert_allclose(r0[1], z["final_rel_mean"], rtol=0, atol=1e-9)
    np.testing.assert_array_equal(r0[1], r1[1])
    assert r0[2][0] == 0 and r0[2][1] == r1[2][0] and r1[2][1] == len(z["X"])   # each rank held half of the rows


def _nccl_worker(rank, world, port, name, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["ITAL_FORCE_COLLECTIVES"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        from ital_amd import ITAL, MCMI_min, mvn_stream
        from ital_amd.baselines import EMOC, EntropySampling
        z = np.load(os.path.join(HERE, "golden", name + ".npz"))
        spec = FIXTURES[name]
        cls = {"ITAL": ITAL, "MCMI_min": MCMI_min, "EMOC": EMOC, "EntropySampling": EntropySampling}[spec["learner"]]
        np.random.seed(0)
        mvn_stream.GLOBAL.reset()
        L = cls(z["X"], length_scale=float(z["length_scale"]), device="cuda:0", rank=rank, world=world,
                group=dist.group.WORLD, **spec["kw"])
        assert L.gp.collective
        L.update({int(z["query"]): 1})
        rel = z["rel"]
        picks = []
        for r in range(int(z["rounds"])):
            ret = L.fetch_unlabelled(int(z["k"]))
            picks.append(ret)
            L.update({int(i): float(rel[i]) for i in ret})
        out[rank] = (picks, np.asarray(L.rel_mean).copy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["usps500", "synth300_mcmi"])
def test_rccl_code_path_on_a_one_rank_group(name):
    """The collectives of the sharded path (record all-gather, row all-reduce, vector all-gather) through RCCL itself --
    on a one-rank "nccl" process group, which is all a one-GPU box can host."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_nccl_worker, args=(1, port, name, out), nprocs=1, join=True)
        r0 = out[0]
    assert r0[0] == [z[f"r{r}_ret"].tolist() for r in range(int(z["rounds"]))]
    np.testing.assert_allclose(r0[1], z["final_rel_mean"], rtol=0, atol=1e-9)
