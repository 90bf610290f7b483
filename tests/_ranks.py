"""Where the ranks of a multi-process GPU test live (shared by test_gpu_multirank*.py and test_gpu_multidevice.py).

mode "gloo":  every rank on cuda:0, records host-staged over gloo -- all a one-GPU box can host (the sharded path end to
              end except the transport).
mode "rccl":  rank r on cuda:r, torch.distributed backend "nccl" (= RCCL over xGMI): the placement of a real multi-GPU run
              (bench.py under torch.distributed.run).  Needs world <= torch.cuda.device_count().
mode "rccl1": a one-rank "nccl" group on cuda:0 with the collectives forced on (ITAL_FORCE_COLLECTIVES).
The worker bodies are the same for all modes: what passes over gloo on one GPU is what runs over RCCL on several."""
import os
import socket


class RanksUnavailable(RuntimeError):
    """The process group of a placement could not be set up on this machine (no RCCL between these devices, a refused
    port ...): an infrastructure condition -- the test is skipped, not failed."""


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def device_count():
    """GPUs visible, without initialising the runtime in the test runner's own process."""
    import torch
    return torch.cuda.device_count()


def join(rank, world, port, mode):
    """Process-group set-up of rank `rank`; returns (device string, group or None)."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if mode is None:
        assert world == 1
        return "cuda:0", None
    try:
        if mode == "gloo":
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", rank=rank, world_size=world)
            return "cuda:0", dist.group.WORLD
        if mode == "rccl1":
            assert world == 1
            os.environ["ITAL_FORCE_COLLECTIVES"] = "1"
            torch.cuda.set_device(0)
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
            return "cuda:0", dist.group.WORLD
        if mode == "rccl":
            if world > torch.cuda.device_count():
                raise RanksUnavailable("%d devices visible, %d ranks" % (torch.cuda.device_count(), world))
            torch.cuda.set_device(rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
            # the first collective connects the devices: a machine whose GPUs cannot reach each other fails here
            probe = torch.ones(1, device="cuda:%d" % rank)
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise RanksUnavailable("all-reduce over %d devices returned %r" % (world, probe.item()))
            return "cuda:%d" % rank, dist.group.WORLD
    except RanksUnavailable:
        raise
    except Exception as e:      # noqa: BLE001 -- set-up of the process group, not the code under test
        raise RanksUnavailable("%s process group of %d ranks: %s: %s" % (mode, world, type(e).__name__, e)) from e
    raise ValueError(mode)


def leave(group):
    import torch.distributed as dist
    if group is not None:
        dist.destroy_process_group()


def _guarded(rank, world, port, worker, args, out):
    try:
        worker(rank, world, port, *args, out)
    except RanksUnavailable as e:
        out[rank] = ("__unavailable__", str(e))


SPAWN_TIMEOUT_S = 1500       # a sharded run that has not finished by then hangs (mismatched collectives): fail, do not block the suite


def spawn(worker, world, *args):
    """Runs worker(rank, world, port, *args, out) on `world` processes; returns [out[0], ..., out[world - 1]].  Skips the
    calling test when the placement's process group cannot be set up here; fails it (and ends the ranks) on a hang."""
    import time
    import pytest
    import torch.multiprocessing as mp
    port = free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        ctx = mp.spawn(_guarded, args=(world, port, worker, tuple(args), out), nprocs=world, join=False)
        deadline = time.time() + SPAWN_TIMEOUT_S
        while not ctx.join(timeout=5):
            if time.time() > deadline:
                for proc in ctx.processes:
                    if proc.is_alive():
                        proc.kill()
                pytest.fail("%d ranks did not finish within %d s (a collective that not every rank entered?)" % (world, SPAWN_TIMEOUT_S))
        res = [out.get(r) for r in range(world)]
    for r in res:
        if isinstance(r, tuple) and len(r) == 2 and r[0] == "__unavailable__":
            pytest.skip("process group not available on this machine: " + r[1])
    assert all(r is not None for r in res), "a rank ended without a result"
    return res
