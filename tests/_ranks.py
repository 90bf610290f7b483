"""Where the ranks of a multi-process GPU test live (shared by test_gpu_multirank*.py and test_gpu_multidevice.py).

mode "gloo":  every rank on cuda:0, records host-staged over gloo -- all a one-GPU box can host (the sharded path end to
              end except the transport).
mode "rccl":  rank r on cuda:r, torch.distributed backend "nccl" (= RCCL over xGMI): the placement of a real multi-GPU run
              (bench.py under torch.distributed.run).  Needs world <= torch.cuda.device_count().
mode "rccl1": a one-rank "nccl" group on cuda:0 with the collectives forced on (ITAL_FORCE_COLLECTIVES).
The worker bodies are the same for all modes: what passes over gloo on one GPU is what runs over RCCL on several."""
import os
import socket


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
        assert world <= torch.cuda.device_count(), "one device per rank"
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
        return "cuda:%d" % rank, dist.group.WORLD
    raise ValueError(mode)


def leave(group):
    import torch.distributed as dist
    if group is not None:
        dist.destroy_process_group()


def spawn(worker, world, *args):
    """Runs worker(rank, world, port, *args, out) on `world` processes; returns [out[0], ..., out[world - 1]]."""
    import torch.multiprocessing as mp
    port = free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(worker, args=(world, port) + tuple(args) + (out,), nprocs=world, join=True)
        return [out[r] for r in range(world)]
