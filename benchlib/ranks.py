"""Rank adapters: what a rank of the benchmark needs from its launcher -- broadcast / all-gather of a few bytes, barrier, max / min of
a float.  TorchRanks = one process per GPU over torch.distributed (the driver's launch); ThreadRanks = host threads of one process on
one GPU against the RCCL test double (--fake-ranks, development and tests).  The halo traffic never goes through these."""
import os

from .launcher import free_port

class TorchRanks:
    """One process per GPU (the driver's launch): torch.distributed over RCCL for rendezvous, barrier and the max over ranks.
    The halo traffic itself does not go through torch: libtetsim_hip owns its RCCL communicator."""

    def __init__(self, local_rank, rank=0, world=1):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")     # --force-dist without a launcher: a one-rank rendezvous with itself
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(free_port())
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    def broadcast_bytes(self, data, n):           # rank 0's `data` (n bytes) to everyone
        t = self.torch.zeros(n, dtype=self.torch.uint8, device="cuda")
        if data is not None:
            t.copy_(self.torch.tensor(list(data), dtype=self.torch.uint8))
        self.dist.broadcast(t, src=0)
        return bytes(t.cpu().tolist())

    def all_gather_bytes(self, data, n):          # every rank's `data` (n bytes), in rank order
        t = self.torch.tensor(list(data), dtype=self.torch.uint8, device="cuda")
        out = self.torch.empty(n * self.dist.get_world_size(), dtype=self.torch.uint8, device="cuda")
        self.dist.all_gather_into_tensor(out, t)
        flat = bytes(out.cpu().tolist())
        return [flat[i * n:(i + 1) * n] for i in range(self.dist.get_world_size())]

    def barrier(self):
        self.torch.cuda.synchronize()
        self.dist.barrier()

    def max_float(self, x):
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def min_float(self, x):
        return -self.max_float(-x)

    def close(self):
        self.dist.destroy_process_group()

class ThreadRanks:
    """--fake-ranks N (development / tests on a ONE-GPU box): the N ranks are host threads of this process, all on device 0,
    and librccl is the strict test double of tests/mock_rccl (TETSIM_RCCL_LIB).  Same code path as the real launch from
    `run()` down; only this adapter differs."""

    def __init__(self, shared, rank):
        self.s, self.rank = shared, rank

    def broadcast_bytes(self, data, n):
        if data is not None:
            self.s["bytes"] = bytes(data)
        self.s["barrier"].wait()
        out = self.s["bytes"]
        self.s["barrier"].wait()
        return out

    def all_gather_bytes(self, data, n):
        self.s.setdefault("gather", [None] * len(self.s["vals"]))[self.rank] = bytes(data)
        self.s["barrier"].wait()
        out = list(self.s["gather"])
        self.s["barrier"].wait()
        return out

    def barrier(self):
        self.s["barrier"].wait()

    def max_float(self, x):
        self.s["vals"][self.rank] = x
        self.s["barrier"].wait()
        out = max(self.s["vals"])
        self.s["barrier"].wait()
        return out

    def min_float(self, x):
        return -self.max_float(-x)

    def close(self):
        pass
