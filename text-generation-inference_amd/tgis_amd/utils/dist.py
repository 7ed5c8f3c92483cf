"""Process-group bootstrap: one process per GPU, RCCL (the `nccl` backend of PyTorch-ROCm) over xGMI.

Mirrors utils/dist.py:21-96 of the reference: a FakeGroup for world size 1, `nccl` with a high-priority
stream and a 60 s timeout on GPUs, `gloo` on CPU (used by the world_size-2 tests).  Rendezvous through
MASTER_ADDR/MASTER_PORT (launcher/src/main.rs:697-698)."""
import os
from datetime import timedelta

import torch

RANK = int(os.getenv("RANK", "0"))
WORLD_SIZE = int(os.getenv("WORLD_SIZE", "1"))


class FakeBarrier:
    def wait(self):
        pass


class FakeGroup:
    def __init__(self, rank, size):
        self._rank = rank
        self._size = size

    def allreduce(self, *args, **kwargs):
        return FakeBarrier()

    def allgather(self, inputs, local_tensor, **kwargs):
        assert len(inputs[0]) == len(local_tensor) == 1, f"{len(inputs[0])} != {len(local_tensor)} != 1"
        for input_ in inputs:
            input_[0].data = local_tensor[0].data
        return FakeBarrier()

    def barrier(self, *args, **kwargs):
        return FakeBarrier()

    def size(self):
        return self._size

    def rank(self):
        return self._rank


def print_rank_n(*values, rank=0):
    if RANK == rank:
        print(*values, flush=True)


def get_torch_dtype(dtype_str: str) -> torch.dtype:
    dt = getattr(torch, dtype_str, None)
    if type(dt) != torch.dtype:
        raise ValueError(f"Unrecognized data type: {dtype_str}")
    return dt


def initialize_torch_distributed(world_size: int, rank: int):
    if world_size == 1 or os.getenv("DEBUG", None) == "1":
        return FakeGroup(rank, world_size)
    if not torch.distributed.is_initialized():
        # 60 s as in the reference; benchmarks on a cold box raise it (first imports and weight set-up skew the ranks)
        timeout = timedelta(seconds=int(os.getenv("TGIS_DIST_TIMEOUT_S", "60")))
        # TGIS_DIST_BACKEND=gloo: host-mediated collectives on GPU tensors, for TP tests on a single-GPU box
        if torch.cuda.is_available() and os.getenv("TGIS_DIST_BACKEND", "nccl") != "gloo":
            from torch.distributed import ProcessGroupNCCL

            backend = "nccl"  # RCCL on ROCm
            options = ProcessGroupNCCL.Options()
            options.is_high_priority_stream = True
            options._timeout = timeout
        else:
            backend = "gloo"
            options = None
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.distributed.init_process_group(
            backend=backend, world_size=world_size, rank=rank, timeout=timeout, pg_options=options)
    else:
        print("WARN: torch.distributed is already initialized")
    return torch.distributed.group.WORLD
