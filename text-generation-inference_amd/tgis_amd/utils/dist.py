"""Process-group bootstrap: one process per GPU; the collectives run on RCCL (PyTorch-ROCm's `nccl` backend) over xGMI.

Behaviour the rest of the package relies on, and that a user of the reference's `utils/dist.py:21-96` expects:
`initialize_torch_distributed(world_size, rank)` returns an object with `.size()` / `.rank()` — the WORLD process group
when there are several shards, a local stand-in (`FakeGroup`) when there is one (or `DEBUG=1`).  GPU shards use `nccl`
with a high-priority communication stream; CPU runs use `gloo` (the world-size-2 tests).  Rendezvous is
`MASTER_ADDR` / `MASTER_PORT` as the launcher exports them (launcher/src/main.rs:697-698), defaulting to
127.0.0.1:29500.

Knobs that are ours: `TGIS_DIST_TIMEOUT_S` (collective timeout, 60 s as in the reference unless a benchmark on a cold
box raises it) and `TGIS_DIST_BACKEND=gloo` (host-mediated collectives on GPU tensors, so that tensor-parallel code can
be exercised with several ranks on a single-GPU box)."""
import os
from dataclasses import dataclass
from datetime import timedelta

import torch

RANK = int(os.environ.get("RANK", 0))
WORLD_SIZE = int(os.environ.get("WORLD_SIZE", 1))


class _Done:
    """What a finished collective hands back: something to `.wait()` on."""

    @staticmethod
    def wait():
        return None


@dataclass(frozen=True)
class FakeGroup:
    """Stand-in for a process group of one: collectives are identities.  Also used to compute what shard `rank_` of
    `size_` would load (weight-sharding tests, the single-GPU TP emulation in tools/)."""
    rank_: int
    size_: int

    def rank(self) -> int:
        return self.rank_

    def size(self) -> int:
        return self.size_

    def barrier(self, *_, **__):
        return _Done

    def allreduce(self, *_, **__):  # a sum over one rank
        return _Done

    def allgather(self, outputs, inputs, **__):
        if not (len(outputs) and len(outputs[0]) == 1 and len(inputs) == 1):
            raise AssertionError("FakeGroup.allgather handles exactly one tensor from one rank")
        for per_rank in outputs:
            per_rank[0].data = inputs[0].data
        return _Done


def print_rank_n(*values, rank: int = 0) -> None:
    """print() on one rank only."""
    if rank == RANK:
        print(*values, flush=True)


def get_torch_dtype(dtype_str: str) -> torch.dtype:
    """"float16" -> torch.float16; anything that does not name a torch dtype is an error."""
    candidate = getattr(torch, dtype_str, None)
    if isinstance(candidate, torch.dtype):
        return candidate
    raise ValueError(f"Unrecognized data type: {dtype_str}")


def _backend_and_options(timeout: timedelta):
    use_rccl = torch.cuda.is_available() and os.environ.get("TGIS_DIST_BACKEND", "nccl") != "gloo"
    if not use_rccl:
        return "gloo", None
    from torch.distributed import ProcessGroupNCCL

    opts = ProcessGroupNCCL.Options()
    opts.is_high_priority_stream = True  # collectives are on the critical path of every layer
    opts._timeout = timeout
    return "nccl", opts


def initialize_torch_distributed(world_size: int, rank: int):
    single = world_size == 1 or os.environ.get("DEBUG") == "1"
    if single:
        return FakeGroup(rank, world_size)
    if torch.distributed.is_initialized():
        print("WARN: torch.distributed is already initialized")
        return torch.distributed.group.WORLD
    timeout = timedelta(seconds=int(os.environ.get("TGIS_DIST_TIMEOUT_S", 60)))
    backend, options = _backend_and_options(timeout)
    for var, default in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29500")):
        os.environ.setdefault(var, default)
    torch.distributed.init_process_group(backend=backend, rank=rank, world_size=world_size, timeout=timeout,
                                         pg_options=options)
    return torch.distributed.group.WORLD
