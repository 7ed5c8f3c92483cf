"""Batch lifecycle contract used by the servicer (mirrors models/types.py:15-62 of the reference)."""
from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch

from tgis_amd.pb import generate_pb2


@dataclass
class GenerateError:
    request_id: int
    message: str

    def to_pb(self) -> generate_pb2.GenerateError:
        return generate_pb2.GenerateError(request_id=self.request_id, message=self.message)


class Batch(ABC):
    """A set of requests decoded in lock-step.  Instances are keyed by `batch_id` in the shard's Cache
    between RPCs; `concatenate` consumes its inputs, `prune` mutates and returns the batch (or None)."""

    batch_id: int

    @abstractmethod
    def get_id(self) -> int:
        ...

    @abstractmethod
    def __len__(self) -> int:
        ...

    @classmethod
    @abstractmethod
    def from_pb(cls, pb: generate_pb2.Batch, tokenizer, dtype: torch.dtype, device: torch.device,
                embeddings_lookup: Optional, prefix_cache: Optional, use_position_ids: bool = False,
                ) -> Tuple[Optional["Batch"], List[GenerateError]]:
        ...

    @classmethod
    @abstractmethod
    def concatenate(cls, batches: List["Batch"]) -> "Batch":
        ...

    @classmethod
    @abstractmethod
    def prune(cls, batch: "Batch", completed_ids: List[int]) -> Optional["Batch"]:
        ...

    def compact(self):
        """Optional: release over-allocated storage."""
