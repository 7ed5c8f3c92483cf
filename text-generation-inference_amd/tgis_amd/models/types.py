"""What the servicer needs from a batch type, and the error record it reports per request.

A batch is a set of requests decoded in lock-step.  The shard keeps live batches in its `Cache` under `batch_id`
between RPCs; the servicer builds one from a `generate.v1.Batch` (`from_pb`), merges the batches a `NextToken` names
(`concatenate`, which consumes its inputs) and drops finished requests (`prune`, which mutates and returns the batch,
or None when nothing is left) — the lifecycle of the reference's models/types.py:27-62."""
import abc
from typing import List, Optional, Tuple

import torch

from tgis_amd.pb import generate_pb2 as pb2


class GenerateError:
    """A request-scoped failure (tokenisation, validation, token decoding) reported next to the tokens."""
    __slots__ = ("request_id", "message")

    def __init__(self, request_id: int, message: str):
        self.request_id, self.message = request_id, message

    def __eq__(self, other) -> bool:
        return isinstance(other, GenerateError) and (self.request_id, self.message) == (other.request_id, other.message)

    def __repr__(self) -> str:
        return f"GenerateError(request_id={self.request_id}, message={self.message!r})"

    def to_pb(self) -> pb2.GenerateError:
        return pb2.GenerateError(request_id=self.request_id, message=self.message)


class Batch(abc.ABC):
    batch_id: int

    # ---- construction and membership ------------------------------------------------------------------------------
    @classmethod
    @abc.abstractmethod
    def from_pb(cls, pb: pb2.Batch, tokenizer, dtype: torch.dtype, device: torch.device, embeddings_lookup: Optional,
                prefix_cache: Optional, use_position_ids: bool = False,
                ) -> Tuple[Optional["Batch"], List[GenerateError]]:
        """(batch or None if every request failed validation, per-request errors)."""

    @classmethod
    @abc.abstractmethod
    def concatenate(cls, batches: List["Batch"]) -> "Batch":
        """One batch holding every request of `batches`; the inputs must not be used afterwards."""

    @classmethod
    @abc.abstractmethod
    def prune(cls, batch: "Batch", completed_ids: List[int]) -> Optional["Batch"]:
        """`batch` without the completed requests, or None if none remain."""

    # ---- introspection ------------------------------------------------------------------------------------------------
    @abc.abstractmethod
    def get_id(self) -> int:
        """The id the router knows this batch by."""

    @abc.abstractmethod
    def __len__(self) -> int:
        """Number of live requests."""

    def compact(self) -> None:
        """Give back over-allocated storage, if the batch type keeps any (optional)."""

    def release(self) -> None:
        """Return device resources the batch owns outside its tensors (KV pages of the paged batch).  The servicer calls
        this on every batch it drops — after a failed step, a health check or an emptied prune — so every batch type
        answers it; batches that own nothing but tensors (the padded `CausalLMBatch`) have nothing to do."""
