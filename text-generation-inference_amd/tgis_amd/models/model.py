"""Model ABC: what the shard servicer drives (mirrors models/model.py:34-188 of the reference).

`generate_token(batch, first, for_concat) -> (List[TokenInfo], Optional[List[InputTokens]],
List[GenerateError], forward_time_ns)` is THE hot-path entry point (models/model.py:169-173).
Attributes read by the servicer: tokenizer, dtype, device, word_embeddings, prefix_cache,
use_position_ids, context_manager, engine, config, model (server.py:131-139,326-332)."""
from abc import ABC, abstractmethod
from typing import List, Optional, Tuple, Type, TypeVar

import torch

from tgis_amd.models.types import Batch, GenerateError
from tgis_amd.pb import generate_pb2
from tgis_amd.prompt_cache import PrefixCache, max_prompt_prefix_length, prefix_store_path
from tgis_amd.utils.token_types import InputTokens, TokenInfo

B = TypeVar("B", bound=Batch)


class Model(ABC):
    def __init__(self, engine, dtype: torch.dtype, max_seq_length: Optional[int] = None):
        self.engine = engine
        self.config, self.tokenizer, self.model = engine.get_components()
        self.device = engine.get_device()
        self.dtype = dtype
        self.max_seq_length = max_seq_length
        # the tokenizer config does not always carry the eos id (model.py:42-43)
        if getattr(self.config, "eos_token_id", None) is not None and self.tokenizer is not None:
            self.tokenizer.model_eos_token_id = self.config.eos_token_id
        self.use_position_ids = True
        try:
            self.word_embeddings = self.model.get_input_embeddings()
        except Exception:
            self.word_embeddings = None
        # prompt-prefix (soft prompt) cache, enabled by $PREFIX_STORE_PATH (models/model.py:56-95 of the reference)
        self.prefix_cache = None
        if prefix_store_path() is not None and self.word_embeddings is not None:
            if max_seq_length is None:
                raise ValueError("max_seq_length must be set when a prompt prefix store is configured")
            from tgis_amd.utils.layers import TensorParallelEmbedding
            return_zero = False
            # an embedding that leaves the all-reduce to its caller holds partial rows on every rank: only rank 0 may
            # add the real prefix, the others add zeros (model.py:76-82)
            if isinstance(self.word_embeddings, TensorParallelEmbedding) and not self.word_embeddings.reduce:
                return_zero = self.word_embeddings.process_group.rank() != 0
            self.prefix_cache = PrefixCache(
                device=self.device, dtype=dtype, max_length=max_prompt_prefix_length(max_seq_length),
                hidden_size=getattr(self.config, "hidden_size", None) or getattr(self.config, "n_embd", None),
                return_zero=return_zero)
        self.context_manager = torch.inference_mode

    @property
    @abstractmethod
    def batch_type(self) -> Type[B]:
        ...

    @abstractmethod
    def generate_token(self, batch: B, first: bool = False, for_concat: bool = False,
                       ) -> Tuple[List[TokenInfo], Optional[List[InputTokens]], List[GenerateError], int]:
        ...

    @staticmethod
    def get_indices_to_keep(requests: List[generate_pb2.Request], completed_ids: List[int]) -> List[int]:
        """Indices of `requests` whose id is NOT in `completed_ids`.  Both sequences ascend by id (the router
        allocates ids monotonically, router/src/queue.rs:161-162), so one merge pass suffices — the same
        assumption the reference makes (models/model.py:175-188)."""
        keep = []
        it = iter(completed_ids)
        nxt = next(it, None)
        for i, r in enumerate(requests):
            while nxt is not None and nxt < r.id:
                nxt = next(it, None)
            if nxt != r.id:
                keep.append(i)
        return keep
