"""Prompt-prefix ("soft prompt") cache for decoder-only models: SURVEY.md §8(f) row 4.

Behaviour follows the reference's `prompt_cache.py:145-462` as seen by `FlashCausalLMBatch.from_pb`
(`flash_causal_lm.py:97-107,157-168`) and the `PrefixLookup` RPC (`server.py:86-103`):

* a prefix id names a directory under `$PREFIX_STORE_PATH`; it holds either `decoder.pt` (a `[n_virtual, hidden]`
  tensor saved with `torch.save`) or a peft prompt-tuning adapter (`adapter_model.safetensors` / `.bin`, tensor
  `prompt_embeddings`) — :219-260,278-292;
* ids are restricted to `[/\\w-]+` and may not escape the store — :205-217;
* tensors are validated (2-D, `1 <= rows <= max_length`, `cols == hidden`, finite before and after the cast to the
  model dtype) — :294-330;
* entries live on the device in the model dtype in an LRU bounded by `$PROMPT_CACHE_SIZE_MB` (default 512 MiB, sizes
  rounded up to 512 B) — :349-409; a prefix larger than the whole budget is refused;
* `return_zero` (ranks > 0 of a tensor-parallel embedding that reduces later) yields a zero tensor of the same shape
  so that the all-reduce adds the prefix once — `models/model.py:76-82`, `prompt_cache.py:332-347`.

Encoder-decoder prefixes (`encoder.pt`) are outside the hot path (decoder-only flash models) and are
rejected.  The LRU is an `OrderedDict` guarded by one lock; loads from disk happen outside the lock, as in the
reference, so a slow disk never blocks lookups of cached ids."""
import logging
import math
import os
import re
import threading
from collections import OrderedDict
from pathlib import Path
from typing import Optional, Tuple

import torch

logger = logging.getLogger(__name__)

VALID_PREFIX_ID = re.compile(r"[/\w\-]+")


def prefix_store_path() -> Optional[Path]:
    p = os.getenv("PREFIX_STORE_PATH")
    return Path(p) if p else None


def cache_budget_mb() -> int:
    return int(os.getenv("PROMPT_CACHE_SIZE_MB", "512"))


class PrefixNotFound(Exception):
    pass


def _size_mb(t: torch.Tensor) -> float:
    raw = t.element_size() * t.nelement()
    return math.ceil(raw / 512) * 512 / (1024 ** 2)


class PrefixCache:
    def __init__(self, device: torch.device, dtype: torch.dtype, max_length: int, hidden_size: Optional[int],
                 return_zero: bool = False, store: Optional[Path] = None, budget_mb: Optional[int] = None):
        self.device, self.dtype = device, dtype
        self.max_length = max_length
        self.hidden_size = hidden_size
        self.zero = torch.zeros((1,), dtype=dtype, device=device) if return_zero else None
        self.store = store if store is not None else prefix_store_path()
        self.budget_mb = budget_mb if budget_mb is not None else cache_budget_mb()
        self._lru: "OrderedDict[str, Tuple[torch.Tensor, float]]" = OrderedDict()
        self._used_mb = 0.0
        self._lock = threading.Lock()

    # ---- lookups ------------------------------------------------------------------------------------------------
    def get(self, prefix_id: str) -> torch.Tensor:
        """The `[n_virtual, hidden]` prompt tensor for `prefix_id`, loading and caching it on first use."""
        hit = self._touch(prefix_id)
        if hit is not None:
            return hit
        tensor = self._load(prefix_id)
        return self._insert(prefix_id, tensor)

    def __len__(self) -> int:
        return len(self._lru)

    def keys(self):
        with self._lock:
            return list(self._lru.keys())  # least recently used first

    @property
    def size_mb(self) -> float:
        return self._used_mb

    def clear(self) -> None:
        with self._lock:
            self._lru.clear()
            self._used_mb = 0.0

    # ---- internals ----------------------------------------------------------------------------------------------
    def _touch(self, prefix_id: str) -> Optional[torch.Tensor]:
        with self._lock:
            entry = self._lru.get(prefix_id)
            if entry is None:
                return None
            self._lru.move_to_end(prefix_id)
            return entry[0]

    def _dir(self, prefix_id: str) -> Path:
        if self.store is None:
            raise PrefixNotFound("no PREFIX_STORE_PATH is configured")
        if not VALID_PREFIX_ID.fullmatch(prefix_id):
            raise ValueError(f"Invalid prefix id {prefix_id}, must contain only alphanumeric, _ and - and /")
        d = self.store / prefix_id
        if not os.path.normpath(d).startswith(str(self.store).rstrip("/") + "/"):
            raise ValueError(f"Invalid prefix id {prefix_id}")
        return d

    @staticmethod
    def _read(path: Path):
        if not path.is_file():
            return None
        logger.info("Loading new prefix %s", path)
        if path.suffix == ".safetensors":
            from safetensors.torch import load_file
            return load_file(str(path), device="cpu")
        return torch.load(path, weights_only=True, map_location=torch.device("cpu"))

    def _load(self, prefix_id: str) -> torch.Tensor:
        d = self._dir(prefix_id)
        raw = None
        if d.is_dir() and any(f.stem == "adapter_model" for f in d.iterdir()):  # peft.save_pretrained() layout
            data = self._read(d / "adapter_model.safetensors")
            if data is None:
                data = self._read(d / "adapter_model.bin")
            if data is not None:
                raw = data["prompt_embeddings"]
        else:
            if (d / "encoder.pt").is_file():
                raise ValueError(f"Prefix id {prefix_id}: encoder prompts are not supported by decoder-only models")
            raw = self._read(d / "decoder.pt")
        if raw is None:
            raise PrefixNotFound(f"Prefix id {prefix_id} not found")
        prefix = self._validate(raw)
        if self.zero is not None:
            return self.zero.expand(prefix.shape)
        return prefix.to(self.device, non_blocking=True)

    def _validate(self, prefix) -> torch.Tensor:
        if not torch.is_tensor(prefix) or prefix.dim() != 2:
            raise ValueError("Invalid prefix embedding tensor")
        if prefix.shape[0] == 0 or prefix.shape[0] > self.max_length:
            raise ValueError(f"Invalid prefix embedding length of {prefix.shape[0]}")
        if self.hidden_size is not None and prefix.shape[1] != self.hidden_size:
            raise ValueError(f"Prefix embedding tensor dim {prefix.shape[1]} does not match model ({self.hidden_size})")
        converted = prefix.to(self.dtype)
        if not converted.isfinite().all():
            if not prefix.isfinite().all():
                raise ValueError("Prefix contains non-finite elements")
            raise ValueError(f"Prefix contains non-finite elements after conversion from {prefix.dtype} to {self.dtype}")
        return converted.requires_grad_(False)

    def _insert(self, prefix_id: str, tensor: torch.Tensor) -> torch.Tensor:
        size = _size_mb(tensor)
        evicted = []
        with self._lock:
            entry = self._lru.get(prefix_id)  # another thread may have loaded it meanwhile
            if entry is not None:
                self._lru.move_to_end(prefix_id)
                return entry[0]
            if size > self.budget_mb:
                raise ValueError(f"Prefix ID object {prefix_id} exceeds the allowed cache size")
            while self._used_mb + size > self.budget_mb:
                old_id, (_, old_size) = self._lru.popitem(last=False)
                self._used_mb -= old_size
                evicted.append(old_id)
            self._lru[prefix_id] = (tensor, size)
            self._used_mb += size
            total = self._used_mb
        if evicted:
            logger.info("Deleted prefixes %s from the prompt cache", evicted)
        logger.info("Added prefix %s to the prompt cache, has %d virtual tokens, size %.3fMiB, total cache size is now "
                    "%.2fMiB", prefix_id, tensor.shape[0], size, total)
        return tensor


def max_prompt_prefix_length(max_seq_length: int) -> int:
    """Half the maximum sequence length unless $MAX_PROMPT_PREFIX_LENGTH says otherwise (models/model.py:59-70)."""
    limit = math.ceil(max_seq_length * 0.5)
    env = os.getenv("MAX_PROMPT_PREFIX_LENGTH")
    if env:
        try:
            value = int(env)
        except ValueError as exc:
            raise ValueError("Invalid value for MAX_PROMPT_PREFIX_LENGTH") from exc
        if value > max_seq_length - 1:
            raise ValueError(f"Value for the MAX_PROMPT_PREFIX_LENGTH ({value}) cannot be larger than the max sequence "
                             f"length - 1 ({max_seq_length - 1})")
        limit = value
    return limit
