"""Paged KV cache: one statically allocated page pool per model, 32-token pages, block tables per request.

This replaces the reference's per-batch contiguous `past_key_values[L, total_slots, 2, Hkv, D]` tensor that
is re-concatenated on every decode step, concatenate and prune (models/flash_causal_lm.py:246,281,
307-333,439-447): membership changes and growth become page-table edits, the KV bytes never move.
The *logical* slot index the reference exposes (`cu_seqlens[1:] - 1`, flash_llama_modeling.py:465) is kept
by FlashCausalLMBatch; `PagedKVCache` only maps (request, position) -> physical (page, offset).
Design cue for paging: models/paged_causal_lm.py:300-348 (block size 16 there; 32 here = one MFMA K=32 step).
"""
import heapq
from typing import List

import torch

PAGE = 32


class OutOfPages(RuntimeError):
    """Maps to gRPC RESOURCE_EXHAUSTED like a CUDA OOM does in the reference (server.py:48-51)."""


class PagedKVCache:
    def __init__(self, num_layers: int, num_kv_heads: int, head_dim: int, num_pages: int, dtype, device):
        self.num_layers, self.num_kv_heads, self.head_dim = num_layers, num_kv_heads, head_dim
        self.num_pages = num_pages
        # zero-initialised: masked slots are multiplied by P = 0, so they must never hold NaN/Inf patterns
        self.pool = torch.zeros((num_layers, 2, num_pages, num_kv_heads, PAGE * head_dim), dtype=dtype, device=device)
        # Pages are handed out lowest id first, and a batch spreads what it takes PAGE-MAJOR over its sequences
        # (FlashCausalLMBatch.allocate_pages: page p of every sequence before page p + 1 of any): the decode kernel's blocks
        # walk their sequences' pages in step, so what the chip reads at one instant is then one dense run of the pool,
        # spread evenly over the HBM channels.  Measured at the cfg3 shape (tools/attn_page_order.py, profiles/
        # r05_attn_page_order.log): page-major 80.7 us, a random order (rounds 1-4) 82.6, a sequence's pages next to each
        # other 88-92 (equal-phase blocks camp on the same channels).  A pool that has churned degrades to the random case.
        self._free = list(range(num_pages))  # a heap (heapq), trivially one to begin with

    @property
    def free_pages(self) -> int:
        return len(self._free)

    def bytes_per_token(self) -> int:
        return self.num_layers * 2 * self.num_kv_heads * self.head_dim * self.pool.element_size()

    def k_pool(self, layer: int) -> torch.Tensor:
        return self.pool[layer, 0]

    def v_pool(self, layer: int) -> torch.Tensor:
        return self.pool[layer, 1]

    def alloc(self, n: int) -> List[int]:
        """The n lowest free page ids, ascending."""
        if n > len(self._free):
            raise OutOfPages(f"KV cache exhausted: need {n} pages, {len(self._free)} free of {self.num_pages}")
        return [heapq.heappop(self._free) for _ in range(n)]

    def free(self, pages: List[int]):
        for p in pages:
            heapq.heappush(self._free, p)

    @staticmethod
    def pages_for(tokens: int) -> int:
        return (tokens + PAGE - 1) // PAGE
