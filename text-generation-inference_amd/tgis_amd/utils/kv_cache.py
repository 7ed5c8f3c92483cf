"""Paged KV cache: one statically allocated page pool per model, 32-token pages, block tables per request.

This replaces the reference's per-batch contiguous `past_key_values[L, total_slots, 2, Hkv, D]` tensor that
is re-concatenated on every decode step, concatenate and prune (models/flash_causal_lm.py:246,281,
307-333,439-447): membership changes and growth become page-table edits, the KV bytes never move.
The *logical* slot index the reference exposes (`cu_seqlens[1:] - 1`, flash_llama_modeling.py:465) is kept
by FlashCausalLMBatch; `PagedKVCache` only maps (request, position) -> physical (page, offset).
Design cue for paging: models/paged_causal_lm.py:300-348 (block size 16 there; 32 here = one MFMA K=32 step).

Where a sequence's pages sit in the pool.  The decode attention's blocks walk their sequences' pages in step, so what the
chip reads at one instant is "page p of every sequence of the batch".  The pool hands out its LOWEST free ids and a batch deals
what it takes PAGE-MAJOR over its sequences (FlashCausalLMBatch.allocate_pages): on a pristine pool those rows are dense runs
(cfg3: 80.7 us per launch against 82.6 for a random order and 88-92 for a sequence's pages side by side,
profiles/r05_attn_page_order.log), and a finished sequence's column of ids is the next one out, so the request that replaces
it inherits it.  Round 6 measured what churn does to that (bench.py --churn, profiles/r06_churn_cfg3.json: requests leaving
and joining every 4-8 steps on a pool aged by random allocations and frees): decode p50 4.149 ms against 4.144 ms on the
pristine pool — lowest-id-first dealing over the holes of an aged pool is still mostly dense.  A channel-aware free list
(one heap per page-id residue class, page p of sequence b from class (b + p) mod C — VERDICT r05 item 2a) was built and
measured on aged pools for C = 2 .. 128 at five shapes: never better than this one, 1-4 % worse for several C
(profiles/r06_page_classes.log; the allocator lives on in tools/attn_page_order.py), so it is not here.
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
        # zero-initialised, and every value a kernel ever writes into it is finite: masked slots of a page (the unwritten tail
        # of the page a sequence is filling, stale tokens of a page's previous owner) are multiplied by P = 0 and must never
        # hold NaN / Inf patterns (tests/test_ops_gpu.py::test_attention_decode_every_fill_of_the_last_page runs over stale
        # tails of +-65504).  One page more than `num_pages`: the NULL page (id num_pages), never handed out — the inactive
        # rows of a padded decode graph (FlashCausalLM, batch-size buckets) write their token there and attend to it.
        self.pool = torch.zeros((num_layers, 2, num_pages + 1, num_kv_heads, PAGE * head_dim), dtype=dtype, device=device)
        self.null_page = num_pages
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
        """The n lowest free page ids, ascending.  All or nothing: OutOfPages before anything is taken."""
        if n > len(self._free):
            raise OutOfPages(f"KV cache exhausted: need {n} pages, {len(self._free)} free of {self.num_pages}")
        return [heapq.heappop(self._free) for _ in range(n)]

    def free(self, pages: List[int]):
        for p in pages:
            heapq.heappush(self._free, p)

    @staticmethod
    def pages_for(tokens: int) -> int:
        return (tokens + PAGE - 1) // PAGE
