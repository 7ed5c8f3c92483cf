"""Paged KV cache: one statically allocated page pool per model, 32-token pages, block tables per request.

This replaces the reference's per-batch contiguous `past_key_values[L, total_slots, 2, Hkv, D]` tensor that
is re-concatenated on every decode step, concatenate and prune (models/flash_causal_lm.py:246,281,
307-333,439-447): membership changes and growth become page-table edits, the KV bytes never move.
The *logical* slot index the reference exposes (`cu_seqlens[1:] - 1`, flash_llama_modeling.py:465) is kept
by FlashCausalLMBatch; `PagedKVCache` only maps (request, position) -> physical (page, offset).
Design cue for paging: models/paged_causal_lm.py:300-348 (block size 16 there; 32 here = one MFMA K=32 step).

Where a sequence's pages sit in the pool (rounds 5 and 6).  The decode attention's blocks walk their sequences' pages in
step, so what the chip reads at one instant is "page p of every sequence of the batch".  When those pages are one dense run
of the pool the launch is fastest (cfg3: 80.7 us; a random order 82.6; a sequence's own pages side by side 88-92,
profiles/r05_attn_page_order.log) — the run spreads evenly over the HBM channels.  A pool that hands out its lowest free id
gives that only while it is pristine.  What survives frees is the weaker property that the pages of one such "row" cover the
residues of the page id modulo C evenly: the free pages are kept in C heaps, one per residue class, every sequence gets a
LANE when it enters the cache (the lane with the fewest live sequences, so the sequences alive together spread over the lanes), and page p of the
sequence in lane l is taken from class (l + p) mod C, lowest id first.  On a pristine pool this is still the dense page-major
run (row p of a batch of C sequences is the ids [p C, p C + C)); after any amount of churn each row still holds every residue
class equally often (tests/test_batch_cpu.py, profiles/r06_page_classes.log for the measured effect and the choice of C).
"""
import heapq
import os
from typing import List, Optional, Sequence

import torch

PAGE = 32


class OutOfPages(RuntimeError):
    """Maps to gRPC RESOURCE_EXHAUSTED like a CUDA OOM does in the reference (server.py:48-51)."""


def default_classes(page_bytes: int, num_pages: int) -> int:
    """Residue classes of the free list: the page ids of one row of a batch should cover a span of SPAN bytes of the pool
    evenly (the span over which the memory channels interleave, measured: profiles/r06_page_classes.log); never more classes
    than an eighth of the pool (every class must keep pages to give)."""
    env = os.getenv("TGIS_KV_CLASSES")
    if env:
        return max(1, int(env))
    span = 8 << 20
    c = 1
    while c * 2 * page_bytes <= span and c * 2 <= 64 and c * 16 <= num_pages:
        c *= 2
    return c


class PagedKVCache:
    def __init__(self, num_layers: int, num_kv_heads: int, head_dim: int, num_pages: int, dtype, device,
                 classes: Optional[int] = None):
        self.num_layers, self.num_kv_heads, self.head_dim = num_layers, num_kv_heads, head_dim
        self.num_pages = num_pages
        # zero-initialised, and every value a kernel ever writes into it is finite: masked slots of a page (the unwritten tail
        # of the page a sequence is filling, stale tokens of a page's previous owner) are multiplied by P = 0 and must never
        # hold NaN / Inf patterns.  One page more than `num_pages`: the NULL page (id num_pages), never handed out — inactive
        # rows of a padded decode graph (FlashCausalLM, batch-size buckets) write their token there and attend to it.
        self.pool = torch.zeros((num_layers, 2, num_pages + 1, num_kv_heads, PAGE * head_dim), dtype=dtype, device=device)
        self.null_page = num_pages
        page_bytes = num_kv_heads * PAGE * head_dim * self.pool.element_size()
        self.reset_free_lists(int(classes) if classes else default_classes(page_bytes, num_pages))

    def reset_free_lists(self, classes: int):
        """Every page free, `classes` residue classes (construction; bench.py --churn re-runs one pool under several policies)."""
        assert getattr(self, "_nfree", self.num_pages) == self.num_pages, "pages are still held"
        self.classes = C = max(1, int(classes))
        self._free = [list(range(c, self.num_pages, C)) for c in range(C)]  # one heap per residue class (sorted: heaps)
        self._nfree = self.num_pages
        self._next_lane = 0
        self._lane_use = [0] * C  # live sequences per lane

    @property
    def free_pages(self) -> int:
        return self._nfree

    def bytes_per_token(self) -> int:
        return self.num_layers * 2 * self.num_kv_heads * self.head_dim * self.pool.element_size()

    def k_pool(self, layer: int) -> torch.Tensor:
        return self.pool[layer, 0]

    def v_pool(self, layer: int) -> torch.Tensor:
        return self.pool[layer, 1]

    def new_lanes(self, n: int) -> List[int]:
        """Lanes for n sequences entering the cache: each takes the lane with the fewest live sequences (ties: the next one of
        a rotating counter), so the live sequences' lanes — and with them the residue classes of every row of their pages —
        stay as even as they can be whatever has joined and left.  `drop_lanes` gives them back."""
        C, use, out = self.classes, self._lane_use, []
        for _ in range(n):
            lo = min(use)
            lane = next(l % C for l in range(self._next_lane, self._next_lane + C) if use[l % C] == lo)
            self._next_lane = (lane + 1) % C
            use[lane] += 1
            out.append(lane)
        return out

    def drop_lanes(self, lanes: Sequence[int]):
        for l in lanes:
            self._lane_use[l] -= 1

    def alloc_classes(self, wants: Sequence[int]) -> List[int]:
        """One page per entry of `wants`, the lowest free id of residue class wants[i] mod C; a class that has run dry is
        replaced by the one with the most pages left.  All or nothing: OutOfPages before anything is taken."""
        n = len(wants)
        if n > self._nfree:
            raise OutOfPages(f"KV cache exhausted: need {n} pages, {self._nfree} free of {self.num_pages}")
        C, free = self.classes, self._free
        out = []
        for w in wants:
            h = free[w % C]
            if not h:
                h = max(free, key=len)
            out.append(heapq.heappop(h))
        self._nfree -= n
        return out

    def alloc(self, n: int) -> List[int]:
        """n pages, one from each of the classes 0, 1, ... in turn (a pristine pool: the n lowest ids, ascending)."""
        return self.alloc_classes(range(n))

    def free(self, pages: List[int]):
        C = self.classes
        for p in pages:
            heapq.heappush(self._free[p % C], p)
        self._nfree += len(pages)

    @staticmethod
    def pages_for(tokens: int) -> int:
        return (tokens + PAGE - 1) // PAGE
