"""CPU: FlashCausalLMBatch host bookkeeping (no kernels involved): from_pb tensors, page ownership through
concatenate / prune / release, and the reference's logical cu_seqlens arithmetic
(models/flash_causal_lm.py:67-194,196-285,290-353)."""
import numpy as np
import pytest
import torch

from tests.fixture_utils import FixtureTokenizer, prompt_text
from tgis_amd.models.flash_causal_lm import FlashCausalLMBatch
from tgis_amd.models.model import Model
from tgis_amd.pb import generate_pb2 as pb2
from tgis_amd.utils.kv_cache import OutOfPages, PagedKVCache

CPU = torch.device("cpu")


def _pb(prompts, max_new, first_id=0, batch_id=0, truncate_to=None):
    reqs = []
    for i, p in enumerate(prompts):
        n = len(p) if truncate_to is None else truncate_to[i]
        reqs.append(pb2.Request(id=first_id + i, inputs=prompt_text(p), input_length=n, truncate=truncate_to is not None,
                                max_output_length=max_new))
    return pb2.Batch(id=batch_id, requests=reqs)


def _batch(prompts, max_new, **kw):
    tok = FixtureTokenizer(256)
    b, errs = FlashCausalLMBatch.from_pb(_pb(prompts, max_new, **kw), tok, torch.float16, CPU, None, None, True)
    assert not errs
    return b


def test_from_pb_concatenates_without_padding():
    prompts = [[5, 6, 7], [9], [11, 12, 13, 14, 15]]
    b = _batch(prompts, 4)
    assert b.input_ids.tolist() == [5, 6, 7, 9, 11, 12, 13, 14, 15]
    assert b.position_ids.tolist() == [0, 1, 2, 0, 0, 1, 2, 3, 4]
    assert b.cu_seqlens.tolist() == [0, 3, 4, 9] and b.cu_seqlens.dtype == torch.int32
    assert b.max_seqlen == 5 and b.input_lengths == [3, 1, 5] and b.total_lengths == [7, 5, 9]
    assert b.all_input_ids_tensor.shape == (3, 9)
    assert b.all_input_ids_tensor[1].tolist() == [9] + [0] * 8  # pad-filled
    assert b.past_key_values is None and b.pages is None and len(b) == 3 and b.get_id() == 0


def test_from_pb_left_truncation_and_bos():
    tok = FixtureTokenizer(256)
    tok.add_bos_token = True
    b, _ = FlashCausalLMBatch.from_pb(_pb([[5, 6, 7, 8, 9]], 2, truncate_to=[3]), tok, torch.float16, CPU, None, None)
    assert b.input_ids.tolist() == [1, 8, 9]  # last 3 tokens kept, BOS re-inserted in front


def test_pages_are_taken_lowest_first_and_page_major():
    """kv_cache.py: the pool hands out its lowest free ids, and a batch lays them
    PAGE-major over its sequences (what the decode blocks read at one instant is then a dense run of the pool); a finished
    sequence's pages are the next ones out."""
    cache = PagedKVCache(1, 1, 64, 32, torch.float16, CPU)
    a = _batch([[5] * 70, [6] * 33, [7] * 3], 40, batch_id=1)   # 3 + 2 + 1 pages
    a.allocate_pages(cache)
    assert a.pages == [[0, 3, 5], [1, 4], [2]]
    a.input_lengths = [97, 65, 4]
    a.grow_pages()                                              # the two sequences that cross a page: neighbours again
    assert a.pages == [[0, 3, 5, 6], [1, 4, 7], [2]]
    cache.free(a.pages[1])
    b = _batch([[8] * 40], 5, first_id=5, batch_id=2)
    b.allocate_pages(cache)
    assert b.pages == [[1, 4]] and cache.free_pages == 32 - 8 + 3 - 2
    a.pages[1] = []
    a.release()
    b.release()
    assert cache.free_pages == 32 and cache.alloc(4) == [0, 1, 2, 3]


def test_churned_pool_keeps_columns_and_the_null_page_is_never_handed_out():
    """Round 6: requests join and leave a running batch (prefill + concatenate + prune, the way the router drives the shard,
    flash_causal_lm.py:196-353).  Lowest-id-first dealing means the request that replaces a finished one inherits its pages
    (its column of the page-major layout); the null page (inactive rows of a bucketed decode graph) is outside the free list;
    nothing leaks."""
    import random

    rng = random.Random(7)
    B, npages = 8, 512
    cache = PagedKVCache(1, 1, 64, npages, torch.float16, CPU)
    assert cache.null_page == npages and cache.pool.shape[2] == npages + 1
    run = _batch([[5] * 70 for _ in range(B)], 400, batch_id=1)
    run.allocate_pages(cache)
    assert [run.pages[i][p] for p in range(3) for i in range(B)] == list(range(3 * B))   # dense rows on the pristine pool
    run.cu_seqlens_q = torch.arange(B + 1, dtype=torch.int32)
    next_id = B
    for step in range(120):
        run.input_lengths = [n + 1 for n in run.input_lengths]
        run.position_ids = torch.tensor(run.input_lengths)
        run.input_ids = torch.zeros(len(run), dtype=torch.int64)
        run.grow_pages()
        if step % 6 == 5:
            victim = rng.choice(run.requests).id
            freed = sorted(run.pages[[r.id for r in run.requests].index(victim)])
            run = FlashCausalLMBatch.prune(run, [victim])
            new = _batch([[6] * 70], 400, first_id=next_id, batch_id=2 + step)
            next_id += 1
            new.allocate_pages(cache)
            assert new.pages[0] == freed[:3], "the newcomer takes the lowest free ids: the column that was just vacated"
            new.cu_seqlens_q = torch.arange(2, dtype=torch.int32)
            new.position_ids = torch.tensor(new.input_lengths)
            new.input_ids = torch.zeros(1, dtype=torch.int64)
            run = FlashCausalLMBatch.concatenate([run, new])
        assert all(pg < npages for p in run.pages for pg in p), "the null page was handed out"
    run.release()
    assert cache.free_pages == npages


def test_page_ownership_concat_prune_release():
    cache = PagedKVCache(2, 2, 64, 16, torch.float16, CPU)
    a = _batch([[5] * 70, [6] * 33], 40, batch_id=1)         # prompt + first token: 71 and 34 slots -> 3 + 2 pages
    a.allocate_pages(cache)                                  # (max_output_length 40 reserves nothing up front)
    assert [len(p) for p in a.pages] == [3, 2] and cache.free_pages == 11
    assert a.block_tables.shape == (2, 8) and a.block_tables.dtype == torch.int32
    # decode growth: a page is taken when a sequence's next token crosses onto it, the table is edited in place
    a.input_lengths = [96, 34]
    a.grow_pages()
    assert [len(p) for p in a.pages] == [3, 2] and cache.free_pages == 11
    a.input_lengths = [97, 65]
    a.grow_pages()
    assert [len(p) for p in a.pages] == [4, 3] and cache.free_pages == 9
    assert a.block_tables[0, :4].tolist() == a.pages[0] and a.block_tables[1, :3].tolist() == a.pages[1]
    cache.free([a.pages[0].pop(), a.pages[1].pop()])
    a.input_lengths = [70, 33]
    b = _batch([[7] * 10], 5, first_id=2, batch_id=2)
    b.allocate_pages(cache)
    a.cu_seqlens_q = torch.arange(3, dtype=torch.int32)
    b.cu_seqlens_q = torch.arange(2, dtype=torch.int32)
    m = FlashCausalLMBatch.concatenate([a, b])
    assert a.pages is None and b.pages is None and [len(p) for p in m.pages] == [3, 2, 1]
    assert m.batch_id == 1 and [r.id for r in m.requests] == [0, 1, 2] and cache.free_pages == 10
    assert m.cu_seqlens.tolist() == [0, 70, 103, 113]
    # prune the middle request: its pages return to the pool, logical cu_seqlens are re-packed
    m.position_ids = torch.tensor([70, 33, 10])
    kept = FlashCausalLMBatch.prune(m, [1])
    assert kept is m and [r.id for r in m.requests] == [0, 2] and cache.free_pages == 12
    assert m.cu_seqlens.tolist() == [0, 71, 82]  # cumsum(position + 1): each kept run plus its free slot
    assert FlashCausalLMBatch.prune(m, []) is m
    assert FlashCausalLMBatch.prune(m, [0, 2]) is None and cache.free_pages == 16
    c = _batch([[1] * 600], 10)
    with pytest.raises(OutOfPages):
        c.allocate_pages(cache)
    assert cache.free_pages == 16  # nothing leaked by the failed allocation


def test_get_indices_to_keep_merge():
    reqs = [pb2.Request(id=i) for i in (2, 3, 5, 8, 13)]
    assert Model.get_indices_to_keep(reqs, [3, 8]) == [0, 2, 4]
    assert Model.get_indices_to_keep(reqs, [1, 2, 13, 99]) == [1, 2, 3]
    assert Model.get_indices_to_keep(reqs, []) == [0, 1, 2, 3, 4]


def test_decode_graph_buckets():
    """flash_causal_lm.graph_bucket: which captured decode step serves a batch of B requests (powers of two to 8, then
    multiples of 8): a batch wandering over 24..32 requests replays two graphs, not nine."""
    from tgis_amd.models.flash_causal_lm import graph_bucket

    assert [graph_bucket(b) for b in (1, 2, 3, 4, 5, 8, 9, 16, 17, 24, 25, 32, 33, 63, 64, 65)] == \
        [1, 2, 4, 4, 8, 8, 16, 16, 24, 24, 32, 32, 40, 64, 64, 72]
    assert all(graph_bucket(b) >= b for b in range(1, 300))
    assert len({graph_bucket(b) for b in range(24, 33)}) == 2
