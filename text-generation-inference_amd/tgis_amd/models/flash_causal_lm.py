"""FlashCausalLMBatch + FlashCausalLM: the batched prefill/decode hot path on the MI355X kernels.

Mirrors models/flash_causal_lm.py of the reference: batch fields and `from_pb` (:28-194), `concatenate`
(:196-285), `prune` (:290-353), `generate_token` (:405-460), `_process_prefill/_decode/_new_tokens`
(:462-588).  Same names, same argument meaning, same returned tuple, same logical bookkeeping
(`cu_seqlens`, `cu_seqlens_q`, `max_seqlen`, `position_ids`, `all_input_ids_tensor`, `input_lengths`).

What is different by design:
  * `past_key_values` is always None: KV lives in the model's PagedKVCache and a batch owns a list of page
    ids per request.  The reference re-concatenates the whole KV tensor on every decode step for B > 1
    (:439-447) and on every concatenate/prune; here those are page-table edits.
  * the decode step (embedding -> layers -> head -> greedy) is replayed from a captured HIP graph keyed by
    (batch size, block-table width); the only host<->device traffic per step is three small input copies
    and, for plain greedy batches, one copy of `[B]` token ids (+ logprobs) instead of B `.item()` syncs
    (reference :546-586 / utils/tokens.py:394).
"""
import logging
import os
import time
from collections import OrderedDict
from dataclasses import dataclass
from operator import itemgetter
from typing import Any, List, Optional, Tuple, Type, Union

import numpy as np
import torch

from tgis_amd import native
from tgis_amd.models.custom_modeling.flash_llama_modeling import KVArgs
from tgis_amd.models.model import Model
from tgis_amd.models.types import Batch, GenerateError
from tgis_amd.pb import generate_pb2
from tgis_amd.utils.graph_segments import SegmentedGraph, no_gc_during_capture
from tgis_amd.utils.kv_cache import PAGE, PagedKVCache
from tgis_amd.utils.token_types import InputTokens, TokenInfo
from tgis_amd.utils.tokens import HeterogeneousNextTokenChooser, get_input_tokens_info, get_token_info

logger = logging.getLogger(__name__)

USE_GRAPHS = os.getenv("TGIS_DISABLE_GRAPHS", "false").lower() not in ("1", "true")


@dataclass
class FlashCausalLMBatch(Batch):
    batch_id: int
    requests: List[generate_pb2.Request]

    # tensors hold the sequences of the batch concatenated: [sum(seq_lengths)] (prefill) / [B] (decode)
    input_ids: Optional[torch.Tensor]
    position_ids: torch.Tensor
    inputs_embeds: Optional[torch.Tensor]
    # cumulative (logical) sequence lengths, and cumulative query lengths (decode only)
    cu_seqlens: torch.Tensor
    cu_seqlens_q: Optional[torch.Tensor]
    # kept for the servicer's clean_attribute("past_key_values") call; always None (paged cache)
    past_key_values: Optional[torch.Tensor]
    # maximum of the input lengths across the batch (including prefix)
    max_seqlen: int

    all_input_ids_tensor: torch.Tensor
    input_lengths: List[int]
    # (truncated) input length + prefix length + max output tokens: sizes all_input_ids_tensor and the pages
    total_lengths: List[int]
    pad_token_id: int

    next_token_chooser: HeterogeneousNextTokenChooser

    # paged-KV ownership: page ids per request, filled by the prefill generate_token
    kv_cache: Optional[PagedKVCache] = None
    pages: Optional[List[List[int]]] = None
    block_tables: Optional[torch.Tensor] = None

    def get_id(self) -> int:
        return self.batch_id

    def __len__(self):
        return len(self.requests)

    # ---- page ownership --------------------------------------------------------------------------
    def allocate_pages(self, kv_cache: PagedKVCache):
        """Pages for the prompt and the first generated token.  The cache then grows one page at a time as sequences
        cross page boundaries (`grow_pages`), like the reference's KV grows with the tokens produced — the router's
        batch-weight model (router/src/batch_types.rs:46-118) counts tokens present, not max_output_length."""
        assert self.pages is None
        need = [PagedKVCache.pages_for(n + 1) for n in self.input_lengths]
        flat = kv_cache.alloc(sum(need))  # raises OutOfPages before anything is taken
        self.kv_cache = kv_cache
        # page-major: page p of every sequence, then page p + 1 (kv_cache.py: the pages the decode blocks read at the same
        # time are then neighbours in the pool)
        self.pages, it = [[] for _ in need], iter(flat)
        for p in range(max(need, default=0)):
            for i in range(len(need)):
                if p < need[i]:
                    self.pages[i].append(next(it))
        self._rebuild_block_tables()

    def grow_pages(self):
        """Before a decode step: every sequence owns the page its next token (position input_length - 1) lands on."""
        short = [i for i, (p, n) in enumerate(zip(self.pages, self.input_lengths)) if len(p) * PAGE < n]
        if not short:
            return
        flat = self.kv_cache.alloc(len(short))  # all or nothing: OutOfPages leaves the batch as it was
        for i, pg in zip(short, flat):
            self.pages[i].append(pg)
        width = self.block_tables.shape[1]
        if max(len(self.pages[i]) for i in short) > width or getattr(self, "_bt_host", None) is None:
            self._rebuild_block_tables()
        else:  # edit the host copy and upload the few KB again (the same copy the prefill made: nothing new to load)
            for i in short:
                self._bt_host[i, len(self.pages[i]) - 1] = self.pages[i][-1]
            self.block_tables = torch.from_numpy(self._bt_host).to(self.block_tables.device, non_blocking=True)

    def _rebuild_block_tables(self):
        # as wide as the longest sequence can ever get (pages themselves are taken lazily): the decode graph of a batch
        # is keyed by (size, table width), so the width must not creep up while the batch generates
        width = max(max(len(p) for p in self.pages), max(PagedKVCache.pages_for(t) for t in self.total_lengths))
        width = (width + 7) // 8 * 8  # few distinct widths -> few captured graphs
        bt = np.zeros((len(self.pages), width), dtype=np.int32)
        for i, p in enumerate(self.pages):
            bt[i, :len(p)] = p
        self._bt_host = bt
        self.block_tables = torch.from_numpy(bt).to(self.cu_seqlens.device, non_blocking=True)

    def release(self):
        if self.pages is not None and self.kv_cache is not None:
            for p in self.pages:
                self.kv_cache.free(p)
        self.pages = None
        self.block_tables = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    # ---- construction ---------------------------------------------------------------------------------
    @classmethod
    def from_pb(cls, pb: generate_pb2.Batch, tokenizer, dtype: torch.dtype, device: torch.device,
                embeddings_lookup: Optional, prefix_cache: Optional, use_position_ids: bool = True,
                ) -> Tuple[Optional["FlashCausalLMBatch"], List[GenerateError]]:
        errors: List[GenerateError] = []
        requests, batch_inputs, prefix_embeds_by_index = [], [], {}
        input_lengths, total_lengths = [], []
        cu_seqlens = [0]
        for r in pb.requests:
            input_length = r.input_length
            if r.prefix_id:
                try:
                    prefix_embeds = prefix_cache.get(r.prefix_id)
                except Exception:
                    message = f"Prefix lookup error for request #{r.id}, prefix id {r.prefix_id}"
                    logging.error(message)
                    errors.append(GenerateError(request_id=r.id, message=message))
                    continue  # the request is left out of the batch
                prefix_embeds_by_index[len(requests)] = prefix_embeds
                input_length += prefix_embeds.shape[0]  # input_lengths include the prefix
            requests.append(r)
            batch_inputs.append(r.inputs)
            input_lengths.append(input_length)
            total_lengths.append(input_length + r.max_output_length)
            cu_seqlens.append(cu_seqlens[-1] + input_length)
        if not requests:
            return None, errors
        max_seqlen = max(input_lengths)

        # no padding: sequences are concatenated across the batch
        tokenized = tokenizer(batch_inputs, truncation=True, max_length=max_seqlen,
                              return_token_type_ids=False)["input_ids"]
        all_input_ids_tensor = torch.full((len(requests), max(total_lengths)), tokenizer.pad_token_id,
                                          dtype=torch.int64, device=device)
        input_ids, position_ids, chooser_params, return_logprobs = [], [], [], []
        for i, (r, toks, input_length) in enumerate(zip(requests, tokenized, input_lengths)):
            if r.truncate:
                toks = toks[-r.input_length:]
                if getattr(tokenizer, "add_bos_token", False):
                    toks[0] = tokenizer.bos_token_id  # keep a BOS at the front after left-truncation
            toks = all_input_ids_tensor.new_tensor(toks)
            # a prefix occupies the first (input_length - r.input_length) positions as pad ids
            all_input_ids_tensor[i, input_length - r.input_length:input_length] = toks
            input_ids.append(toks if input_length == r.input_length else all_input_ids_tensor[i, :input_length])
            chooser_params.append(r.parameters)
            return_logprobs.append(r.details.logprobs)
            position_ids.append(torch.arange(0, input_length))
        input_ids = torch.cat(input_ids)

        if prefix_embeds_by_index:
            inputs_embeds = embeddings_lookup(input_ids)
            input_ids = None
            for i, p in prefix_embeds_by_index.items():
                inputs_embeds[cu_seqlens[i]:cu_seqlens[i] + p.shape[0], :] = p
        else:
            inputs_embeds = None

        next_token_chooser = HeterogeneousNextTokenChooser.from_pb(
            pb=chooser_params,
            model_eos_token_id=getattr(tokenizer, "model_eos_token_id", tokenizer.eos_token_id),
            model_pad_token_id=tokenizer.pad_token_id,
            return_logprobs=return_logprobs, dtype=torch.float32, device=device)

        return cls(
            batch_id=pb.id, requests=requests, input_ids=input_ids, inputs_embeds=inputs_embeds,
            position_ids=torch.cat(position_ids).to(device, non_blocking=True),
            cu_seqlens=torch.tensor(cu_seqlens, dtype=torch.int32, device=device), cu_seqlens_q=None,
            max_seqlen=max_seqlen, past_key_values=None, input_lengths=input_lengths,
            total_lengths=total_lengths, all_input_ids_tensor=all_input_ids_tensor,
            next_token_chooser=next_token_chooser, pad_token_id=tokenizer.pad_token_id,
        ), errors

    @classmethod
    def concatenate(cls, batches: List["FlashCausalLMBatch"]) -> "FlashCausalLMBatch":
        first = batches[0]
        device = first.cu_seqlens_q.device
        requests, input_lengths, total_lengths, pages = [], [], [], []
        chooser_params, ntc_current_tokens, ntc_samplings, ntc_return_logprobs = [], [], [], []
        input_ids, position_ids = [], []
        cu_seqlens = [torch.tensor([0], dtype=torch.int32, device=device)]
        new_batch_size = sum(len(b) for b in batches)
        max_total_length = max(t for b in batches for t in b.total_lengths)
        all_input_ids_tensor = first.all_input_ids_tensor.new_full((new_batch_size, max_total_length),
                                                                   first.pad_token_id)
        cumulative_length = torch.tensor(0, device=device)
        start, max_seqlen = 0, 0
        for batch in batches:
            requests.extend(batch.requests)
            input_lengths.extend(batch.input_lengths)
            total_lengths.extend(batch.total_lengths)
            chooser_params.extend(r.parameters for r in batch.requests)
            ntc_current_tokens.extend(batch.next_token_chooser.current_tokens)
            ntc_samplings.extend(batch.next_token_chooser.samplings)
            ntc_return_logprobs.extend(batch.next_token_chooser.return_logprobs)
            cu_seqlens.append(batch.cu_seqlens[1:] + cumulative_length)
            input_ids.append(batch.input_ids)
            position_ids.append(batch.position_ids)
            # no KV bytes move (reference: torch.cat of the pasts); ownership of the pages changes hands below, once
            # the merged batch exists — an exception before that leaves every page with its source batch
            pages.extend(batch.pages)
            end = start + len(batch)
            all_input_ids_tensor[start:end, :batch.all_input_ids_tensor.shape[1]] = batch.all_input_ids_tensor
            start = end
            max_seqlen = max(max_seqlen, batch.max_seqlen)
            cumulative_length += batch.cu_seqlens[-1]

        ntc0 = first.next_token_chooser
        next_token_chooser = HeterogeneousNextTokenChooser.from_pb(
            pb=chooser_params, model_eos_token_id=ntc0.eos_token_id, model_pad_token_id=ntc0.pad_token_id,
            return_logprobs=ntc_return_logprobs, dtype=ntc0.dtype, device=ntc0.device,
            samplings=ntc_samplings, current_tokens=ntc_current_tokens)

        merged = FlashCausalLMBatch(
            batch_id=first.batch_id, requests=requests, input_ids=torch.cat(input_ids), inputs_embeds=None,
            position_ids=torch.cat(position_ids), cu_seqlens=torch.cat(cu_seqlens),
            cu_seqlens_q=torch.arange(len(requests) + 1, device=device, dtype=torch.int32),
            max_seqlen=max_seqlen, past_key_values=None, input_lengths=input_lengths,
            total_lengths=total_lengths, all_input_ids_tensor=all_input_ids_tensor,
            next_token_chooser=next_token_chooser, pad_token_id=first.pad_token_id,
            kv_cache=first.kv_cache, pages=None)
        merged.pages = pages
        try:
            merged._rebuild_block_tables()
        except BaseException:
            merged.pages = None  # the sources still own them
            raise
        for batch in batches:
            batch.pages = None
            batch.block_tables = None
        return merged

    @classmethod
    def prune(cls, batch: "FlashCausalLMBatch", completed_ids: List[int]) -> Optional["FlashCausalLMBatch"]:
        """Drop completed requests; their pages go back to the pool."""
        if not completed_ids:
            return batch
        keep_indices = Model.get_indices_to_keep(batch.requests, completed_ids)
        new_size = len(keep_indices)
        if new_size == 0:
            batch.release()
            return None
        keep = set(keep_indices)
        for i, p in enumerate(batch.pages):
            if i not in keep:
                batch.kv_cache.free(p)
        pick = (lambda l: [l[i] for i in keep_indices])
        batch.pages = pick(batch.pages)
        batch.input_lengths = pick(batch.input_lengths)
        batch.total_lengths = pick(batch.total_lengths)
        batch.requests = pick(batch.requests)
        batch.next_token_chooser = batch.next_token_chooser.filter(keep_indices)
        batch.max_seqlen = max(batch.input_lengths)
        batch.input_ids = batch.input_ids[keep_indices]
        batch.position_ids = batch.position_ids[keep_indices]
        batch.all_input_ids_tensor = batch.all_input_ids_tensor[keep_indices, :max(batch.total_lengths)]
        if new_size == 1:
            batch.cu_seqlens = batch.cu_seqlens.new_tensor([0, batch.input_lengths[0]])
        else:
            # logical slot layout after re-packing: every kept sequence followed by its free slot
            cu = batch.cu_seqlens[:new_size + 1].clone()
            cu[1:] = batch.position_ids
            cu[1:].add_(1)
            batch.cu_seqlens = torch.cumsum(cu, dim=0, dtype=torch.int32)
        batch.cu_seqlens_q = batch.cu_seqlens_q[:new_size + 1]
        batch._rebuild_block_tables()
        return batch


# page-wise cache writes in prefill (tgis_rope_kv_write_prefill); "false" keeps the per-token kernel for A/B checks
FRESH_PREFILL_KV = os.getenv("TGIS_PREFILL_KV", "true").lower() not in ("0", "false")


# A captured decode step serves every batch size of its BUCKET (round 6): the rows past the batch are inactive — position 0,
# every table entry the pool's null page (utils/kv_cache.py), their token written there and attended to alone, their ids
# never read.  A router that lets a batch wander over 24..32 requests then replays ONE graph instead of capturing nine (a
# capture is a warm-up step + the capture itself: tens of ms in the middle of serving; bench.py --churn reports them).
GRAPH_BUCKETS = os.getenv("TGIS_GRAPH_BUCKETS", "true").lower() not in ("0", "false")


def graph_bucket(B: int) -> int:
    """Rows of the decode graph that serves B requests: powers of two up to 8, then multiples of 8 (the GEMMs cost the same
    up to 32 rows and from 33 to 64; an inactive row costs the attention one page)."""
    if not GRAPH_BUCKETS:
        return B
    if B <= 8:
        return 1 << (B - 1).bit_length()
    return (B + 7) // 8 * 8


class _DecodeGraph:
    """Static buffers + captured HIP graph of one decode step for a (batch-size bucket, table width) pair."""

    def __init__(self, lm: "FlashCausalLM", B: int, width: int):
        dev = lm.device
        self.rows = B
        self.active = 0  # rows [0, active) hold a batch's sequences, the rest are inactive
        self.input_ids = torch.zeros(B, dtype=torch.int64, device=dev)
        self.positions = torch.zeros(B, dtype=torch.int32, device=dev)
        self.block_tables = torch.full((B, width), lm.kv_cache.null_page, dtype=torch.int32, device=dev)
        self.slots = torch.zeros(B, dtype=torch.int32, device=dev)
        self.ctx = torch.ones(B, dtype=torch.int32, device=dev)
        self.cu_q = torch.arange(B + 1, dtype=torch.int32, device=dev)
        self.max_ctx = width * PAGE
        self.num_splits = native.attn_num_splits(B, lm.num_kv_heads, lm.num_heads, 1, self.max_ctx)
        self.lm = lm
        self.graph = None
        self.logits = self.ids = self.logprobs = None
        # greedy ids (int64) and their logprobs (f32) in ONE device buffer: one copy to the (pinned) host mirror per step
        self.out_buf = torch.zeros(B * 12, dtype=torch.uint8, device=dev)
        self.ids_buf = self.out_buf[:B * 8].view(torch.int64)
        self.lps_buf = self.out_buf[B * 8:].view(torch.float32)
        self.argmax_scratch = native.argmax_scratch(B, dev)
        self.host_buf = torch.zeros(B * 12, dtype=torch.uint8).pin_memory()
        self.host_ready = torch.cuda.Event()
        # whose next-step inputs the static buffers already hold (tgis_decode_advance wrote them): identity of the
        # batch's tensors, so that a pruned / concatenated / other batch always stages its own
        self.staged_ids = self.staged_pos = self.staged_bt = None

    def fetch_greedy(self):
        """ids and logprobs of the step that just ran, as host lists: one device->host copy, one wait."""
        self.host_buf.copy_(self.out_buf, non_blocking=True)
        self.host_ready.record()
        return self._read_host

    def _read_host(self, want_logprobs: bool):
        self.host_ready.synchronize()
        B = self.ids_buf.numel()
        ids = self.host_buf[:B * 8].view(torch.int64).tolist()
        return ids, (self.host_buf[B * 8:].view(torch.float32).tolist() if want_logprobs else None)

    def _step(self):
        lm = self.lm
        native.decode_slots(self.positions, self.block_tables, self.slots, self.ctx)
        kv = KVArgs(cache=lm.kv_cache, block_tables=self.block_tables, ctx_lens=self.ctx, slots=self.slots,
                    max_q_len=1, max_ctx=self.max_ctx, num_splits=self.num_splits)
        logits = lm.model.forward(self.input_ids, self.positions, self.cu_q, self.max_ctx, None, kv)
        ids, lps = native.argmax_logprob(logits, ids_out=self.ids_buf, logprob_out=self.lps_buf, scratch=self.argmax_scratch)
        return logits, ids, lps

    def run(self, input_ids, position_ids, block_tables):
        """One decode step of a batch of n <= rows sequences; returns (logits, ids, logprobs) of its n rows."""
        n = input_ids.numel()
        if n < self.active:  # rows a larger batch used before: inactive again
            self.positions[n:self.active].zero_()
            self.input_ids[n:self.active].zero_()
            self.block_tables[n:self.active].fill_(self.lm.kv_cache.null_page)
        self.active = n
        if input_ids is not self.staged_ids or position_ids is not self.staged_pos:
            self.input_ids[:n].copy_(input_ids, non_blocking=True)
            self.positions[:n].copy_(position_ids, non_blocking=True)
        if block_tables is not self.staged_bt:
            self.block_tables[:n].copy_(block_tables, non_blocking=True)
            self.staged_bt = block_tables
        self.staged_ids = self.staged_pos = None
        logits, ids, lps = self._run()
        return (logits, ids, lps) if n == self.rows else (logits[:n], ids[:n], lps[:n])

    def _run(self):
        if not self.lm.use_graphs:
            return self._step()
        if self.graph is None:
            t_capture = time.perf_counter()
            # warm-up: sizes the workspaces, builds rope tables and (tp > 1) initialises the RCCL communicators outside
            # the capture
            self._step()
            torch.cuda.current_stream().synchronize()
            g = None
            if self.lm.resolve_graph_mode() == "full":
                g = torch.cuda.CUDAGraph()
                # tp > 1: RCCL's proxy and watchdog threads may call the runtime while this thread captures
                kw = {"capture_error_mode": "thread_local"} if self.lm.tp_world > 1 else {}
                ok = True
                try:
                    with no_gc_during_capture(), torch.cuda.graph(g, pool=self.lm.graph_pool, **kw):
                        self.logits, self.ids, self.logprobs = self._step()
                except Exception as exc:
                    if self.lm.tp_world == 1:
                        raise
                    logger.warning("capturing the tensor-parallel step with RCCL inside failed (%s)", exc)
                    native.clear_error()
                    ok = False
                if self.lm.tp_world > 1 and not self.lm.all_ranks_agree(ok):
                    # one rank failing is every rank's failure: all of them leave `full` together, or their collective
                    # sequences would diverge (a capture issues no collective, so nobody is waiting inside one here)
                    self.lm.graph_mode = "segments"
                    g = None
            if g is None:
                if not any(d.graph is not None for d in self.lm._graphs.values()):
                    # no captured graph holds the pool (e.g. the full capture above failed and took the pool's only
                    # graph with it): the allocator has retired it, start a new one
                    self.lm.graph_pool = torch.cuda.graph_pool_handle()
                g = SegmentedGraph(self.lm.device, pool=self.lm.graph_pool)
                try:
                    self.logits, self.ids, self.logprobs = g.record(self._step)
                except Exception as exc:  # keep serving: the eager step needs nothing the capture set up
                    logger.warning("segmented capture of the decode step failed (%s); running eagerly", exc)
                    native.clear_error()
                    self.lm.use_graphs = False
                    return self._step()
            self.graph = g
            # (rows, table width, host ms of warm-up step + capture): what a new (bucket, width) pair costs a serving step
            self.lm.graph_captures.append((self.rows, self.block_tables.shape[1], (time.perf_counter() - t_capture) * 1e3))
        self.graph.replay()
        return self.logits, self.ids, self.logprobs


class FlashCausalLM(Model):
    def __init__(self, model_name: str, revision: Optional[str], deployment_framework: str, dtype: torch.dtype,
                 quantize: Optional[str], model_config: Union[Any] = None, auto_model_class=None,
                 max_sequence_length: Optional[int] = None, engine=None, kv_cache_pages: Optional[int] = None):
        if not torch.cuda.is_available():
            raise NotImplementedError("FlashCausalLM is only available on GPU")
        if engine is None:
            from tgis_amd.inference_engine import get_inference_engine_class
            from tgis_amd.utils.hub import get_model_path

            model_path = get_model_path(model_name, revision)
            engine = get_inference_engine_class(deployment_framework)(
                model_path, auto_model_class, dtype, quantize, model_config, max_sequence_length)
        super().__init__(engine, dtype, max_sequence_length)
        self.use_position_ids = True
        tok = self.tokenizer
        if tok is not None:
            if getattr(self.config, "pad_token_id", None) is not None:
                tok.pad_token_id = self.config.pad_token_id
            elif tok.pad_token_id is None:
                if getattr(self.config, "eos_token_id", None) is not None:
                    tok.pad_token_id = self.config.eos_token_id
                elif tok.eos_token_id is not None:
                    tok.pad_token_id = tok.eos_token_id
                else:
                    tok.add_special_tokens({"pad_token": "[PAD]"})

        inner = self.model.model  # FlashLlamaModel / FlashSantacoderModel
        self.num_heads = inner.num_heads
        self.num_kv_heads = inner.num_key_value_heads
        self.head_size = inner.head_size
        self.num_layers = len(inner.layers)
        if hasattr(self.model, "post_init"):
            self.model.post_init()
        if kv_cache_pages is None:
            kv_cache_pages = self._default_kv_pages()
        # Tensor parallel: every rank must hold the SAME number of pages.  Each rank sizes its pool from its own free
        # memory, and `grow_pages` raises OutOfPages before a decode step: a rank with fewer pages would leave the step
        # while the others enter its all-reduces (a hang instead of RESOURCE_EXHAUSTED), and the memory model a rank
        # reports to the router would differ from its peers'.  The smallest pool decides.
        kv_cache_pages = self._agree_on_min(kv_cache_pages, engine)
        self.kv_cache = PagedKVCache(self.num_layers, self.num_kv_heads, self.head_size, kv_cache_pages, dtype,
                                     self.device)
        # tp > 1, TGIS_TP_GRAPHS = auto (default) | full | segments | false:
        #   full      one graph per step with the RCCL all-reduces / all-gather inside it (RCCL is capture-aware);
        #   segments  a chain of graphs with the collectives launched between them (utils/graph_segments.py);
        #   false     every kernel launched eagerly (host-bound: 4.4 ms/step whatever the shard size);
        #   auto      full if a one-collective probe graph captures, replays and reduces correctly on every rank of an
        #             RCCL group, else segments.
        # One rank's step at TP=8 shapes, collectives on a world-size-1 RCCL group (tools/tp_segments_rccl1.py):
        # 2.0 ms full, 3.2 ms segments, 4.4 ms eager.
        tp = engine.world_size if hasattr(engine, "world_size") else 1
        self.tp_world = tp
        self.process_group = getattr(engine, "process_group", None)
        tp_mode = os.getenv("TGIS_TP_GRAPHS", "auto").lower()
        if tp == 1 or tp_mode in ("full", "1", "true"):
            self.graph_mode = "full"
        elif tp_mode in ("auto", "segments"):
            self.graph_mode = tp_mode
        else:
            self.graph_mode = None
        self.use_graphs = USE_GRAPHS and self.graph_mode is not None
        if tp > 1 and isinstance(self.process_group, torch.distributed.ProcessGroup):
            # requests without a seed must still draw the same tokens on every rank
            from tgis_amd.utils import tokens

            base = tokens.seed_base()
            t = torch.tensor([base >> 32, base & 0xFFFFFFFF], dtype=torch.int64, device=self.device)
            torch.distributed.broadcast(t, src=0, group=self.process_group)
            hi, lo = t.tolist()
            tokens.set_seed_base((hi << 32) | lo)
        # captured decode steps, least recently used first; they share one memory pool (a step's intermediates are dead
        # once it has run, and its outputs are consumed before the next replay), and the number kept is bounded
        self._graphs = OrderedDict()
        self.graph_captures: List[Tuple[int, int, float]] = []
        self.max_graphs = int(os.getenv("TGIS_MAX_DECODE_GRAPHS", "48"))
        self.graph_pool = torch.cuda.graph_pool_handle() if self.use_graphs else None

    def resolve_graph_mode(self) -> str:
        """"full" or "segments"; `auto` is settled once, identically on every rank."""
        if self.graph_mode == "auto":
            try:
                works = self._collective_capture_works()
            except Exception as exc:  # an unusable probe must not take the server down: segments need no capture of RCCL
                logger.warning("probing RCCL graph capture failed (%s)", exc)
                works = False
            self.graph_mode = "full" if works else "segments"
            logger.info("tensor-parallel decode graphs: %s", self.graph_mode)
        return self.graph_mode

    def all_ranks_agree(self, ok: bool) -> bool:
        """True iff `ok` holds on every rank of the tensor-parallel group (one small all-reduce, outside any capture)."""
        pg = self.process_group
        if self.tp_world == 1 or not isinstance(pg, torch.distributed.ProcessGroup):
            return ok
        dev = self.device if torch.distributed.get_backend(pg) == "nccl" else "cpu"
        flag = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN, group=pg)
        return bool(flag.item())

    def _collective_capture_works(self) -> bool:
        pg = self.process_group
        if not isinstance(pg, torch.distributed.ProcessGroup) or torch.distributed.get_backend(pg) != "nccl":
            return False  # host-mediated collectives synchronise: they can never be inside a capture
        world = pg.size()
        t = torch.ones(1024, device=self.device, dtype=torch.float32)
        ok = True
        try:
            torch.distributed.all_reduce(t, group=pg)  # communicator up before any capture
            torch.cuda.synchronize(self.device)
            t.fill_(1.0)
            g = torch.cuda.CUDAGraph()
            with no_gc_during_capture(), torch.cuda.graph(g, capture_error_mode="thread_local"):
                torch.distributed.all_reduce(t, group=pg)
            g.replay()
            g.replay()
            torch.cuda.synchronize(self.device)
            ok = bool((t == float(world * world)).all().item())
        except Exception as exc:
            logger.warning("RCCL inside a captured graph is not usable here (%s)", exc)
            native.clear_error()
            ok = False
        flag = torch.tensor([1 if ok else 0], device=self.device, dtype=torch.int32)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN, group=pg)
        return bool(flag.item())

    def _agree_on_min(self, value: int, engine) -> int:
        """min of `value` over the tensor-parallel group (all-reduce MIN); the value itself on one rank."""
        pg = getattr(engine, "process_group", None)
        world = engine.world_size if hasattr(engine, "world_size") else 1
        if world == 1 or not isinstance(pg, torch.distributed.ProcessGroup):
            return int(value)
        dev = self.device if torch.distributed.get_backend(pg) == "nccl" else torch.device("cpu")
        t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN, group=pg)
        return int(t.item())

    def _default_kv_pages(self) -> int:
        free, _total = torch.cuda.mem_get_info(self.device)
        frac = float(os.getenv("TGIS_KV_CACHE_FRACTION", "0.85"))
        per_page = self.num_layers * 2 * self.num_kv_heads * PAGE * self.head_size * 2
        return max(64, int(free * frac) // per_page)

    @property
    def batch_type(self) -> Type[FlashCausalLMBatch]:
        return FlashCausalLMBatch

    # ---- the hot path -------------------------------------------------------------------------------------
    def generate_token(self, batch: FlashCausalLMBatch, first: bool = False, for_concat: bool = False,
                       ) -> Tuple[List[TokenInfo], Optional[List[InputTokens]], List[GenerateError], int]:
        start_time = time.time_ns()
        if first:
            out, fused = self._prefill_forward(batch), None
        else:
            out, fused = self._decode_forward(batch)
        forward_time_ns = time.time_ns() - start_time

        if first:
            generated_tokens, input_token_infos, decode_errors = self._process_prefill(batch, out)
        else:
            generated_tokens, decode_errors = self._process_decode(batch, out, fused)
            input_token_infos = None

        # logical slot bookkeeping of the reference: one more slot per sequence (:457-458); a decode step has done the
        # addition on the device already (tgis_decode_advance)
        if first:
            batch.cu_seqlens.add_(batch.cu_seqlens_q)
        batch.max_seqlen += 1
        return generated_tokens, input_token_infos, decode_errors, forward_time_ns

    def _prefill_forward(self, batch: FlashCausalLMBatch):
        batch.allocate_pages(self.kv_cache)
        lens = batch.input_lengths
        slots = np.concatenate([
            np.asarray(p, dtype=np.int64)[np.arange(l) // PAGE] * PAGE + np.arange(l) % PAGE
            for p, l in zip(batch.pages, lens)]).astype(np.int32)
        dev = self.device
        kv = KVArgs(cache=self.kv_cache, block_tables=batch.block_tables,
                    ctx_lens=torch.tensor(lens, dtype=torch.int32, device=dev),
                    slots=torch.from_numpy(slots).to(dev, non_blocking=True),
                    max_q_len=max(lens), max_ctx=max(lens), num_splits=1, fresh_prefill=FRESH_PREFILL_KV)
        self._need_all_logits = any(r.details.input_toks for r in batch.requests)
        lm_head_indices = None if self._need_all_logits else (batch.cu_seqlens[1:] - 1).long()
        return self.model.forward(batch.input_ids, batch.position_ids.to(torch.int32), batch.cu_seqlens,
                                  batch.max_seqlen, batch.inputs_embeds, kv, lm_head_indices)

    def _decode_forward(self, batch: FlashCausalLMBatch):
        batch.grow_pages()
        key = (graph_bucket(len(batch)), batch.block_tables.shape[1])
        g = self._graphs.get(key)
        if g is None:
            while len(self._graphs) >= self.max_graphs:
                self._graphs.popitem(last=False)
            if not self._graphs and self.graph_pool is not None:
                # the allocator retires a pool with its last graph: captures that follow start a new one
                self.graph_pool = torch.cuda.graph_pool_handle()
            g = self._graphs[key] = _DecodeGraph(self, *key)
        else:
            self._graphs.move_to_end(key)
        logits, ids, lps = g.run(batch.input_ids, batch.position_ids, batch.block_tables)
        return logits, (ids, lps, g)

    def _process_prefill(self, batch: FlashCausalLMBatch, out):
        generated_tokens: List[TokenInfo] = []
        input_token_infos: List[InputTokens] = []
        decode_errors: List[GenerateError] = []
        # position ids of the first generated token, set before input lengths are incremented
        batch.position_ids = batch.position_ids.new_tensor(batch.input_lengths)
        batch.input_ids = self._process_new_tokens(batch, out, generated_tokens, decode_errors, input_token_infos,
                                                   True, None)
        batch.inputs_embeds = None
        batch.cu_seqlens_q = torch.arange(len(batch) + 1, device=self.device, dtype=torch.int32)
        return generated_tokens, input_token_infos, decode_errors

    def _process_decode(self, batch: FlashCausalLMBatch, out, fused):
        generated_tokens: List[TokenInfo] = []
        decode_errors: List[GenerateError] = []
        # position_ids += 1, the scatter into all_input_ids and cu_seqlens += cu_seqlens_q happen in _process_new_tokens
        batch.input_ids = self._process_new_tokens(batch, out, generated_tokens, decode_errors, None, False, fused)
        return generated_tokens, decode_errors

    def _process_new_tokens(self, batch: FlashCausalLMBatch, out, generated_tokens: List[TokenInfo],
                            decode_errors: List[GenerateError], input_token_infos: Optional[List[InputTokens]],
                            prefill: bool, fused):
        if prefill and self._need_all_logits:
            logits = out[batch.cu_seqlens[1:] - 1, :]  # out is [sum(lengths), vocab]
        else:
            logits = out  # already one row per request

        ntc = batch.next_token_chooser
        simple = ntc.is_plain_greedy and not any(r.details.top_n_toks or r.details.ranks for r in batch.requests)
        graph = fused[2] if fused is not None else None
        read_host = None
        if simple:
            # one kernel (already part of the decode graph), one device->host copy for the whole batch
            next_token_ids, next_logprobs = fused[:2] if fused is not None else ntc.choose_greedy_fused(logits)
            if graph is not None:
                read_host = graph.fetch_greedy()  # the copy is on its way while the bookkeeping launch below runs
        else:
            # EOS mask / length penalty, repetition penalty, warpers, argmax or draw, log-softmax at the chosen id:
            # one launch (tgis_warp_sample) and, below, one device->host copy for the whole batch
            next_token_ids, next_logprobs, lse, next_token_scores = ntc.choose_fused(
                batch.all_input_ids_tensor[:, :batch.max_seqlen], logits)

        if prefill:
            batch.all_input_ids_tensor.scatter_(dim=1, index=batch.position_ids[:, None], src=next_token_ids[:, None])
        else:
            # reference :499 `position_ids += 1`, :533 the scatter, :457 `cu_seqlens.add_`, and the copy of the ids out of
            # the buffer the next step overwrites — one launch, which also leaves the next step's inputs in the static
            # buffers of the graph that will run it
            next_token_ids = native.decode_advance(
                next_token_ids, batch.position_ids, batch.all_input_ids_tensor, batch.cu_seqlens, batch.cu_seqlens_q,
                stage_ids=graph.input_ids[:len(batch)] if graph is not None else None,
                stage_positions=graph.positions[:len(batch)] if graph is not None else None)
            if graph is not None:
                graph.staged_ids, graph.staged_pos = next_token_ids, batch.position_ids

        if read_host is not None:
            ids_host, lps_host = read_host(any(ntc.return_logprobs))
        else:
            ids_host = next_token_ids.tolist()
            lps_host = next_logprobs.tolist() if any(ntc.return_logprobs) else None
        for i, request in enumerate(batch.requests):
            try:
                if not simple and (request.details.top_n_toks or request.details.ranks):
                    # top-n tokens and ranks are read from the warped scores of this row (tokens.py:388-425)
                    row = next_token_scores[i:i + 1]
                    info = get_token_info(request, row, next_token_ids[i:i + 1],
                                          row - lse[i] if request.details.logprobs else None)
                else:
                    info = TokenInfo(request_id=request.id, token_id=ids_host[i])
                    if lps_host is not None and request.details.logprobs:
                        info.logprob = lps_host[i]
                generated_tokens.append(info)
                if prefill and request.details.input_toks:
                    self._append_input_tokens(batch, out, i, request, input_token_infos)
            except Exception as e:
                logging.exception(f"token decoding error for request #{request.id}")
                decode_errors.append(GenerateError(request_id=request.id,
                                                   message=f"Token decoding error: {str(e)}"))
            batch.input_lengths[i] += 1
        return next_token_ids

    @staticmethod
    def _append_input_tokens(batch, out, i, request, input_token_infos):
        start = int(batch.cu_seqlens[i])
        input_length = batch.input_lengths[i]
        # the last position's logits predict the generated token, not an input token
        logits = out[start:start + input_length - 1, :]
        input_token_infos.append(get_input_tokens_info(request, batch.all_input_ids_tensor[i, :input_length], logits))
