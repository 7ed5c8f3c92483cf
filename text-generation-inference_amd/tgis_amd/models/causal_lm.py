"""CausalLMBatch + CausalLM: the padded (rectangular) batch path on a HF `AutoModelForCausalLM`.

This is the reference's plumbing path (models/causal_lm.py: batch fields :21-57, `from_pb` :65-216, `concatenate`
:218-461, `prune` :463-526, `CausalLM.forward` :604-634, `generate_token` :636-739) — BASELINE config 1 (GPT-2 small,
fp32, CPU) and the path the parity oracle was captured from.  It is host code around a library model, on whatever
device the engine chose; it is NOT a fallback of the flash path (FlashCausalLM still refuses to run without the HIP
library).

Geometry every method below relies on — a batch is a right-aligned window with headroom on the right:

    attention_mask        [B, window + headroom]   columns [0, window) = past + current inputs (left-padded per row),
                                                   columns [window, window + headroom) = slots of tokens to come
    all_input_ids_tensor  [B, tokens + headroom]   the same without any alignment padding on the left
    window   = max_sequence_length (+ the one-off pad-to-8 columns of a CUDA prefill)
    headroom = padding_right_offset = the largest number of tokens any row may still generate

Every step moves the boundary one column to the right (`max_sequence_length += 1`, `padding_right_offset -= 1`).
All slicing is done with explicit column indices computed from those two numbers.

KV layouts (`KVLayout`): the standard `[B, heads, T, head_dim]` pair per layer (GPT-2, Llama, GPT-BigCode under
transformers >= 4.4x), the BLOOM-era pair with flattened heads and transposed keys (`[B * heads, head_dim, T]` /
`[B * heads, T, head_dim]`: `KeysDimTransposedCausalLMBatch`, reference :742-748) and the merged multi-query tensor
`[B, T, C]` of older GPT-BigCode (`CombinedKVCausalLMBatch`, :750-756).  Every membership operation below touches the
cache only through the layout's time axis, so the three share one code path (the reference branches per layout:
:337-442, :504-509, :528-545, :722-729)."""
import dataclasses
import inspect
import logging
import os
import time
from dataclasses import dataclass
from typing import Any, List, Optional, Tuple, Type, Union

import torch

from tgis_amd.models.model import Model
from tgis_amd.models.types import Batch, GenerateError
from tgis_amd.pb import generate_pb2
from tgis_amd.utils.token_types import InputTokens, TokenInfo
from tgis_amd.utils.tokens import HeterogeneousNextTokenChooser, get_input_tokens_info, get_token_info

# prefill windows on CUDA are padded on the left to a multiple of 8 columns (models/model.py:24-25 of the reference)
CUDA_PAD_TO_MULT_OF_8 = os.getenv("CUDA_PAD_TO_MULT_OF_8", "true").lower() != "false"

KVLayers = List[Any]  # per layer [keys, values] (each [B, heads, T, head_dim], see KVLayout) or one merged tensor


@dataclass(frozen=True)
class KVLayout:
    """How the library model lays out one layer of its cache (the reference's `keys_head_dim_last` / `merged_kv_cache`
    batch flags, causal_lm.py:54-57, and its `three_dim_pkvs` shape test, :253)."""
    merged: bool = False          # one tensor [B, T, C] per layer instead of a [keys, values] pair
    keys_time_last: bool = False  # keys end in [head_dim, T] instead of [T, head_dim]

    @staticmethod
    def probe(past: KVLayers) -> "KVLayout":
        """From the cache of a one-token forward (:582-586)."""
        if torch.is_tensor(past[0]):
            return KVLayout(merged=True)
        k, v = past[0]
        return KVLayout(keys_time_last=k.shape[-1] != v.shape[-1])

    def slots(self, layer) -> List[torch.Tensor]:
        return [layer] if self.merged else [layer[0], layer[1]]

    def pack(self, tensors: List[torch.Tensor]):
        return tensors[0] if self.merged else tensors

    def time_axis(self, slot: int, t: torch.Tensor) -> int:
        if self.merged:
            return 1
        return t.dim() - 1 if (slot == 0 and self.keys_time_last) else t.dim() - 2

    def by_row(self, t: torch.Tensor, rows: int) -> torch.Tensor:
        """A view whose first axis is the batch row (flattened `[B * heads, ., .]` tensors get their head axis back)."""
        if not self.merged and t.dim() == 3:
            return t.view(rows, -1, *t.shape[-2:])
        return t

    def like(self, t4: torch.Tensor, proto: torch.Tensor) -> torch.Tensor:
        """Back to the model's own rank (`[B * heads, ., .]` if that is what it produced)."""
        if not self.merged and proto.dim() == 3:
            return t4.reshape(-1, *t4.shape[-2:])
        return t4

    def last(self, slot: int, t: torch.Tensor, n: int) -> torch.Tensor:
        """The last `n` cache positions."""
        ax = self.time_axis(slot, t)
        return t.narrow(ax, t.shape[ax] - n, n)


def _right_aligned(dst: torch.Tensor, rows: slice, src: torch.Tensor, right_edge: int) -> None:
    """Copy `src` ([b, w]) into `dst[rows]` so that its last column lands at column `right_edge - 1`."""
    w = src.shape[1]
    dst[rows, right_edge - w:right_edge] = src


@dataclass
class CausalLMBatch(Batch):
    batch_id: int
    requests: List[generate_pb2.Request]

    # model inputs: exactly one of input_ids / inputs_embeds is set during prefill; decode feeds input_ids [B, 1]
    input_ids: Optional[torch.Tensor]
    inputs_embeds: Optional[torch.Tensor]
    attention_mask: torch.Tensor
    position_ids: Optional[torch.Tensor]
    past_key_values: Optional[KVLayers]

    all_input_ids_tensor: torch.Tensor
    input_lengths: List[int]
    next_token_chooser: HeterogeneousNextTokenChooser

    max_sequence_length: int
    padding_right_offset: int
    max_remaining_tokens: List[int]
    pad_token_id: int
    kv_layout: KVLayout = KVLayout()

    # the reference's names for the two layout flags (causal_lm.py:54-57)
    # (assignable, as the reference's plain attributes are: its KeysDimTransposed / CombinedKV from_pb set them, and so
    # may an external subclass; kv_layout stays the one source of truth)
    @property
    def keys_head_dim_last(self) -> bool:
        return not self.kv_layout.keys_time_last

    @keys_head_dim_last.setter
    def keys_head_dim_last(self, value: bool) -> None:
        self.kv_layout = dataclasses.replace(self.kv_layout, keys_time_last=not value)

    @property
    def merged_kv_cache(self) -> bool:
        return self.kv_layout.merged

    @merged_kv_cache.setter
    def merged_kv_cache(self, value: bool) -> None:
        self.kv_layout = dataclasses.replace(self.kv_layout, merged=bool(value))

    def get_id(self) -> int:
        return self.batch_id

    def __len__(self) -> int:
        return len(self.requests)

    # ---- geometry -----------------------------------------------------------------------------------------------
    @property
    def window_end(self) -> int:
        """First headroom column of attention_mask."""
        return self.attention_mask.shape[1] - self.padding_right_offset

    @property
    def tokens_end(self) -> int:
        """First headroom column of all_input_ids_tensor."""
        return self.all_input_ids_tensor.shape[1] - self.padding_right_offset

    # ---- construction -------------------------------------------------------------------------------------------
    @classmethod
    def from_pb(cls, pb: generate_pb2.Batch, tokenizer, dtype: torch.dtype, device: torch.device,
                embeddings_lookup: Optional, prefix_cache: Optional, use_position_ids: bool = False,
                ) -> Tuple[Optional["CausalLMBatch"], List[GenerateError]]:
        errors: List[GenerateError] = []
        kept, prefixes = [], {}
        for r in pb.requests:
            if r.prefix_id:
                try:
                    prefixes[len(kept)] = prefix_cache.get(r.prefix_id)
                except Exception:
                    logging.exception(f"Prefix lookup error for request #{r.id}, prefix id {r.prefix_id}")
                    errors.append(GenerateError(request_id=r.id,
                                                message=f"Error retrieving prompt prefix '{r.prefix_id}'"))
                    continue  # the request is left out of the batch
            kept.append(r)
        if not kept:
            return None, errors

        B = len(kept)
        prefix_len = [prefixes[i].shape[0] if i in prefixes else 0 for i in range(B)]
        input_lengths = [r.input_length + p for r, p in zip(kept, prefix_len)]  # a prefix counts as input
        remaining = [r.max_output_length for r in kept]
        tokens = max(input_lengths)
        headroom = max(remaining)

        chooser = HeterogeneousNextTokenChooser.from_pb(
            pb=[r.parameters for r in kept],
            model_eos_token_id=getattr(tokenizer, "model_eos_token_id", tokenizer.eos_token_id),
            model_pad_token_id=tokenizer.pad_token_id,
            return_logprobs=[r.details.logprobs for r in kept], dtype=dtype, device=device)

        align = 0
        if device.type == "cuda" and CUDA_PAD_TO_MULT_OF_8 and tokens % 8:
            align = 8 - tokens % 8
        window = tokens + align
        enc = tokenizer([r.inputs for r in kept], return_tensors="pt", padding="max_length", truncation=True,
                        max_length=window, return_token_type_ids=False).to(device)
        ids = enc["input_ids"]  # [B, window], left-padded / left-truncated by the tokenizer

        mask = ids.new_zeros((B, window + headroom))
        mask[:, :window] = enc["attention_mask"]
        # a request with `truncate` keeps only its last input_length tokens (the text itself is never shortened)
        bos_first = getattr(tokenizer, "add_bos_token", False)
        for i, r in enumerate(kept):
            if r.truncate:
                first_kept = window - r.input_length
                mask[i, :first_kept] = 0
                ids[i, :first_kept] = tokenizer.pad_token_id
                if bos_first:
                    ids[i, first_kept] = tokenizer.bos_token_id

        all_ids = ids.new_full((B, tokens + headroom), tokenizer.pad_token_id)
        all_ids[:, :tokens] = ids[:, align:]

        inputs_embeds = None
        if prefixes:
            inputs_embeds = embeddings_lookup(ids)
            for i, p in prefixes.items():
                start = window - input_lengths[i]
                inputs_embeds[i, start:start + p.shape[0]] = p
                mask[i, start:window] = 1  # the virtual prefix tokens are attended to

        position_ids = None
        if use_position_ids:
            live = mask[:, :window]
            position_ids = live.cumsum(-1) - 1
            position_ids.masked_fill_(live == 0, 1)

        return cls(batch_id=pb.id, requests=kept, input_ids=ids, inputs_embeds=inputs_embeds, attention_mask=mask,
                   position_ids=position_ids, past_key_values=None, all_input_ids_tensor=all_ids,
                   input_lengths=input_lengths, next_token_chooser=chooser, max_sequence_length=tokens,
                   padding_right_offset=headroom, max_remaining_tokens=remaining,
                   pad_token_id=tokenizer.pad_token_id), errors

    # ---- membership ---------------------------------------------------------------------------------------------
    @classmethod
    def concatenate(cls, batches: List["CausalLMBatch"]) -> "CausalLMBatch":
        if any(b.past_key_values is None for b in batches):
            raise ValueError("can only concatenate prefilled batches")
        head = batches[0]
        B = sum(len(b) for b in batches)
        tokens = max(b.max_sequence_length for b in batches)
        headroom = max(b.padding_right_offset for b in batches)

        input_ids = head.input_ids.new_empty((B, 1))
        mask = head.attention_mask.new_zeros((B, tokens + headroom))
        all_ids = head.all_input_ids_tensor.new_full((B, tokens + headroom), head.pad_token_id)
        position_ids = None if head.position_ids is None else head.position_ids.new_empty((B, 1))
        requests, input_lengths, remaining = [], [], []
        params, current_tokens, samplings, return_logprobs = [], [], [], []

        row = 0
        for b in batches:
            rows = slice(row, row + len(b))
            input_ids[rows] = b.input_ids
            if position_ids is not None:
                position_ids[rows] = b.position_ids
            # every source window is right-aligned at the merged window's edge; its own headroom is dropped and the
            # merged headroom (zeros / pad ids) takes its place
            seen = b.max_sequence_length
            _right_aligned(mask, rows, b.attention_mask[:, b.window_end - seen:b.window_end], tokens)
            _right_aligned(all_ids, rows, b.all_input_ids_tensor[:, b.tokens_end - seen:b.tokens_end], tokens)
            requests.extend(b.requests)
            input_lengths.extend(b.input_lengths)
            remaining.extend(b.max_remaining_tokens)
            params.extend(r.parameters for r in b.requests)
            current_tokens.extend(b.next_token_chooser.current_tokens)
            samplings.extend(b.next_token_chooser.samplings)
            return_logprobs.extend(b.next_token_chooser.return_logprobs)
            row += len(b)

        # KV: per layer and tensor one zero-filled block with tokens - 1 positions on its time axis, each source's past
        # right-aligned in it (whatever the layout: KVLayout names the axis)
        lay = head.kv_layout
        merged_kv: KVLayers = []
        for layer in range(len(head.past_key_values)):
            outs = []
            for slot, proto in enumerate(lay.slots(head.past_key_values[layer])):
                p4 = lay.by_row(proto, len(head))
                ax = lay.time_axis(slot, p4)
                shape = list(p4.shape)
                shape[0], shape[ax] = B, tokens - 1
                out = proto.new_zeros(shape)
                row = 0
                for b in batches:
                    src = lay.by_row(lay.slots(b.past_key_values[layer])[slot], len(b))
                    past = b.max_sequence_length - 1
                    lay.last(slot, out.narrow(0, row, len(b)), past).copy_(lay.last(slot, src, past))
                    row += len(b)
                outs.append(lay.like(out, proto))
            for b in batches:
                b.past_key_values[layer] = None  # release the source as soon as it is copied
            merged_kv.append(lay.pack(outs))

        ntc0 = head.next_token_chooser
        chooser = HeterogeneousNextTokenChooser.from_pb(
            pb=params, model_eos_token_id=ntc0.eos_token_id, model_pad_token_id=ntc0.pad_token_id,
            return_logprobs=return_logprobs, dtype=ntc0.dtype, device=ntc0.device, samplings=samplings,
            current_tokens=current_tokens)

        return cls(batch_id=head.batch_id, requests=requests, input_ids=input_ids, inputs_embeds=None,
                   attention_mask=mask, position_ids=position_ids, past_key_values=merged_kv,
                   all_input_ids_tensor=all_ids, input_lengths=input_lengths, next_token_chooser=chooser,
                   max_sequence_length=tokens, padding_right_offset=headroom, max_remaining_tokens=remaining,
                   pad_token_id=head.pad_token_id, kv_layout=lay)

    @classmethod
    def prune(cls, batch: "CausalLMBatch", completed_ids: List[int]) -> Optional["CausalLMBatch"]:
        if not completed_ids:
            return batch
        keep = Model.get_indices_to_keep(batch.requests, completed_ids)
        if not keep:
            return None
        size_before = len(batch)
        pick = lambda xs: [xs[i] for i in keep]  # noqa: E731
        batch.requests = pick(batch.requests)
        batch.input_lengths = pick(batch.input_lengths)
        batch.max_remaining_tokens = pick(batch.max_remaining_tokens)
        batch.next_token_chooser = batch.next_token_chooser.filter(keep)

        # the survivors may need a narrower window (drop columns on the left) and less headroom (on the right)
        tokens = max(batch.input_lengths)
        headroom = max(batch.max_remaining_tokens)
        m_end, t_end = batch.window_end, batch.tokens_end
        batch.attention_mask = batch.attention_mask[keep, m_end - tokens:m_end + headroom]
        batch.all_input_ids_tensor = batch.all_input_ids_tensor[keep, t_end - tokens:t_end + headroom]
        batch.input_ids = batch.input_ids[keep]
        if batch.position_ids is not None:
            batch.position_ids = batch.position_ids[keep]
        past = tokens - 1
        lay = batch.kv_layout
        batch.past_key_values = [
            lay.pack([lay.like(lay.last(slot, lay.by_row(t, size_before)[keep], past), t)
                      for slot, t in enumerate(lay.slots(layer))])
            for layer in batch.past_key_values]
        batch.max_sequence_length = tokens
        batch.padding_right_offset = headroom
        return batch


class KeysDimTransposedCausalLMBatch(CausalLMBatch):
    """Keys `[.., head_dim, T]` (BLOOM before transformers 4.4x; reference causal_lm.py:742-748)."""

    @classmethod
    def from_pb(cls, *args, **kwargs):
        batch, errors = super().from_pb(*args, **kwargs)
        if batch is not None:
            batch.kv_layout = KVLayout(keys_time_last=True)
        return batch, errors


class CombinedKVCausalLMBatch(CausalLMBatch):
    """One merged `[B, T, C]` tensor per layer (multi-query GPT-BigCode before transformers 4.4x; :750-756)."""

    @classmethod
    def from_pb(cls, *args, **kwargs):
        batch, errors = super().from_pb(*args, **kwargs)
        if batch is not None:
            batch.kv_layout = KVLayout(merged=True)
        return batch, errors


class CausalLM(Model):
    def __init__(self, model_name: str, revision: Optional[str], deployment_framework: str, dtype: torch.dtype,
                 quantize: Optional[str], model_config: Union[Any] = None, max_sequence_length: Optional[int] = None,
                 engine=None):
        if engine is None:
            from transformers import AutoModelForCausalLM

            from tgis_amd.inference_engine import get_inference_engine_class
            from tgis_amd.utils.hub import get_model_path

            engine = get_inference_engine_class(deployment_framework)(
                get_model_path(model_name, revision), AutoModelForCausalLM, dtype, quantize, model_config,
                max_sequence_length)
        super().__init__(engine, dtype, max_sequence_length)
        # only models whose forward takes position_ids get them (models/model.py:46)
        self.use_position_ids = "position_ids" in inspect.signature(self.model.forward).parameters

        tok, cfg = self.tokenizer, self.model.config
        if getattr(cfg, "pad_token_id", None) is not None:
            tok.pad_token_id = cfg.pad_token_id
        elif tok.pad_token_id is None:
            if getattr(cfg, "eos_token_id", None) is not None:
                tok.pad_token_id = cfg.eos_token_id
            elif tok.eos_token_id is not None:
                tok.pad_token_id = tok.eos_token_id
            else:
                tok.add_special_tokens({"pad_token": "[PAD]"})

        # probe the KV layout once with a one-token forward and pick the batch type for it (:580-586)
        self.kv_layout = KVLayout()
        one = torch.tensor([[1]], device=self.device)
        _, probe, _ = self.forward(input_ids=one, attention_mask=one)
        self.kv_layout = KVLayout.probe(probe)
        self._batch_type = (CombinedKVCausalLMBatch if self.kv_layout.merged
                            else KeysDimTransposedCausalLMBatch if self.kv_layout.keys_time_last else CausalLMBatch)

    @property
    def batch_type(self) -> Type[CausalLMBatch]:
        return self._batch_type

    @batch_type.setter
    def batch_type(self, value):
        self._batch_type = value

    # ---- library boundary: per-layer [k, v] lists <-> whatever cache object this transformers version wants -------
    def _to_library_cache(self, past: Optional[KVLayers]):
        if past is None:
            return None
        if self.kv_layout != KVLayout():
            # pre-4.4x layouts are only produced by models that take the legacy tuples back
            return tuple(past) if self.kv_layout.merged else tuple((k, v) for k, v in past)
        pairs = [(k, v) for k, v in past]
        try:
            from transformers.cache_utils import DynamicCache
        except ImportError:  # tuple-of-tuples era (the reference's 4.40 accepts either)
            return tuple(pairs)
        try:
            return DynamicCache(ddp_cache_data=pairs, config=self.model.config)
        except TypeError:
            return DynamicCache.from_legacy_cache(tuple(pairs))

    @staticmethod
    def _from_library_cache(cache) -> KVLayers:
        if hasattr(cache, "layers"):
            return [[layer.keys, layer.values] for layer in cache.layers]
        if hasattr(cache, "to_legacy_cache"):
            cache = cache.to_legacy_cache()
        if torch.is_tensor(cache[0]):  # merged K/V: one tensor per layer
            return list(cache)
        return [[k, v] for k, v in cache]

    def forward(self, input_ids: Optional[torch.Tensor], attention_mask: torch.Tensor,
                position_ids: Optional[torch.Tensor] = None, past_key_values: Optional[KVLayers] = None,
                inputs_embeds: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, KVLayers, int]:
        """(logits [B, T, V], per-layer [k, v], forward ns).  The model is called directly with explicit inputs — the
        reference goes through `prepare_inputs_for_generation` (:612-628), whose signature is version-dependent; for
        decoder-only models it only selects these same tensors."""
        kwargs = dict(attention_mask=attention_mask, past_key_values=self._to_library_cache(past_key_values),
                      use_cache=True, return_dict=True)
        if inputs_embeds is not None:
            kwargs["inputs_embeds"] = inputs_embeds  # the ids under a soft prompt are placeholders
        else:
            kwargs["input_ids"] = input_ids
        if position_ids is not None:
            kwargs["position_ids"] = position_ids
        t0 = time.time_ns()
        out = self.model.forward(**kwargs)
        took = time.time_ns() - t0
        return out.logits, self._from_library_cache(out.past_key_values), took

    def generate_token(self, batch: CausalLMBatch, first: bool = False, for_concat: bool = False,
                       ) -> Tuple[List[TokenInfo], Optional[List[InputTokens]], List[GenerateError], int]:
        w_end, t_end = batch.window_end, batch.tokens_end
        logits, past, forward_time_ns = self.forward(batch.input_ids, batch.attention_mask[:, :w_end], batch.position_ids,
                                                     batch.past_key_values, batch.inputs_embeds)
        seen = batch.all_input_ids_tensor[:, :t_end]
        next_ids, scores, logprobs = batch.next_token_chooser(input_ids=seen, scores=logits[:, -1, :])

        generated: List[TokenInfo] = []
        input_infos: Optional[List[InputTokens]] = [] if first else None
        decode_errors: List[GenerateError] = []
        for i, request in enumerate(batch.requests):
            try:
                row_scores = scores[i].view(1, -1)
                row_logprobs = logprobs[i].view(1, -1) if request.details.logprobs else None
                generated.append(get_token_info(request, row_scores, next_ids[i].view(-1), row_logprobs))
                if first and request.details.input_toks:
                    n = batch.input_lengths[i]
                    # logits of position p predict token p + 1: the last position belongs to the generated token
                    input_infos.append(get_input_tokens_info(request, seen[i, t_end - n:t_end],
                                                             logits[i, logits.shape[1] - n:-1, :]))
            except Exception as e:
                logging.exception(f"token decoding error for request #{request.id}")
                decode_errors.append(GenerateError(request_id=request.id, message=f"Token decoding error: {str(e)}"))
            batch.input_lengths[i] += 1
            batch.max_remaining_tokens[i] -= 1

        # the new token takes the first headroom column
        batch.attention_mask[:, w_end] = 1
        batch.all_input_ids_tensor[:, t_end] = next_ids

        if first and not for_concat:
            align = w_end - batch.max_sequence_length
            if align:  # the pad-to-8 columns of a CUDA prefill are dropped so that the full length stays reachable
                batch.attention_mask = batch.attention_mask[:, align:]
                lay = batch.kv_layout
                past = [lay.pack([lay.last(slot, t, t.shape[lay.time_axis(slot, t)] - align)
                                  for slot, t in enumerate(lay.slots(layer))]) for layer in past]

        if batch.position_ids is not None:
            batch.position_ids = batch.position_ids[:, -1:] + 1
        batch.input_ids = next_ids.view(-1, 1)
        batch.inputs_embeds = None
        batch.past_key_values = past
        batch.max_sequence_length += 1
        batch.padding_right_offset -= 1
        return generated, input_infos, decode_errors, forward_time_ns
