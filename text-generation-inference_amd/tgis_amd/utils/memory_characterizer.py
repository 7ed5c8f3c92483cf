"""The batch-weight model the shard reports to the router in ModelInfo (mirrors what
utils/memory_characterizer.py:42-143 of the reference produces; the router turns it into `FlashBatch` /
`PaddedBatch` weights, router/src/server.rs:277-311, batch_types.rs:46-208).

Flash path (paged KV).  The unit of weight is one cached token:
  * next-token: a request holds ceil(tokens so far / 32) pages, so the batch costs `tokens + at most 31 per request`:
    nexttoken_linear_coef1 = 1 and the limit leaves one page per request of the largest batch aside;
  * prefill: activations come from the memory left outside the KV pool.  Their peak is measured once at start-up with
    two synthetic prefills (the reference samples the same way, :152-539, with its "test test ..." batches) and fitted
    as `intercept + slope * tokens` — the intercept holds the dequantisation scratch and the logits, the slope the
    [T, E], [T, (H + 2 Hkv) D], [T, 2 I] buffers.  The router checks `prefill tokens * prefill_linear_coef0 <=
    weight_limit`, so the coefficient is `weight_limit / (largest prefill that fits)`.
Padded path (CausalLM on a library model): the reference's manual quadratic / linear models."""
import os
import sys
from typing import Optional, Tuple

from tgis_amd.pb import generate_pb2

ESTIMATE_MEMORY = os.getenv("ESTIMATE_MEMORY", "auto")  # auto | manual | off
PAGE = 32


class MemoryScalingModel:
    def __init__(self, weight_limit: int, prefill_params: Tuple[float, float, float] = (0.0, 0.0, 0.0),
                 next_token_params: Tuple[float, float] = (0.0, 0.0)):
        self.weight_limit = int(weight_limit)
        self.prefill_params = tuple(float(x) for x in prefill_params)      # linear, quadratic0, quadratic1
        self.next_token_params = tuple(float(x) for x in next_token_params)  # encoder / input, decoder / total

    def as_pb(self):
        return generate_pb2.MemoryScalingModel(
            prefill_linear_coef0=self.prefill_params[0], prefill_quadratic_coef0=self.prefill_params[1],
            prefill_quadratic_coef1=self.prefill_params[2], nexttoken_linear_coef0=self.next_token_params[0],
            nexttoken_linear_coef1=self.next_token_params[1], weight_limit=self.weight_limit)

    # ---- what the router evaluates (router/src/batch_types.rs) ------------------------------------------------------
    def prefill_weight(self, batch_size: int, input_len: int) -> float:
        lin, q0, q1 = self.prefill_params
        tokens = batch_size * input_len
        return max(lin * tokens, q0 * tokens + q1 * input_len * tokens)

    def next_token_weight(self, batch_size: int, input_len: int, output_len: int) -> float:
        c0, c1 = self.next_token_params
        if c0 == 0.0:  # flash batches: one coefficient on all tokens present
            return c1 * batch_size * (input_len + output_len)
        return batch_size * (c0 * input_len + c1 * output_len)

    # ---- constructors -------------------------------------------------------------------------------------------
    @classmethod
    def disabled(cls):
        return cls(sys.maxsize >> 1)

    @classmethod
    def manual_quadratic(cls, safety_margin: int, max_seq_len: int, max_batch_size: int):
        """Percent-of-capacity model: 100 = max_seq_len^2 * batch in prefill, max_seq_len * batch per step (:110-128)."""
        p = (100.0 - safety_margin) / 100.0
        lin = 100.0 / (p * max_seq_len * max_batch_size)
        return cls(100, (0.0, 0.0, 100.0 / (p * max_seq_len * max_seq_len * max_batch_size)), (lin, lin))

    @classmethod
    def manual_linear(cls, safety_margin: int, max_seq_len: int, max_batch_size: int):
        p = (100.0 - safety_margin) / 100.0
        lin = 100.0 / (p * max_seq_len * max_batch_size)
        return cls(100, (lin, 0.0, 0.0), (lin, lin))

    @classmethod
    def paged(cls, num_pages: int, max_batch_size: int, safety_margin: int, prefill_tokens_max: Optional[float]):
        """Token-unit model of a paged KV pool.  `prefill_tokens_max`: largest Σ input tokens one Prefill may carry
        (None: bounded by the pool only)."""
        usable_pages = max(1, num_pages - max_batch_size)  # page round-up: at most one partly filled page per request
        limit = usable_pages * PAGE * (100 - safety_margin) // 100
        coef = 1.0
        if prefill_tokens_max is not None and prefill_tokens_max > 0:
            coef = max(1.0, limit / prefill_tokens_max)  # a prefill can never need less than its own KV
        return cls(limit, (coef, 0.0, 0.0), (0.0, 1.0))


def measure_prefill_peak(model, tokens_per_request: int, requests: int) -> int:
    """Peak bytes the allocator handed out during one synthetic prefill (inputs as the reference's estimator builds them:
    'test ' repeated, truncated to input_length, greedy)."""
    import torch

    word = getattr(model.tokenizer, "probe_word", "test")  # test tokenizers with a closed vocabulary name one of theirs
    reqs = [generate_pb2.Request(id=i, inputs=f"{word} " * (tokens_per_request + 8), input_length=tokens_per_request,
                                 truncate=True, max_output_length=1) for i in range(requests)]
    pb = generate_pb2.Batch(id=0, requests=reqs, total_tokens=tokens_per_request * requests)
    torch.cuda.synchronize(model.device)
    torch.cuda.empty_cache()
    base = torch.cuda.memory_allocated(model.device)
    torch.cuda.reset_peak_memory_stats(model.device)
    with model.context_manager():
        batch, errs = model.batch_type.from_pb(pb, model.tokenizer, model.dtype, model.device, model.word_embeddings,
                                               model.prefix_cache, model.use_position_ids)
        if batch is None or errs:
            raise RuntimeError(f"synthetic prefill batch could not be built: {errs}")
        try:
            model.generate_token(batch, first=True)
            torch.cuda.synchronize(model.device)
        finally:
            batch.release()
    peak = torch.cuda.max_memory_allocated(model.device) - base
    del batch
    torch.cuda.empty_cache()
    return int(peak)


def characterize_paged(model, max_sequence_length: int, max_batch_size: int, safety_margin: int) -> MemoryScalingModel:
    """Start-up measurement for a FlashCausalLM: fit the prefill activation peak, read the free memory next to the KV
    pool, and turn both into the router's coefficients."""
    import torch

    kv = model.kv_cache
    if ESTIMATE_MEMORY == "off":
        return MemoryScalingModel.disabled()
    prefill_max = None
    if ESTIMATE_MEMORY != "manual":
        per_req = max(PAGE, min(max_sequence_length - 1, 512))
        small, large = per_req, per_req * 4
        fit_pages = (kv.free_pages - 8) * PAGE
        if large + 4 * PAGE <= fit_pages:
            p1 = measure_prefill_peak(model, per_req, 1)
            p2 = measure_prefill_peak(model, per_req, 4)
            slope = max(1.0, (p2 - p1) / float(large - small))
            intercept = max(0.0, p1 - slope * small)
            free, _total = torch.cuda.mem_get_info(model.device)
            free += torch.cuda.memory_reserved(model.device) - torch.cuda.memory_allocated(model.device)
            budget = free * (100 - safety_margin) / 100.0 - intercept
            prefill_max = max(float(per_req), budget / slope)
            model.prefill_memory_fit = {"bytes_per_token": slope, "fixed_bytes": intercept, "free_bytes": int(free),
                                        "max_prefill_tokens": int(prefill_max)}
    return MemoryScalingModel.paged(kv.num_pages, max_batch_size, safety_margin, prefill_max)
