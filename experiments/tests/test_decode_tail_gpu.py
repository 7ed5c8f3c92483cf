"""-m gpu: the persistent decode tail (csrc/decode_tail.hip: o_proj -> norm -> gate_up -> down -> norm -> next qkv ->
rope + cache write in ONE launch, grid barriers between the phases) against the seven separate launches it replaces.

The phases run the same device code on the same data in the same order, so everything must be BIT-identical: logits of
every step, token ids, and the KV pages the fused launch wrote for the next layer.  A wrong hand-off between
workgroups (a stale line read after a barrier) shows up as a difference in some step, so the comparison is repeated
over many decode steps on buffers that are re-used every layer — the situation in which a missing write-through or an
L1 hit on another CU's data goes wrong — and the launch's own barrier-timeout flag must stay clear."""
import gc

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.experimental, pytest.mark.gpu]
# Model-level driver: it switched FlashLlamaModel to the tail through the TGIS_DECODE_TAIL hooks (_decode_tails /
# _forward_decode_tail) that round 4 removed from the product's modeling code together with the entry point.  Kept as the
# record of what was compared (bit-identical logits, ids and KV pages over 24-40 decode steps); to re-run it, check out
# 169815d (the end of round 3), where the hooks and the test live in the product tree.
pytest.skip("needs the round-3 modeling hooks (git checkout 169815d)", allow_module_level=True)


def _build(cfg_kw, layers, seed, use_tail, monkeypatch, B, L, steps, quantize="gptq", dtype=torch.float16):
    monkeypatch.setenv("TGIS_DECODE_TAIL", "true" if use_tail else "false")
    import importlib

    from tests.fixture_utils import FixtureTokenizer
    from tgis_amd.inference_engine.synthetic import InferenceEngine, llama_tensors
    from tgis_amd.models.custom_modeling import flash_llama_modeling as M
    from tgis_amd.models.flash_causal_lm import FlashCausalLM
    from tgis_amd.utils.kv_cache import PagedKVCache

    M.DECODE_TAIL = use_tail
    # the comparison is between the tail's phases and the SAME units as separate launches: the fused qkv + rope launch of
    # round 3 keeps the whole k range in one block (another summation order), so the reference path here runs without it
    from tgis_amd.utils import layers as _layers

    monkeypatch.setattr(_layers, "FUSED_ROPE_GEMM", False)
    cfg = M.LlamaConfig(num_hidden_layers=layers, max_position_embeddings=4096, **cfg_kw)
    tensors = llama_tensors(cfg, quantize, seed=seed, device="cpu", dtype=dtype)
    tok = FixtureTokenizer(cfg.vocab_size)
    eng = InferenceEngine(tensors, cfg, dtype, quantize, tokenizer=tok)
    pages = B * PagedKVCache.pages_for(L + steps + 2) + 8
    lm = FlashCausalLM("tail", None, "synthetic", dtype, quantize, engine=eng, kv_cache_pages=pages)
    return lm, tok, cfg


def _run(lm, tok, prompts, steps):
    from tests.test_fullwidth_gpu import _run_product

    batch, got = _run_product(lm, tok, prompts, steps)
    kv = lm.kv_cache.pool.detach().clone()
    pages = [list(p) for p in batch.pages]
    batch.release()
    return got, kv, pages


LLAMA_7B = dict(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_attention_heads=32,
                num_key_value_heads=32, rms_norm_eps=1e-5)
TINYLLAMA = dict(vocab_size=32000, hidden_size=2048, intermediate_size=5632, num_attention_heads=32,
                 num_key_value_heads=4, rms_norm_eps=1e-5)


@pytest.mark.parametrize("name,cfg_kw,layers,B,L,steps,quantize,dtype", [
    ("llama-7b width, 3 layers, B=32", LLAMA_7B, 3, 32, 40, 24, "gptq", torch.float16),
    ("llama-7b width, 2 layers, B=5", LLAMA_7B, 2, 5, 70, 12, "gptq", torch.float16),
    ("llama-7b width, 4 layers, B=17", LLAMA_7B, 4, 17, 33, 40, "gptq", torch.float16),
    # dense layers (TinyLlama-1.1B width, GQA 8:1, D = 64): SiLU * up runs in down's operand staging
    ("tinyllama width bf16, 4 layers, B=16", TINYLLAMA, 4, 16, 40, 40, None, torch.bfloat16),
    ("tinyllama width f16, 3 layers, B=32", TINYLLAMA, 3, 32, 33, 24, None, torch.float16),
    ("tinyllama width bf16, 2 layers, B=3", TINYLLAMA, 2, 3, 70, 12, None, torch.bfloat16),
])
def test_tail_is_bit_identical_to_separate_launches(gpu_device, monkeypatch, name, cfg_kw, layers, B, L, steps,
                                                    quantize, dtype):
    from tgis_amd import native

    rng = np.random.default_rng(11)
    prompts = [rng.integers(3, cfg_kw["vocab_size"], size=L).tolist() for _ in range(B)]
    lm_t, tok, _ = _build(cfg_kw, layers, 77, True, monkeypatch, B, L, steps, quantize, dtype)
    got_t, kv_t, pages_t = _run(lm_t, tok, prompts, steps)
    assert lm_t.model.model._tails, "the tail path was not taken"
    assert native.decode_tail_status() == 0, "a grid barrier of the tail timed out"
    del lm_t
    gc.collect()  # the model and its captured graphs form a cycle: free them now, not in the middle of the next run
    torch.cuda.empty_cache()
    lm_s, tok, _ = _build(cfg_kw, layers, 77, False, monkeypatch, B, L, steps, quantize, dtype)
    got_s, kv_s, pages_s = _run(lm_s, tok, prompts, steps)
    assert not lm_s.model.model._tails
    assert pages_t == pages_s
    for i, (a, b) in enumerate(zip(got_t, got_s)):
        assert a[0] == b[0], f"{name}: step {i} token ids differ"
        assert np.array_equal(a[1], b[1]), f"{name}: step {i} logits differ (max {np.abs(a[1] - b[1]).max()})"
    assert torch.equal(kv_t, kv_s), f"{name}: KV pages written by the tail differ from the separate launches'"


def test_tail_eager_equals_graph_and_survives_many_replays(gpu_device, monkeypatch):
    """300 graph replays of a 3-layer model (900 persistent launches, 5400 grid barriers on the same buffers): the
    stream of greedy ids must equal the eager run's, and no barrier may time out."""
    from tgis_amd import native

    B, L, steps = 32, 20, 300
    rng = np.random.default_rng(5)
    prompts = [rng.integers(3, LLAMA_7B["vocab_size"], size=L).tolist() for _ in range(B)]
    lm, tok, _ = _build(LLAMA_7B, 3, 3, True, monkeypatch, B, L, steps)
    got_g, _, _ = _run(lm, tok, prompts, steps)
    assert lm.use_graphs and native.decode_tail_status() == 0
    lm.use_graphs = False
    got_e, _, _ = _run(lm, tok, prompts, 40)
    for i, (a, b) in enumerate(zip(got_g, got_e)):
        assert a[0] == b[0] and np.array_equal(a[1], b[1]), f"step {i}: graph replay differs from the eager launch"
    assert native.decode_tail_status() == 0
