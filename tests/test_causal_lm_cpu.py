"""BASELINE config 1 (row a19): the padded `CausalLMBatch` / `CausalLM` path on the `hf_transformers` engine, on CPU,
against fixtures captured from the reference's own `causal_lm` path (tests/golden/make_gpt2_fixtures.py).

Pinned bit-exact: every batch tensor after `from_pb` (input_ids, attention_mask, position_ids, all_input_ids_tensor,
input_lengths, padding_right_offset, max_sequence_length), after decode steps, after `concatenate` and after `prune`;
token ids, ranks and request order of every step — including seeded sampling, whose per-request torch.Generator
streams are reproducible on CPU.  Logits / logprobs: the same library model on the same machine class, fp32 — 2e-4
absolute (thread-count dependent summation order)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.tiny_models import TinyGPT2Config, tiny_gpt2_tensors
from tests.fixture_utils import GOLDEN, load_fixture

LOGIT_TOL = 2e-4


def _extra(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files if not k.startswith("s") or not k[1].isdigit()}


@pytest.fixture(scope="module")
def lm():
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import GPT2Config, GPT2LMHeadModel, PreTrainedTokenizerFast

    from tgis_amd.inference_engine.hf_transformers import InferenceEngine
    from tgis_amd.models.causal_lm import CausalLM

    cfg = TinyGPT2Config()
    vocab = {"<pad>": 0, "<s>": 1, "</s>": 2, **{f"t{i}": i for i in range(3, cfg.vocab_size)}}
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<pad>"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    tok = PreTrainedTokenizerFast(tokenizer_object=tk, eos_token="</s>", bos_token="<s>", unk_token="<pad>",
                                  pad_token="<pad>", padding_side="left", truncation_side="left")
    hf = GPT2LMHeadModel(GPT2Config(
        vocab_size=cfg.vocab_size, n_embd=cfg.n_embd, n_layer=cfg.n_layer, n_head=cfg.n_head, n_positions=cfg.n_positions,
        layer_norm_epsilon=cfg.layer_norm_epsilon, activation_function=cfg.activation_function, attn_pdrop=0.0,
        resid_pdrop=0.0, embd_pdrop=0.0, pad_token_id=0, bos_token_id=1, eos_token_id=2))
    sd = {k: v.float() for k, v in tiny_gpt2_tensors(cfg, seed=17).items()}
    sd["lm_head.weight"] = sd["transformer.wte.weight"]
    missing, unexpected = hf.load_state_dict(sd, strict=False)
    assert not unexpected
    eng = InferenceEngine(None, None, torch.float32, None, None, 128, preloaded=hf, tokenizer=tok)
    model = CausalLM("gpt2-tiny", None, "hf_transformers", torch.float32, None, engine=eng)
    assert model.use_position_ids and model.device.type == "cpu"
    return model


def _pb(blob):
    from tgis_amd.pb import generate_pb2 as pb2

    b = pb2.Batch()
    b.ParseFromString(bytes(blob))
    return b


def _requests(prompts, max_new, first_id=0, batch_id=0, truncate_to=None):
    from tests.fixture_utils import prompt_text
    from tgis_amd.pb import generate_pb2 as pb2

    reqs = []
    for i, p in enumerate(prompts):
        keep = len(p) if truncate_to is None or truncate_to[i] is None else truncate_to[i]
        r = pb2.Request(id=first_id + i, inputs=prompt_text(p), input_length=keep, truncate=keep != len(p),
                        max_output_length=max_new[i] if isinstance(max_new, list) else max_new)
        r.details.logprobs = True
        r.details.top_n_toks = 2
        r.details.ranks = True
        reqs.append(r)
    return pb2.Batch(id=batch_id, requests=reqs)


def _from_pb(lm, pb):
    with lm.context_manager():
        batch, errs = lm.batch_type.from_pb(pb, lm.tokenizer, lm.dtype, lm.device, lm.word_embeddings, lm.prefix_cache,
                                            lm.use_position_ids)
    assert not errs
    return batch


class _Tap:
    def __init__(self, lm):
        self.lm, self.last = lm, None
        orig = lm.forward

        def fwd(*a, **kw):
            out = orig(*a, **kw)
            self.last = out[0][:, -1, :].detach().float().numpy().copy()
            return out

        lm.forward = fwd

    def close(self):
        del self.lm.forward


def _step(lm, tap, batch, first=False, for_concat=False):
    with lm.context_manager():
        toks, in_toks, errs, ns = lm.generate_token(batch, first=first, for_concat=for_concat)
    assert not errs and ns > 0
    return toks, in_toks, tap.last


def _same_state(batch, extra, tag):
    np.testing.assert_array_equal(batch.input_ids.numpy(), extra[f"{tag}_input_ids"], err_msg=f"{tag}: input_ids")
    np.testing.assert_array_equal(batch.attention_mask.numpy(), extra[f"{tag}_attention_mask"], err_msg=f"{tag}: mask")
    np.testing.assert_array_equal(batch.all_input_ids_tensor.numpy(), extra[f"{tag}_all_input_ids"],
                                  err_msg=f"{tag}: all_input_ids_tensor")
    np.testing.assert_array_equal(batch.position_ids.numpy(), extra[f"{tag}_position_ids"], err_msg=f"{tag}: positions")
    assert batch.input_lengths == extra[f"{tag}_input_lengths"].tolist(), f"{tag}: input_lengths"
    assert batch.max_remaining_tokens == extra[f"{tag}_remaining"].tolist(), f"{tag}: max_remaining_tokens"
    assert [batch.max_sequence_length, batch.padding_right_offset] == extra[f"{tag}_geometry"].tolist(), f"{tag}: geometry"


def _same_tokens(toks, logits, want, what, exact_ids=True):
    assert [t.request_id for t in toks] == want["request_ids"].tolist(), f"{what}: request order"
    assert [t.token_id for t in toks] == want["ids"].tolist(), f"{what}: token ids"
    assert [t.rank for t in toks] == want["ranks"].tolist(), f"{what}: ranks"
    np.testing.assert_allclose(logits, want["logits"], atol=LOGIT_TOL, rtol=0, err_msg=f"{what}: logits")
    np.testing.assert_allclose([t.logprob for t in toks], want["logprobs"], atol=LOGIT_TOL, rtol=0,
                               err_msg=f"{what}: logprobs")
    for t, wt in zip(toks, want["top"]):
        assert [tt.token_id for tt in t.top_tokens] == [x[0] for x in wt], f"{what}: top-n ids"


def test_equal_length_batch_32_new_tokens(lm):
    """B = 4 prompts of 16 tokens, 32 new tokens: the workload of BASELINE config 1."""
    meta, steps = load_fixture("gpt2_equal")
    extra = _extra("gpt2_equal")
    assert meta["batch_type"] == "CausalLMBatch" and meta["use_position_ids"]
    batch = _from_pb(lm, _pb(extra["pb"]))
    _same_state(batch, extra, "frompb")
    tap = _Tap(lm)
    try:
        for i, want in enumerate(steps):
            toks, in_toks, logits = _step(lm, tap, batch, first=(i == 0))
            _same_tokens(toks, logits, want, f"equal step {i}")
            if i == 0:  # input-token details were requested: one entry per prompt token, the first without a logprob
                assert len(in_toks) == 4 and all(len(it.tokens) == 16 for it in in_toks)
            if f"after{i}_input_ids" in extra:
                _same_state(batch, extra, f"after{i}")
    finally:
        tap.close()
    assert batch.padding_right_offset == 0 and batch.max_sequence_length == 48


def test_padded_batch_with_truncation(lm):
    meta, steps = load_fixture("gpt2_padded")
    extra = _extra("gpt2_padded")
    batch = _from_pb(lm, _requests(meta["prompts"], meta["max_new"], truncate_to=meta["truncate_to"]))
    _same_state(batch, extra, "frompb")
    assert batch.input_lengths == [5, 20, 17, 1] and batch.max_sequence_length == 20
    tap = _Tap(lm)
    try:
        for i, want in enumerate(steps):
            toks, _, logits = _step(lm, tap, batch, first=(i == 0))
            _same_tokens(toks, logits, want, f"padded step {i}")
            _same_state(batch, extra, f"after{i}")
    finally:
        tap.close()


def test_continuous_batching_concatenate_prune(lm):
    meta, steps = load_fixture("gpt2_continuous")
    extra = _extra("gpt2_continuous")
    tap = _Tap(lm)
    try:
        a = _from_pb(lm, _requests(meta["prompts_a"], meta["max_new_a"], first_id=0, batch_id=1))
        got = [_step(lm, tap, a, first=True), _step(lm, tap, a), _step(lm, tap, a)]
        b = _from_pb(lm, _requests(meta["prompts_b"], meta["max_new_b"], first_id=2, batch_id=2))
        got.append(_step(lm, tap, b, first=True, for_concat=True))
        _same_state(a, extra, "a_before_concat")
        _same_state(b, extra, "b_before_concat")
        with lm.context_manager():
            merged = lm.batch_type.concatenate([a, b])
        _same_state(merged, extra, "merged")
        assert merged.batch_id == 1 and len(merged) == 3
        kv = merged.past_key_values[0][0]
        assert kv.shape[0] == 3 and kv.shape[2] == merged.max_sequence_length - 1
        got += [_step(lm, tap, merged), _step(lm, tap, merged)]
        with lm.context_manager():
            merged = lm.batch_type.prune(merged, [2])
        _same_state(merged, extra, "pruned")
        got += [_step(lm, tap, merged), _step(lm, tap, merged)]
        _same_state(merged, extra, "final")
    finally:
        tap.close()
    for i, ((toks, _, logits), want) in enumerate(zip(got, steps)):
        _same_tokens(toks, logits, want, f"continuous step {i}")
    with lm.context_manager():
        assert lm.batch_type.prune(merged, [0, 1]) is None
        with pytest.raises(ValueError):
            lm.batch_type.concatenate([_from_pb(lm, _requests(meta["prompts_b"], 3)), merged])


def test_seeded_sampling_and_processors_reproduce_the_reference_on_cpu(lm):
    meta, steps = load_fixture("gpt2_sampled")
    batch = _from_pb(lm, _pb(_extra("gpt2_sampled")["pb"]))
    tap = _Tap(lm)
    try:
        for i, want in enumerate(steps):
            toks, _, logits = _step(lm, tap, batch, first=(i == 0))
            assert [t.token_id for t in toks] == want["ids"].tolist(), f"sampled step {i}: token ids"
            np.testing.assert_allclose(logits, want["logits"], atol=LOGIT_TOL, rtol=0)
            np.testing.assert_allclose([t.logprob for t in toks], want["logprobs"], atol=1e-3, rtol=0)
    finally:
        tap.close()


def test_get_model_dispatch_honours_the_deployment_framework(tmp_path, lm, monkeypatch):
    """A checkpoint directory + `hf_transformers` -> CausalLM on that engine plugin (CPU here); asking for the flash
    path without a GPU fails loudly instead of falling back."""
    from tgis_amd.models import get_model
    from tgis_amd.models.causal_lm import CausalLM

    lm.model.save_pretrained(tmp_path, safe_serialization=True)
    lm.tokenizer.save_pretrained(tmp_path)
    monkeypatch.delenv("FLASH_ATTENTION", raising=False)
    m = get_model(str(tmp_path), None, "hf_transformers", "float32", None, 128)
    assert isinstance(m, CausalLM) and type(m.engine).__module__.endswith("inference_engine.hf_transformers")
    assert m.dtype == torch.float32 and m.batch_type.__name__ == "CausalLMBatch"
    meta, steps = load_fixture("gpt2_equal")
    batch = _from_pb(m, _pb(_extra("gpt2_equal")["pb"]))
    with m.context_manager():
        toks, _, errs, _ = m.generate_token(batch, first=True)
    assert [t.token_id for t in toks] == steps[0]["ids"].tolist()
    with pytest.raises(ValueError, match="Quantization requires CUDA"):
        get_model(str(tmp_path), None, "hf_transformers", "float32", "gptq", 128)
    if not torch.cuda.is_available():
        monkeypatch.setenv("FLASH_ATTENTION", "true")
        with pytest.raises(NotImplementedError):
            get_model(str(tmp_path), None, "tgis_native", "float16", None, 128)
        monkeypatch.delenv("FLASH_ATTENTION")
        with pytest.raises(ModuleNotFoundError):
            get_model(str(tmp_path), None, "no_such_engine", "float32", None, 128)


def test_servicer_runs_the_padded_batch_type(lm):
    """The gRPC servicer drops batches through `release()` (health check, failed step): the padded batch type answers it,
    so the cfg1 / `hf_transformers` shard passes the router's health probe and a failing step surfaces its own error
    (reference server.py:105-180: the health-check batch is generated and discarded)."""
    import asyncio
    import tempfile

    import grpc

    from tgis_amd.cache import Cache
    from tgis_amd.pb import generate_pb2 as pb2
    from tgis_amd.pb import generate_pb2_grpc
    from tgis_amd.server import HEALTHCHECK_BATCH_ID, MemoryScalingModel, TextGenerationService

    async def run():
        with tempfile.TemporaryDirectory() as d:
            url = f"unix://{d}/shard-0"
            server = grpc.aio.server()
            svc = TextGenerationService(lm, Cache(), [url], MemoryScalingModel(1000))
            generate_pb2_grpc.add_TextGenerationServiceServicer_to_server(svc, server)
            server.add_insecure_port(url)
            await server.start()
            async with grpc.aio.insecure_channel(url) as ch:
                stub = generate_pb2_grpc.TextGenerationServiceStub(ch)
                probe = _requests([[5, 6, 7]], 2, batch_id=HEALTHCHECK_BATCH_ID)
                r = await stub.Prefill(pb2.PrefillRequest(batch=probe))
                assert len(r.result.output_tokens) == 1 and len(svc.cache) == 0
                # a normal batch: prefill, one decode step, then finished
                r = await stub.Prefill(pb2.PrefillRequest(batch=_requests([[5, 6, 7], [8, 9]], 3, batch_id=1)))
                assert [t.request_id for t in r.result.output_tokens] == [0, 1] and svc.cache.keys() == [1]
                cb = pb2.CachedBatch(batch_id=1)
                cb.status.completed_ids.extend([])
                r = await stub.NextToken(pb2.NextTokenRequest(batches=[cb]))
                assert len(r.result.output_tokens) == 2
                # a failing step reports ITS error (not an AttributeError from the clean-up) and drops the batch
                real = lm.generate_token

                def boom(*a, **k):
                    raise RuntimeError("HIP out of memory. (injected)")

                lm.generate_token = boom
                try:
                    with pytest.raises(grpc.aio.AioRpcError) as e:
                        await stub.NextToken(pb2.NextTokenRequest(batches=[cb]))
                    assert "injected" in (e.value.details() or "")
                finally:
                    lm.generate_token = real
            await server.stop(0)

    asyncio.run(run())


# ---- the reference's other KV-cache layouts (causal_lm.py:742-756) -------------------------------------------------------
class _LayoutAdapter(torch.nn.Module):
    """The SAME tiny GPT-2, speaking one of the pre-4.4x cache layouts at its boundary: `bloom` = keys
    [B * heads, head_dim, T], values [B * heads, T, head_dim]; `merged` = one [B, T, 2 * heads * head_dim] tensor per
    layer.  What goes in and out of the wrapped model is the standard layout, so the logits are those of the fixtures."""

    def __init__(self, hf, kind, to_lib, from_lib):
        super().__init__()
        self.hf, self.kind, self.config = hf, kind, hf.config
        self._to_lib, self._from_lib = to_lib, from_lib
        self.heads = hf.config.n_head

    def _standard(self, past, B):
        out = []
        for layer in past:
            if self.kind == "merged":
                T = layer.shape[1]
                k, v = layer.view(B, T, 2, self.heads, -1).unbind(2)
                out.append([k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3)])
            else:
                k, v = layer
                D, T = k.shape[-2:]
                out.append([k.view(B, self.heads, D, T).permute(0, 1, 3, 2), v.view(B, self.heads, T, D)])
        return out

    def _native(self, past):
        out = []
        for k, v in past:
            B, H, T, D = k.shape
            if self.kind == "merged":
                out.append(torch.stack([k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3)], dim=2).reshape(B, T, 2 * H * D))
            else:
                out.append((k.permute(0, 1, 3, 2).reshape(B * H, D, T), v.reshape(B * H, T, D)))
        return tuple(out)

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, use_cache=True,
                return_dict=True, inputs_embeds=None):
        B = (input_ids if input_ids is not None else inputs_embeds).shape[0]
        past = None if past_key_values is None else self._to_lib([[k.contiguous(), v.contiguous()]
                                                                  for k, v in self._standard(past_key_values, B)])
        kw = dict(attention_mask=attention_mask, position_ids=position_ids, past_key_values=past, use_cache=True,
                  return_dict=True)
        kw["input_ids" if inputs_embeds is None else "inputs_embeds"] = input_ids if inputs_embeds is None else inputs_embeds
        out = self.hf(**kw)
        out.past_key_values = self._native(self._from_lib(out.past_key_values))
        return out


@pytest.mark.parametrize("kind", ["bloom", "merged"])
def test_continuous_batching_on_the_older_kv_layouts(lm, kind):
    """`KeysDimTransposedCausalLMBatch` / `CombinedKVCausalLMBatch`: the continuous-batching scenario of the reference
    fixture (prefill, decode, prefill-for-concat, concatenate, decode, prune, decode) with the model's cache in the BLOOM
    or the merged multi-query layout.  Same model, same inputs: batch tensors bit for bit and tokens / logits as the
    fixture captured from the reference."""
    from tgis_amd.inference_engine.hf_transformers import InferenceEngine
    from tgis_amd.models import causal_lm as clm

    std = clm.CausalLM._to_library_cache.__get__(lm)  # the standard-layout conversion of the fixture model
    adapter = _LayoutAdapter(lm.model, kind, std, clm.CausalLM._from_library_cache)
    eng = InferenceEngine(None, None, torch.float32, None, None, 128, preloaded=adapter, tokenizer=lm.tokenizer)
    alt = clm.CausalLM("gpt2-tiny-" + kind, None, "hf_transformers", torch.float32, None, engine=eng)
    want_type = clm.CombinedKVCausalLMBatch if kind == "merged" else clm.KeysDimTransposedCausalLMBatch
    assert alt.batch_type is want_type and alt.use_position_ids
    meta, steps = load_fixture("gpt2_continuous")
    extra = _extra("gpt2_continuous")
    tap = _Tap(alt)
    try:
        a = _from_pb(alt, _requests(meta["prompts_a"], meta["max_new_a"], first_id=0, batch_id=1))
        assert isinstance(a, want_type) and a.merged_kv_cache == (kind == "merged") and a.keys_head_dim_last == (kind != "bloom")
        got = [_step(alt, tap, a, first=True), _step(alt, tap, a), _step(alt, tap, a)]
        b = _from_pb(alt, _requests(meta["prompts_b"], meta["max_new_b"], first_id=2, batch_id=2))
        got.append(_step(alt, tap, b, first=True, for_concat=True))
        with alt.context_manager():
            merged = alt.batch_type.concatenate([a, b])
        _same_state(merged, extra, "merged")
        layer0 = merged.past_key_values[0]
        past = merged.max_sequence_length - 1
        if kind == "merged":
            assert torch.is_tensor(layer0) and layer0.shape[:2] == (3, past)
        else:
            H = lm.model.config.n_head
            assert layer0[0].dim() == 3 and layer0[0].shape[0] == 3 * H and layer0[0].shape[2] == past
            assert layer0[1].shape[:2] == (3 * H, past)
        got += [_step(alt, tap, merged), _step(alt, tap, merged)]
        with alt.context_manager():
            merged = alt.batch_type.prune(merged, [2])
        _same_state(merged, extra, "pruned")
        got += [_step(alt, tap, merged), _step(alt, tap, merged)]
        _same_state(merged, extra, "final")
    finally:
        tap.close()
    for i, ((toks, _, logits), want) in enumerate(zip(got, steps)):
        _same_tokens(toks, logits, want, f"{kind} layout, continuous step {i}")


def test_kv_layout_time_axis_operations():
    """KVLayout's primitives on synthetic caches of the three layouts: the last n positions, the per-row view of flattened
    heads and the way back — what concatenate / prune / the pad-to-8 trim of a CUDA prefill are built from."""
    from tgis_amd.models.causal_lm import KVLayout

    B, H, T, D = 3, 2, 7, 4
    k = torch.arange(B * H * T * D, dtype=torch.float32).view(B, H, T, D)
    v = -k
    std, bloom, merged = KVLayout(), KVLayout(keys_time_last=True), KVLayout(merged=True)
    assert KVLayout.probe([[k, v]]) == std
    kb, vb = k.permute(0, 1, 3, 2).reshape(B * H, D, T), v.reshape(B * H, T, D)
    assert KVLayout.probe([[kb, vb]]) == bloom
    m = torch.cat([k.permute(0, 2, 1, 3).reshape(B, T, H * D), v.permute(0, 2, 1, 3).reshape(B, T, H * D)], dim=-1)
    assert KVLayout.probe([m]) == merged
    n = 3
    assert torch.equal(std.last(0, k, n), k[:, :, T - n:, :]) and torch.equal(std.last(1, v, n), v[:, :, T - n:, :])
    assert torch.equal(bloom.last(0, kb, n), kb[:, :, T - n:]) and torch.equal(bloom.last(1, vb, n), vb[:, T - n:, :])
    assert torch.equal(merged.last(0, m, n), m[:, T - n:, :])
    keep = [0, 2]
    rows = bloom.by_row(kb, B)
    assert rows.shape == (B, H, D, T)
    back = bloom.like(bloom.last(0, rows[keep], n), kb)
    assert back.shape == (len(keep) * H, D, n)
    assert torch.equal(back.view(len(keep), H, D, n).permute(0, 1, 3, 2), k[keep][:, :, T - n:, :])
    assert merged.by_row(m, B) is m and merged.slots(m) == [m] and merged.pack([m]) is m


def test_reference_layout_flags_are_assignable():
    """The reference's batch classes SET keys_head_dim_last / merged_kv_cache (causal_lm.py:742-756); here they are views of
    kv_layout, and assigning them rewrites it (they used to be read-only properties)."""
    from tgis_amd.models.causal_lm import CausalLMBatch, KVLayout

    b = CausalLMBatch.__new__(CausalLMBatch)
    b.kv_layout = KVLayout()
    assert b.keys_head_dim_last is True and b.merged_kv_cache is False
    b.keys_head_dim_last = False
    assert b.kv_layout == KVLayout(keys_time_last=True) and b.keys_head_dim_last is False
    b.merged_kv_cache = True
    assert b.kv_layout == KVLayout(merged=True, keys_time_last=True) and b.merged_kv_cache is True
