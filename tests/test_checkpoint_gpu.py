"""-m gpu: a real checkpoint directory (config.json, model.safetensors, quantize_config.json, tokenizer files) through the
production entry point `get_model` / the tgis_native engine — safetensors slicing, GPTQ parameter discovery, HF
tokenizer with left truncation — against the oracle run on the same tensors."""
import json

import numpy as np
import pytest
import torch

from oracle.llama_ref import LlamaRef
from oracle.tiny_models import TinyLlamaConfig, tiny_llama_tensors

pytestmark = pytest.mark.gpu


def _write_checkpoint(path, cfg, tensors, quantize, groupsize):
    from safetensors.torch import save_file
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast

    conf = dict(cfg.to_dict(), model_type="llama", architectures=["LlamaForCausalLM"], torch_dtype="float16")
    (path / "config.json").write_text(json.dumps(conf))
    if quantize == "gptq":
        (path / "quantize_config.json").write_text(json.dumps({"bits": 4, "group_size": groupsize, "desc_act": False}))
    save_file({k: v.contiguous() for k, v in tensors.items()}, str(path / "model.safetensors"))
    vocab = {"<pad>": 0, "<s>": 1, "</s>": 2}
    vocab.update({f"t{i}": i for i in range(3, cfg.vocab_size)})
    tk = Tokenizer(models.WordLevel(vocab=vocab, unk_token="<pad>"))
    tk.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    PreTrainedTokenizerFast(tokenizer_object=tk, pad_token="<pad>", bos_token="<s>", eos_token="</s>").save_pretrained(str(path))


@pytest.mark.parametrize("quantize", [None, "gptq"])
def test_checkpoint_directory_end_to_end(gpu_device, tmp_path, quantize, monkeypatch):
    from tgis_amd.models import get_model

    monkeypatch.setenv("TGIS_KV_CACHE_FRACTION", "0.01")  # the default pool (85 % of the free memory) is not the subject
    from tgis_amd.pb import generate_pb2 as pb2

    cfg = TinyLlamaConfig()
    tensors = tiny_llama_tensors(cfg, seed=5, quantize=quantize, groupsize=64)
    _write_checkpoint(tmp_path, cfg, tensors, quantize, 64)
    lm = get_model(str(tmp_path), None, "tgis_native", "float16", quantize, max_sequence_length=256)
    assert lm.tokenizer.truncation_side == "left" and lm.tokenizer.padding_side == "left"
    rng = np.random.default_rng(3)
    prompts = [rng.integers(3, cfg.vocab_size, size=n).tolist() for n in (9, 33, 2)]
    # the second request is left-truncated to its last 20 tokens by the batch builder (truncate=True)
    reqs = [pb2.Request(id=i, inputs=" ".join(f"t{t}" for t in p), input_length=(20 if i == 1 else len(p)), truncate=True,
                        max_output_length=6) for i, p in enumerate(prompts)]
    for r in reqs:
        r.details.logprobs = True
    with lm.context_manager():
        batch, errs = lm.batch_type.from_pb(pb2.Batch(id=0, requests=reqs), lm.tokenizer, lm.dtype, lm.device,
                                            lm.word_embeddings, None, True)
        assert not errs
        steps = []
        for i in range(4):
            toks, _, errs, _ = lm.generate_token(batch, first=(i == 0))
            assert not errs
            steps.append(toks)
    batch.release()
    seen = [prompts[0], prompts[1][-20:], prompts[2]]
    ref = LlamaRef(cfg, tensors, quantize=quantize, groupsize=64)
    want = ref.generate_greedy(seen, 4, forced=[[t.token_id for t in s] for s in steps])
    for i, (got, w) in enumerate(zip(steps, want)):
        top2 = torch.topk(w["logits"], 2, dim=-1).values
        for j, t in enumerate(got):
            if t.token_id != int(w["token_ids"][j]):  # only an fp32 near-tie may flip in fp16
                assert float(top2[j, 0] - top2[j, 1]) < 0.75, f"step {i} request {j}: {t.token_id} vs {int(w['token_ids'][j])}"
            else:
                assert abs(t.logprob - float(w["logprobs"][j])) < 0.35
    assert lm.kv_cache.free_pages == lm.kv_cache.num_pages


def test_bigcode_checkpoint_directory_end_to_end(gpu_device, tmp_path, monkeypatch):
    """GPT-BigCode (multi-query) checkpoint through AutoConfig: field names come from transformers' GPTBigCodeConfig
    (n_embd / n_head / n_layer ...), not from the test's own config class."""
    monkeypatch.setenv("TGIS_KV_CACHE_FRACTION", "0.01")
    from oracle.santacoder_ref import SantacoderRef
    from oracle.tiny_models import TinyBigCodeConfig, tiny_bigcode_tensors
    from tgis_amd.models import get_model
    from tgis_amd.pb import generate_pb2 as pb2

    cfg = TinyBigCodeConfig()
    tensors = tiny_bigcode_tensors(cfg, seed=9)
    conf = dict(model_type="gpt_bigcode", architectures=["GPTBigCodeForCausalLM"], vocab_size=cfg.vocab_size,
                n_embd=cfg.hidden_size, n_inner=cfg.n_inner, n_layer=cfg.num_hidden_layers, n_head=cfg.num_attention_heads,
                n_positions=cfg.n_positions, layer_norm_epsilon=cfg.layer_norm_epsilon,
                activation_function=cfg.activation_function, multi_query=True, tie_word_embeddings=True,
                pad_token_id=0, bos_token_id=1, eos_token_id=2, torch_dtype="float16")
    _write_checkpoint(tmp_path, cfg, tensors, None, 0)
    (tmp_path / "config.json").write_text(json.dumps(conf))
    lm = get_model(str(tmp_path), None, "tgis_native", "float16", None, max_sequence_length=256)
    rng = np.random.default_rng(4)
    prompts = [rng.integers(3, cfg.vocab_size, size=n).tolist() for n in (12, 40)]
    reqs = [pb2.Request(id=i, inputs=" ".join(f"t{t}" for t in p), input_length=len(p), truncate=False, max_output_length=5)
            for i, p in enumerate(prompts)]
    with lm.context_manager():
        batch, errs = lm.batch_type.from_pb(pb2.Batch(id=0, requests=reqs), lm.tokenizer, lm.dtype, lm.device,
                                            lm.word_embeddings, None, lm.use_position_ids)
        assert not errs
        steps = []
        for i in range(3):
            toks, _, errs, _ = lm.generate_token(batch, first=(i == 0))
            assert not errs
            steps.append(toks)
    batch.release()
    want = SantacoderRef(cfg, tensors).generate_greedy(prompts, 3, forced=[[t.token_id for t in s] for s in steps])
    for i, (got, w) in enumerate(zip(steps, want)):
        top2 = torch.topk(w["logits"], 2, dim=-1).values
        for j, t in enumerate(got):
            assert t.token_id == int(w["token_ids"][j]) or float(top2[j, 0] - top2[j, 1]) < 0.3, f"step {i} request {j}"
