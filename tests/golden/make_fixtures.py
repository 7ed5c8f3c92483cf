#!/usr/bin/env python
"""Generate the golden fixtures of tests/golden/ by running the REFERENCE's CPU `causal_lm` path, unmodified, in
this container.  Needs /root/reference; the outputs (small .npz files) are committed,
this script is how they were made.  Nothing here is imported by the product or at test time.

    python tests/golden/make_fixtures.py

What is captured, per scenario: the request batch (serialized generate.v1.Batch), the prompt token ids as the
reference tokenized them, and per generate_token call: the tokens / logprobs it returned and the fp32 logits of
the last position for every row (hooked from the HF model the reference drives).
Harness-side shims only (no edits to reference files): a `loguru`/`texttable` stub, `transformers.LogitsWarper`
alias (class removed upstream), `text_generation_server.pb.generate_pb2` built without protoc (our own
tgis_amd.pb.generate_pb2 re-exported), and a legacy-KV adapter around the HF model because the reference indexes
past_key_values as tuples (models/causal_lm.py:580) while transformers 5.x returns Cache objects.
"""
import inspect
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/server"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))

from oracle.tiny_models import (  # noqa: E402
    TinyBigCodeConfig,
    TinyLlamaConfig,
    dense_state_dict,
    tiny_bigcode_tensors,
    tiny_llama_tensors,
)


def install_shims(tmp):
    shim = os.path.join(tmp, "shims")
    os.makedirs(shim)
    with open(os.path.join(shim, "loguru.py"), "w") as f:
        f.write("import logging\nlogger = logging.getLogger('loguru-stub')\n")
    with open(os.path.join(shim, "texttable.py"), "w") as f:
        f.write("class Texttable:\n    pass\n")
    sys.path.insert(0, shim)
    sys.path.insert(0, REF)
    os.environ["HF_HUB_OFFLINE"] = "1"
    import transformers

    transformers.LogitsWarper = transformers.LogitsProcessor
    import text_generation_server

    overlay = os.path.join(tmp, "overlay", "text_generation_server")
    os.makedirs(os.path.join(overlay, "pb"))
    open(os.path.join(overlay, "pb", "__init__.py"), "w").close()
    with open(os.path.join(overlay, "pb", "generate_pb2.py"), "w") as f:
        f.write("from tgis_amd.pb.generate_pb2 import *  # noqa\nfrom tgis_amd.pb import generate_pb2 as _m\n"
                "globals().update({k: getattr(_m, k) for k in _m.__all__})\n")
    text_generation_server.__path__.append(overlay)
    transformers.LogitsWarper = transformers.LogitsProcessor

    # legacy-KV adapter (Appendix C step 4)
    from transformers.cache_utils import DynamicCache

    from text_generation_server.inference_engine import hf_transformers

    orig_init = hf_transformers.InferenceEngine.__init__

    def patched_init(self, *a, **kw):
        orig_init(self, *a, **kw)
        model = self.model
        inner_forward = model.forward
        sig = inspect.signature(type(model).forward)

        def forward(*args, **kwargs):
            pkv = kwargs.get("past_key_values")
            if pkv is not None and not hasattr(pkv, "layers"):
                kwargs["past_key_values"] = DynamicCache(ddp_cache_data=[(k, v) for k, v in pkv], config=model.config)
            out = inner_forward(*args, **kwargs)
            cache = out.past_key_values
            if cache is not None and hasattr(cache, "layers"):
                out.past_key_values = tuple((l.keys, l.values) for l in cache.layers)
            CAPTURE.append(out.logits[:, -1, :].detach().float().clone())
            return out

        forward.__signature__ = sig
        model.forward = forward
        inner_prep = model.prepare_inputs_for_generation

        def prep(input_ids, past_key_values=None, **kwargs):
            return inner_prep(input_ids, past_key_values=past_key_values, **kwargs)

        model.prepare_inputs_for_generation = prep

    hf_transformers.InferenceEngine.__init__ = patched_init


CAPTURE = []
# every token of a fixture is decided by at least this many logits: > 2x the absolute logit tolerance of the fp16 GPU
# tests (0.35 for the Llama fixtures with |logit| up to ~60, 0.08 for the GPT-BigCode ones with |logit| up to ~12)
LLAMA_MARGIN = 0.8
BIGCODE_MARGIN = 0.2


def write_model_dir(path, cfg, tensors, groupsize):
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import LlamaConfig, LlamaForCausalLM, PreTrainedTokenizerFast

    vocab = {"<pad>": 0, "<s>": 1, "</s>": 2}
    for i in range(3, cfg.vocab_size):
        vocab[f"t{i}"] = i
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<pad>"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    PreTrainedTokenizerFast(tokenizer_object=tk, eos_token="</s>", bos_token="<s>", unk_token="<pad>",
                            pad_token="<pad>").save_pretrained(path)
    hf_cfg = LlamaConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                         num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                         num_key_value_heads=cfg.num_key_value_heads, rms_norm_eps=cfg.rms_norm_eps,
                         rope_theta=cfg.rope_theta, max_position_embeddings=cfg.max_position_embeddings,
                         pad_token_id=0, bos_token_id=1, eos_token_id=2, tie_word_embeddings=False,
                         attention_bias=False)
    model = LlamaForCausalLM(hf_cfg)
    sd = dense_state_dict(cfg, tensors, groupsize)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all("rotary" in m or "inv_freq" in m for m in missing), (missing, unexpected)
    model.float().save_pretrained(path, safe_serialization=True)


def write_bigcode_dir(path, cfg, tensors):
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import GPTBigCodeConfig, GPTBigCodeForCausalLM, PreTrainedTokenizerFast

    vocab = {"<pad>": 0, "<s>": 1, "</s>": 2}
    for i in range(3, cfg.vocab_size):
        vocab[f"t{i}"] = i
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<pad>"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    PreTrainedTokenizerFast(tokenizer_object=tk, eos_token="</s>", bos_token="<s>", unk_token="<pad>",
                            pad_token="<pad>").save_pretrained(path)
    hf_cfg = GPTBigCodeConfig(vocab_size=cfg.vocab_size, n_embd=cfg.hidden_size, n_inner=cfg.n_inner,
                              n_layer=cfg.num_hidden_layers, n_head=cfg.num_attention_heads,
                              n_positions=cfg.n_positions, layer_norm_epsilon=cfg.layer_norm_epsilon,
                              activation_function=cfg.activation_function, multi_query=True, attn_pdrop=0.0,
                              resid_pdrop=0.0, embd_pdrop=0.0, pad_token_id=0, bos_token_id=1, eos_token_id=2,
                              scale_attn_weights=True, attention_softmax_in_fp32=True, scale_attention_softmax_in_fp32=True)
    model = GPTBigCodeForCausalLM(hf_cfg)
    sd = {k: v.float() for k, v in tensors.items()}
    sd["lm_head.weight"] = sd["transformer.wte.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all("bias" in m and "attn" in m or "masked_bias" in m for m in missing), (missing, unexpected)
    model.float().save_pretrained(path, safe_serialization=True)


def prompt_text(ids):
    return " ".join(f"t{i}" for i in ids)


def make_requests(pb2, prompts, max_new, first_id=0, batch_id=0, logprobs=True, top_n=0, ranks=False):
    reqs = []
    for i, p in enumerate(prompts):
        r = pb2.Request(id=first_id + i, inputs=prompt_text(p), input_length=len(p), truncate=False,
                        max_output_length=max_new)
        r.details.logprobs = logprobs
        r.details.top_n_toks = top_n
        r.details.ranks = ranks
        reqs.append(r)
    return pb2.Batch(id=batch_id, requests=reqs, total_tokens=sum(len(p) for p in prompts))


def run_reference(model, pb_batch):
    with model.context_manager():
        batch, errs = model.batch_type.from_pb(
            pb_batch, tokenizer=model.tokenizer, dtype=model.dtype, device=model.device,
            embeddings_lookup=model.word_embeddings, prefix_cache=model.prefix_cache,
            use_position_ids=model.use_position_ids)
    assert not errs
    return batch


def step(model, batch, first=False, for_concat=False):
    CAPTURE.clear()
    with model.context_manager():
        toks, in_toks, errs, _ = model.generate_token(batch, first=first, for_concat=for_concat)
    assert not errs
    return {"ids": [t.token_id for t in toks], "logprobs": [t.logprob for t in toks],
            "request_ids": [t.request_id for t in toks], "ranks": [t.rank for t in toks],
            "top": [[(tt.token_id, tt.logprob) for tt in (t.top_tokens or [])] for t in toks],
            "logits": CAPTURE[-1].numpy().copy()}


def min_margin(steps):
    """Smallest top-1 minus top-2 logit over every row of every step."""
    m = float("inf")
    for st in steps:
        top2 = np.sort(st["logits"], axis=1)[:, -2:]
        m = min(m, float((top2[:, 1] - top2[:, 0]).min()))
    return m


def decisive(make, what, need, tries=400):
    """Draw prompts until the reference decides every token of the scenario by at least `need` logits (SURVEY.md §8c
    "fixture design note": margins far above the fp16 tolerance of the GPU tests, so that token ids can be compared
    exactly, without a near-tie rule).  `make()` draws fresh prompts from the shared generator and returns
    (meta_extra, steps)."""
    for attempt in range(tries):
        extra, steps = make()
        m = min_margin(steps)
        if m >= need:
            print(f"{what}: min top-2 margin {m:.3f} after {attempt + 1} draw(s)")
            extra["min_margin"] = m
            return extra, steps
    raise RuntimeError(f"{what}: no draw reached a margin of {need}")


def save(name, meta, steps, extra=None):
    arrays = {"meta": np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)}
    for i, s in enumerate(steps):
        arrays[f"s{i}_ids"] = np.asarray(s["ids"], dtype=np.int64)
        arrays[f"s{i}_request_ids"] = np.asarray(s["request_ids"], dtype=np.int64)
        arrays[f"s{i}_logprobs"] = np.asarray(s["logprobs"], dtype=np.float32)
        arrays[f"s{i}_ranks"] = np.asarray(s["ranks"], dtype=np.int64)
        arrays[f"s{i}_logits"] = s["logits"].astype(np.float32)
        arrays[f"s{i}_top"] = np.frombuffer(json.dumps(s["top"]).encode(), dtype=np.uint8)
    for k, v in (extra or {}).items():
        arrays[k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    print("wrote", name, "steps", len(steps))


def main():
    tmp = tempfile.mkdtemp(prefix="tgis_fixture_")
    install_shims(tmp)
    from text_generation_server.models import get_model
    from text_generation_server.pb import generate_pb2 as pb2

    rng = np.random.default_rng(2024)
    cfg = TinyLlamaConfig()
    GS = 64
    for variant, quantize in (("dense", None), ("gptq", "gptq")):
        tensors = tiny_llama_tensors(cfg, seed=7, quantize=quantize, groupsize=GS)
        mdir = os.path.join(tmp, f"llama_{variant}")
        os.makedirs(mdir)
        write_model_dir(mdir, cfg, tensors, GS)
        model = get_model(mdir, None, "hf_transformers", "float32", None, 256)
        meta_base = {"variant": variant, "quantize": quantize, "groupsize": GS, "seed": 7,
                     "config": cfg.to_dict(), "transformers": __import__("transformers").__version__,
                     "torch": torch.__version__}

        # --- scenario 1: equal-length prompts, greedy, logprobs + top-3 + ranks ------------------------------
        def equal():
            prompts = [rng.integers(3, cfg.vocab_size, size=12).tolist() for _ in range(3)]
            batch = run_reference(model, make_requests(pb2, prompts, max_new=8, top_n=3, ranks=True))
            return {"prompts": prompts}, [step(model, batch, first=True)] + [step(model, batch) for _ in range(7)]

        extra, steps = decisive(equal, f"llama_{variant}_equal", LLAMA_MARGIN)
        save(f"llama_{variant}_equal", {**meta_base, **extra, "max_new": 8, "top_n": 3, "ranks": True}, steps)

        # --- scenario 2: ragged prompts (the reference left-pads; version-sensitive) -------
        def ragged():
            prompts = [rng.integers(3, cfg.vocab_size, size=n).tolist() for n in (5, 33, 17, 1)]
            batch = run_reference(model, make_requests(pb2, prompts, max_new=6))
            return {"prompts": prompts}, [step(model, batch, first=True)] + [step(model, batch) for _ in range(5)]

        extra, steps = decisive(ragged, f"llama_{variant}_ragged", LLAMA_MARGIN)
        save(f"llama_{variant}_ragged", {**meta_base, **extra, "max_new": 6}, steps)

        # --- scenario 3: continuous batching — prefill A, decode, prefill B (for_concat), concatenate, decode,
        #     prune one request of A, decode (server.py:105-231 drives exactly this sequence) ----------------------
        def continuous():
            pa = [rng.integers(3, cfg.vocab_size, size=n).tolist() for n in (9, 14)]
            pb_ = [rng.integers(3, cfg.vocab_size, size=n).tolist() for n in (6,)]
            a = run_reference(model, make_requests(pb2, pa, max_new=10, first_id=0, batch_id=1))
            steps = [step(model, a, first=True), step(model, a), step(model, a)]
            b = run_reference(model, make_requests(pb2, pb_, max_new=10, first_id=2, batch_id=2))
            steps.append(step(model, b, first=True, for_concat=True))
            with model.context_manager():
                merged = model.batch_type.concatenate([a, b])
            steps += [step(model, merged), step(model, merged)]
            with model.context_manager():
                merged = model.batch_type.prune(merged, [0])
            steps += [step(model, merged), step(model, merged)]
            return {"prompts_a": pa, "prompts_b": pb_}, steps

        extra, steps = decisive(continuous, f"llama_{variant}_continuous", LLAMA_MARGIN)
        save(f"llama_{variant}_continuous",
             {**meta_base, **extra, "max_new": 10,
              "script": ["prefill A(ids 0,1)", "decode A", "decode A", "prefill B(id 2, for_concat)",
                         "concatenate[A,B] + decode", "decode", "prune id 0 + decode", "decode"]}, steps)

    # --- GPT-BigCode (multi-query attention, learned positions, LayerNorm, tanh-GELU, tied head) ------------------
    bcfg = TinyBigCodeConfig()
    btensors = tiny_bigcode_tensors(bcfg, seed=13, embed_scale=3.0)
    mdir = os.path.join(tmp, "bigcode")
    os.makedirs(mdir)
    write_bigcode_dir(mdir, bcfg, btensors)
    model = get_model(mdir, None, "hf_transformers", "float32", None, 256)
    bmeta = {"variant": "bigcode", "seed": 13, "embed_scale": 3.0, "config": bcfg.to_dict(),
             "transformers": __import__("transformers").__version__, "torch": torch.__version__}
    def b_equal():
        prompts = [rng.integers(3, bcfg.vocab_size, size=12).tolist() for _ in range(3)]
        batch = run_reference(model, make_requests(pb2, prompts, max_new=6))
        return {"prompts": prompts}, [step(model, batch, first=True)] + [step(model, batch) for _ in range(5)]

    extra, steps = decisive(b_equal, "bigcode_equal", BIGCODE_MARGIN)
    save("bigcode_equal", {**bmeta, **extra, "max_new": 6}, steps)

    def b_ragged():
        prompts = [rng.integers(3, bcfg.vocab_size, size=n).tolist() for n in (4, 35, 18)]
        batch = run_reference(model, make_requests(pb2, prompts, max_new=5))
        return {"prompts": prompts}, [step(model, batch, first=True)] + [step(model, batch) for _ in range(4)]

    extra, steps = decisive(b_ragged, "bigcode_ragged", BIGCODE_MARGIN)
    save("bigcode_ragged", {**bmeta, **extra, "max_new": 5}, steps)

    # --- GPTQ pack pin: the reference's own packer vs oracle.ops_ref.gptq_pack on the same integers -----------
    from text_generation_server.utils.gptq.quant_linear import QuantLinear

    K, N, G = 128, 64, 2
    intw = rng.integers(0, 16, size=(K, N)).astype(np.uint8)
    zeros = rng.integers(1, 17, size=(G, N)).astype(np.uint8)
    scales = (rng.uniform(0.5, 1.5, size=(G, N)) * 0.01).astype(np.float16)
    ql = QuantLinear.new(4, K // G, K, N, bias=False)
    g_idx = torch.tensor([i // (K // G) for i in range(K)], dtype=torch.int32)
    # weights exactly on the quantisation grid so that pack()'s round() recovers intw
    w = (torch.from_numpy(intw.astype(np.float32)) - torch.from_numpy(zeros.astype(np.float32))[g_idx.long()]) \
        * torch.from_numpy(scales.astype(np.float32))[g_idx.long()]
    lin = torch.nn.Linear(K, N, bias=False)
    lin.weight.data = w.t().contiguous()
    ql.pack(lin, torch.from_numpy(scales.astype(np.float32)).t().contiguous(),
            torch.from_numpy(zeros.astype(np.float32)).t().contiguous(), g_idx)
    np.savez_compressed(os.path.join(HERE, "gptq_pack_reference.npz"), intw=intw, zeros=zeros, scales=scales,
                        qweight=ql.qweight.numpy(), qzeros=ql.qzeros.numpy(), ref_scales=ql.scales.numpy(),
                        dequant=w.numpy())
    print("wrote gptq_pack_reference")


if __name__ == "__main__":
    main()
