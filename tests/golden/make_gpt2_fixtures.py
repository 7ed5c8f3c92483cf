#!/usr/bin/env python
"""Golden fixtures of the reference's padded `causal_lm` path on a seeded tiny GPT-2 (BASELINE config 1's family):
what `CausalLMBatch.from_pb / concatenate / prune` build and what `CausalLM.generate_token` returns, captured by
running the REFERENCE unmodified in this container (same harness shims as make_fixtures.py).
Needs /root/reference; the outputs (tests/golden/gpt2_*.npz) are committed, this script is how they were made.

    python tests/golden/make_gpt2_fixtures.py

Per scenario: prompts, request parameters, the batch tensors right after from_pb (input_ids, attention_mask,
position_ids, all_input_ids_tensor, input_lengths, padding_right_offset, max_sequence_length), and per
generate_token / concatenate / prune call the returned TokenInfos, the fp32 logits of the last position and the batch
tensors afterwards."""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as mf  # noqa: E402

from oracle.tiny_models import TinyGPT2Config, tiny_gpt2_tensors  # noqa: E402


def write_gpt2_dir(path, cfg, tensors):
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import GPT2Config, GPT2LMHeadModel, PreTrainedTokenizerFast

    vocab = {"<pad>": 0, "<s>": 1, "</s>": 2}
    for i in range(3, cfg.vocab_size):
        vocab[f"t{i}"] = i
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<pad>"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    PreTrainedTokenizerFast(tokenizer_object=tk, eos_token="</s>", bos_token="<s>", unk_token="<pad>",
                            pad_token="<pad>").save_pretrained(path)
    hf_cfg = GPT2Config(vocab_size=cfg.vocab_size, n_embd=cfg.n_embd, n_layer=cfg.n_layer, n_head=cfg.n_head,
                        n_positions=cfg.n_positions, layer_norm_epsilon=cfg.layer_norm_epsilon,
                        activation_function=cfg.activation_function, attn_pdrop=0.0, resid_pdrop=0.0, embd_pdrop=0.0,
                        pad_token_id=0, bos_token_id=1, eos_token_id=2)
    model = GPT2LMHeadModel(hf_cfg)
    sd = {k: v.float() for k, v in tensors.items()}
    sd["lm_head.weight"] = sd["transformer.wte.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all("attn.bias" in m or "masked_bias" in m for m in missing), (missing, unexpected)
    model.float().save_pretrained(path, safe_serialization=True)


def snap(batch, tag):
    """The batch tensors a CausalLMBatch implementation must reproduce (SURVEY.md §8c last row)."""
    out = {f"{tag}_input_ids": batch.input_ids.numpy().copy(),
           f"{tag}_attention_mask": batch.attention_mask.numpy().copy(),
           f"{tag}_all_input_ids": batch.all_input_ids_tensor.numpy().copy(),
           f"{tag}_input_lengths": np.asarray(batch.input_lengths, dtype=np.int64),
           f"{tag}_remaining": np.asarray(batch.max_remaining_tokens, dtype=np.int64),
           f"{tag}_geometry": np.asarray([batch.max_sequence_length, batch.padding_right_offset], dtype=np.int64)}
    if batch.position_ids is not None:
        out[f"{tag}_position_ids"] = batch.position_ids.numpy().copy()
    return out


def requests(pb2, prompts, max_new, first_id=0, batch_id=0, params=None, truncate_to=None, input_toks=False):
    reqs = []
    for i, p in enumerate(prompts):
        keep = len(p) if truncate_to is None or truncate_to[i] is None else truncate_to[i]
        r = pb2.Request(id=first_id + i, inputs=mf.prompt_text(p), input_length=keep, truncate=keep != len(p),
                        max_output_length=max_new[i] if isinstance(max_new, list) else max_new)
        r.details.logprobs = True
        r.details.top_n_toks = 2
        r.details.ranks = True
        r.details.input_toks = input_toks
        for k, v in ((params or {}).get(i) or {}).items():
            setattr(r.parameters, k, v)
        reqs.append(r)
    return pb2.Batch(id=batch_id, requests=reqs, total_tokens=sum(len(p) for p in prompts))


def pb_bytes(pb):
    return np.frombuffer(pb.SerializeToString(), dtype=np.uint8)


def main():
    tmp = tempfile.mkdtemp(prefix="tgis_gpt2_fixture_")
    mf.install_shims(tmp)
    from text_generation_server.models import get_model
    from text_generation_server.pb import generate_pb2 as pb2

    rng = np.random.default_rng(7)
    cfg = TinyGPT2Config()
    tensors = tiny_gpt2_tensors(cfg, seed=17)
    mdir = os.path.join(tmp, "gpt2")
    os.makedirs(mdir)
    write_gpt2_dir(mdir, cfg, tensors)
    model = get_model(mdir, None, "hf_transformers", "float32", None, 128)
    assert type(model.batch_type).__name__ or True
    meta = {"seed": 17, "embed_scale": 6.0, "config": cfg.to_dict(), "use_position_ids": bool(model.use_position_ids),
            "batch_type": model.batch_type.__name__, "transformers": __import__("transformers").__version__,
            "torch": torch.__version__}

    # --- 1: the shape of BASELINE config 1 on the tiny model: B = 4 equal-length prompts of 16, 32 new tokens ------
    prompts = [rng.integers(3, cfg.vocab_size, size=16).tolist() for _ in range(4)]
    pb = requests(pb2, prompts, 32, input_toks=True)
    batch = mf.run_reference(model, pb)
    extra = {"pb": pb_bytes(pb), **snap(batch, "frompb")}
    steps = []
    for i in range(32):
        steps.append(mf.step(model, batch, first=(i == 0)))
        if i in (0, 1, 30):
            extra.update(snap(batch, f"after{i}"))
    mf.save("gpt2_equal", {**meta, "prompts": prompts, "max_new": 32}, steps, extra)

    # --- 2: ragged prompts, different max_new per row, one left-truncated request (version-sensitive: the padded rows
    #        only agree with their unpadded selves because position ids are passed explicitly) ---------
    prompts = [rng.integers(3, cfg.vocab_size, size=n).tolist() for n in (5, 33, 17, 1)]
    max_new = [6, 4, 9, 7]
    trunc = [None, 20, None, None]
    pb = requests(pb2, prompts, max_new, truncate_to=trunc)
    batch = mf.run_reference(model, pb)
    extra = {"pb": pb_bytes(pb), **snap(batch, "frompb")}
    steps = []
    for i in range(4):
        steps.append(mf.step(model, batch, first=(i == 0)))
        extra.update(snap(batch, f"after{i}"))
    mf.save("gpt2_padded", {**meta, "prompts": prompts, "max_new": max_new, "truncate_to": trunc,
                            "note": "version-sensitive (left padding + explicit position ids)"}, steps, extra)

    # --- 3: continuous batching on the padded batch type: prefill A, decode x2, prefill B (for_concat), concatenate,
    #        decode x2, prune, decode x2 — tensors after every membership change -----------------------------------------
    pa = [rng.integers(3, cfg.vocab_size, size=n).tolist() for n in (9, 14)]
    pb_ = [rng.integers(3, cfg.vocab_size, size=n).tolist() for n in (21,)]
    a = mf.run_reference(model, requests(pb2, pa, [10, 7], first_id=0, batch_id=1))
    steps = [mf.step(model, a, first=True), mf.step(model, a), mf.step(model, a)]
    b = mf.run_reference(model, requests(pb2, pb_, 12, first_id=2, batch_id=2))
    steps.append(mf.step(model, b, first=True, for_concat=True))
    extra = {**snap(a, "a_before_concat"), **snap(b, "b_before_concat")}
    with model.context_manager():
        merged = model.batch_type.concatenate([a, b])
    extra.update(snap(merged, "merged"))
    steps += [mf.step(model, merged), mf.step(model, merged)]
    with model.context_manager():
        merged = model.batch_type.prune(merged, [2])
    extra.update(snap(merged, "pruned"))
    steps += [mf.step(model, merged), mf.step(model, merged)]
    extra.update(snap(merged, "final"))
    mf.save("gpt2_continuous", {**meta, "prompts_a": pa, "prompts_b": pb_, "max_new_a": [10, 7], "max_new_b": 12,
                                "script": ["prefill A(ids 0,1)", "decode A", "decode A", "prefill B(id 2, for_concat)",
                                           "concatenate[A,B] + decode", "decode", "prune id 2 + decode", "decode"]},
            steps, extra)

    # --- 4: the chooser on CPU: seeded sampling (per-request torch.Generator streams are reproducible on CPU),
    #        temperature / top-k / top-p / typical-p / repetition penalty, min_new_tokens and a length penalty ------------
    prompts = [rng.integers(3, cfg.vocab_size, size=12).tolist() for _ in range(5)]
    params = {0: dict(temperature=0.9, top_k=20, seed=11), 1: dict(temperature=1.2, top_p=0.8, seed=12),
              2: dict(temperature=1.0, typical_p=0.7, seed=13, repetition_penalty=1.3),
              3: dict(repetition_penalty=1.5, min_new_tokens=4),
              4: dict()}
    pb = requests(pb2, prompts, 8, params=params)
    pb.requests[4].parameters.length_penalty.start_index = 2
    pb.requests[4].parameters.length_penalty.decay_factor = 1.5
    batch = mf.run_reference(model, pb)
    steps = [mf.step(model, batch, first=(i == 0)) for i in range(8)]
    mf.save("gpt2_sampled", {**meta, "prompts": prompts, "max_new": 8}, steps, {"pb": pb_bytes(pb)})


if __name__ == "__main__":
    main()
