#!/usr/bin/env python
"""Decode-throughput benchmark of the hot path: FlashCausalLM.generate_token on synthetic fixed-length batches.

Metric (BASELINE.json): decode tokens/s (+ p50 step latency), Llama-2-7B int4 GPTQ g128, batch 32, mean context
1024, fp16 activations.  A "step" is one NextToken-equivalent call: generate_token(batch)
including greedy sampling and the one device->host copy of the token ids.  Weights are seeded synthetic tensors
at the real shapes (no checkpoints offline); KV is produced by a real prefill of the same model before timing.

  python bench.py --gpus 1 --steps 32 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W          # tensor parallel over RCCL, same global batch

Rank 0 prints ONE JSON line with the contract fields plus "roofline" (dominant kernel = paged decode attention,
algorithmic KV bytes / HIP-event launch duration) and "cpu_baseline" (fp32 CPU oracle on a bounded sample).
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "text-generation-inference_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

CONFIGS = {
    # name: (llama config kwargs, quantize, dtype, batch, mean ctx)
    "llama2-7b-gptq": (dict(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                            num_attention_heads=32, num_key_value_heads=32, rms_norm_eps=1e-5), "gptq", "float16", 32,
                       1024),
    "tinyllama-1.1b": (dict(vocab_size=32000, hidden_size=2048, intermediate_size=5632, num_hidden_layers=22,
                            num_attention_heads=32, num_key_value_heads=4, rms_norm_eps=1e-5), None, "bfloat16", 16, 512),
    # cfg4 of BASELINE.json (quoted at TP=8; the int4 model + its KV also fit one 288 GB MI355X)
    "llama2-70b-gptq": (dict(vocab_size=32000, hidden_size=8192, intermediate_size=28672, num_hidden_layers=80,
                             num_attention_heads=64, num_key_value_heads=8, rms_norm_eps=1e-5), "gptq", "float16", 64,
                        2048),
    # cfg5 of BASELINE.json (GPT-BigCode, multi-query attention; quoted at TP=4, fits one GPU)
    "starcoder-15b": (dict(vocab_size=49152, hidden_size=6144, n_inner=24576, num_hidden_layers=40,
                           num_attention_heads=48, n_positions=8192), None, "bfloat16", 32, 4096),
    "llama-tiny-gptq": (dict(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                             num_attention_heads=4, num_key_value_heads=2, rms_norm_eps=1e-5), "gptq", "float16", 4, 64),
}
# cfg1 of BASELINE.json: the padded causal_lm path on CPU (plumbing config; no GPU, no HIP kernels involved)
CPU_CONFIGS = {"gpt2-cpu": dict(vocab_size=50257, n_embd=768, n_layer=12, n_head=12, n_positions=1024, batch=4, l_in=16)}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md); ~6300 achievable


def algorithmic_bytes_per_step(cfg, quantize, B, ctx_mean, tp, groupsize=128):
    """W_q + W_sz + W_dense + KV_read + KV_write per rank per decode step."""
    E, I, L, V = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.vocab_size
    H, Hkv = cfg.num_attention_heads, cfg.num_key_value_heads
    D = E // H
    gated = getattr(cfg, "model_type", "llama") == "llama"  # GPT-BigCode: c_fc -> gelu -> c_proj, no gate
    mats = [(E, (H + 2 * Hkv) * D), (E, E), (E, (2 if gated else 1) * I), (I, E)]
    params = sum(k * n for k, n in mats)
    if quantize == "gptq":
        w = L * (params / 2 + sum((k / groupsize) * n * 2.5 for k, n in mats)) / tp
    else:
        w = L * params * 2 / tp
    head = V * E * 2 / tp
    kv_tok = L * 2 * (Hkv / tp if Hkv >= tp else 1) * D * 2
    kv_read = B * ctx_mean * kv_tok
    kv_write = B * kv_tok
    return {"weights": w, "head": head, "kv_read": kv_read, "kv_write": kv_write,
            "total": w + head + kv_read + kv_write}


def cpu_baseline(cfg, quantize, B, ctx, groupsize=128):
    """The CPU oracle's arithmetic (oracle/ops_ref.py + oracle/llama_ref.py: the fp32 restatement of the reference's
    CPU causal_lm forward) timed on this host for ONE decoder layer of one decode step at the full batch and
    context, batched over sequences, then scaled to all layers (+ measured lm_head).  GPTQ weights are dequantised
    to fp32 first, exactly what the reference must do on CPU (server.py:290-291).  A reported baseline only."""
    from oracle import ops_ref

    threads = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(threads)
    bigcode = hasattr(cfg, "n_inner")
    E, H = cfg.hidden_size, cfg.num_attention_heads
    I = cfg.n_inner if bigcode else cfg.intermediate_size
    Hkv = 1 if bigcode else cfg.num_key_value_heads
    D = E // H
    g = torch.Generator().manual_seed(0)

    def weight(k, n):
        if quantize == "gptq":
            qw, qz, sc, gi = ops_ref.make_gptq_tensors(k, n, groupsize, seed=k + n)
            return ops_ref.gptq_dequant(qw, qz, sc, gi, groupsize)
        return torch.randn(k, n, generator=g) * 0.02

    wo = weight(E, E)
    norm_w = torch.ones(E)
    if not bigcode:
        wq, wgu, wd = weight(E, (H + 2 * Hkv) * D), weight(E, 2 * I), weight(I, E)
        Kc = torch.randn(B, Hkv, ctx, D, generator=g)
        Vc = torch.randn(B, Hkv, ctx, D, generator=g)
    x = torch.randn(B, E, generator=g)
    cos, sin = ops_ref.rope_tables(D, 10000.0, ctx, torch.float32)
    pos = torch.full((B,), ctx - 1, dtype=torch.int64)

    def layer(x, residual):
        h, residual = ops_ref.rmsnorm_residual(x, residual, norm_w, cfg.rms_norm_eps)
        qkv = h @ wq
        q = ops_ref.apply_rope(qkv[:, :H * D].reshape(B, H, D), cos[pos], sin[pos])
        k = ops_ref.apply_rope(qkv[:, H * D:(H + Hkv) * D].reshape(B, Hkv, D), cos[pos], sin[pos])
        v = qkv[:, (H + Hkv) * D:].reshape(B, Hkv, D)
        Kc[:, :, -1] = k
        Vc[:, :, -1] = v
        G = H // Hkv
        qg = q.reshape(B, Hkv, G, D)
        s = torch.einsum("bhgd,bhtd->bhgt", qg, Kc) * (D ** -0.5)
        p = torch.softmax(s, dim=-1)
        o = torch.einsum("bhgt,bhtd->bhgd", p, Vc).reshape(B, H * D)
        h2, residual = ops_ref.rmsnorm_residual(o @ wo, residual, norm_w, cfg.rms_norm_eps)
        return ops_ref.silu_mul(h2 @ wgu, I) @ wd, residual

    if bigcode:
        # GPT-BigCode layer (flash_santacoder_modeling.py:226-330 as restated in oracle/santacoder_ref.py): LayerNorm,
        # c_attn with ONE k / v head, attention of the H query heads over it, c_proj, LayerNorm, c_fc -> gelu -> c_proj
        wqkv, wfc, wpr = weight(E, (H + 2) * D), weight(E, I), weight(I, E)
        ln_b = torch.zeros(E)
        Kc1 = torch.randn(B, ctx, D, generator=g)
        Vc1 = torch.randn(B, ctx, D, generator=g)
        tanh = getattr(cfg, "activation_function", "gelu_pytorch_tanh") != "gelu"

        def layer(x, residual):  # noqa: F811
            h, residual = ops_ref.layernorm_residual(x, residual, norm_w, ln_b, cfg.layer_norm_epsilon)
            qkv = h @ wqkv
            q = qkv[:, :H * D].reshape(B, H, D)
            Kc1[:, -1] = qkv[:, H * D:(H + 1) * D]
            Vc1[:, -1] = qkv[:, (H + 1) * D:]
            sc = torch.einsum("bhd,btd->bht", q, Kc1) * (D ** -0.5)
            o = torch.einsum("bht,btd->bhd", torch.softmax(sc, dim=-1), Vc1).reshape(B, H * D)
            h2, residual = ops_ref.layernorm_residual(o @ wo, residual, norm_w, ln_b, cfg.layer_norm_epsilon)
            return ops_ref.gelu(h2 @ wfc, tanh) @ wpr, residual

    layer(x, x)  # warm-up
    t0 = time.perf_counter()
    reps = 0
    while reps < 96 and (reps < 2 or time.perf_counter() - t0 < 12.0):  # ~10 s of CPU work (three steps' worth of layers)
        layer(x, x)
        reps += 1
    t_layer = (time.perf_counter() - t0) / reps
    head_w = torch.randn(E, cfg.vocab_size, generator=g)
    t1 = time.perf_counter()
    ops_ref.greedy(x @ head_w)
    t_head = time.perf_counter() - t1
    step_s = t_layer * cfg.num_hidden_layers + t_head
    return {"value": round(B / step_s, 2), "unit": "tokens/s", "cores": threads, "kind": "port",
            "ms_per_step": round(step_s * 1e3, 1),
            # what this number is: OUR fp32 restatement (oracle/), one layer timed and extrapolated — not the reference's
            # own process (its CPU path needs its pinned transformers / torch: tests/golden/make_*fixtures.py)
            "sample": f"oracle/ port, not the reference binary: one {'GPT-BigCode' if bigcode else 'Llama'} decoder layer of "
                      f"one decode step at B={B}, ctx={ctx}, fp32, whole (unsharded) model, {reps} repetitions, "
                      f"extrapolated x{cfg.num_hidden_layers} layers + measured lm_head/greedy"}


def fused_rope_launches_per_step(lm):
    """Layers whose qkv projection runs the rotary embedding + cache write in its GEMM epilogue at this batch size."""
    n = 0
    for layer in getattr(lm.model.model, "layers", []):
        att = getattr(layer, "self_attn", None)
        lin = getattr(getattr(att, "query_key_value", None), "linear", None)
        if getattr(lin, "rope_handle", None) is not None:
            n += 1
    return n


def pmc_traffic(config, B, ctx_mean):
    """HBM bytes per attention launch from the committed rocprofv3 counter passes (tools/profile_round.sh: FETCH_SIZE
    and WRITE_SIZE in separate --pmc runs of this same command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    for 16 B/lane streaming reads on gfx950).  Counters cannot be collected inside the timed run, so the figure is read
    from profiles/ and only reported for the workload it was measured on; otherwise null."""
    if (config, B, ctx_mean) != ("llama2-7b-gptq", 32, 1024):
        return None, None
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_attn_traffic.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    if d.get("fetch_bytes_per_launch_corrected") is None:
        return None, None
    return int(d["fetch_bytes_per_launch_corrected"] + (d.get("write_bytes_per_launch_uncalibrated") or 0)), \
        "profiles/" + os.path.basename(files[-1])


def pmc_gemm_traffic(config, B, ctx_mean):
    """HBM bytes per int4-GEMM launch (average over the four shapes of a layer) from the same counter passes
    (profiles/r*_gemm_traffic.json); null for workloads it was not collected on."""
    if (config, B, ctx_mean) != ("llama2-7b-gptq", 32, 1024):
        return None, None
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_gemm_traffic.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    if d.get("fetch_bytes_per_launch_corrected") is None:
        return None, None
    return int(d["fetch_bytes_per_launch_corrected"] + (d.get("write_bytes_per_launch_uncalibrated") or 0)), \
        "profiles/" + os.path.basename(files[-1])


def bench_causal_lm_cpu(args):
    """BASELINE config 1: GPT-2 small (124 M, random init, fp32) through CausalLMBatch / CausalLM.generate_token on the
    hf_transformers engine, CPU, batch 4, prompts of 16 tokens.  A step = one NextToken-equivalent generate_token call.
    `cpu_baseline` = the oracle's fp32 GPT-2 restatement driven through the same greedy loop."""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import GPT2Config, GPT2LMHeadModel, PreTrainedTokenizerFast

    from tgis_amd.inference_engine.hf_transformers import InferenceEngine
    from tgis_amd.models.causal_lm import CausalLM
    from tgis_amd.pb import generate_pb2 as pb2

    c = CPU_CONFIGS[args.config]
    B, L_in = args.batch or c["batch"], c["l_in"]
    K, W = args.steps, args.warmup
    os.environ["CUDA_VISIBLE_DEVICES"] = os.environ["HIP_VISIBLE_DEVICES"] = ""  # this config is the CPU path
    threads = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(threads)
    torch.manual_seed(1234)
    hf = GPT2LMHeadModel(GPT2Config(vocab_size=c["vocab_size"], n_embd=c["n_embd"], n_layer=c["n_layer"], n_head=c["n_head"],
                                    n_positions=c["n_positions"], attn_pdrop=0.0, resid_pdrop=0.0, embd_pdrop=0.0,
                                    pad_token_id=0, bos_token_id=1, eos_token_id=2))
    vocab = {"<pad>": 0, "<s>": 1, "</s>": 2, **{f"t{i}": i for i in range(3, c["vocab_size"])}}
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<pad>"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    tok = PreTrainedTokenizerFast(tokenizer_object=tk, eos_token="</s>", bos_token="<s>", unk_token="<pad>",
                                  pad_token="<pad>", padding_side="left", truncation_side="left")
    lm = CausalLM("gpt2-small-random", None, "hf_transformers", torch.float32, None,
                  engine=InferenceEngine(None, None, torch.float32, None, None, 1024, preloaded=hf, tokenizer=tok))
    assert lm.device.type == "cpu"
    g = torch.Generator().manual_seed(1234)
    prompts = [torch.randint(3, c["vocab_size"], (L_in,), generator=g).tolist() for _ in range(B)]
    reqs = [pb2.Request(id=i, inputs=" ".join(f"t{t}" for t in p), input_length=L_in, truncate=True,
                        max_output_length=W + K + 1) for i, p in enumerate(prompts)]
    with lm.context_manager():
        batch, errs = lm.batch_type.from_pb(pb2.Batch(id=0, requests=reqs), tok, lm.dtype, lm.device, lm.word_embeddings,
                                            None, lm.use_position_ids)
        assert not errs
        t0 = time.perf_counter()
        lm.generate_token(batch, first=True)
        prefill_ms = (time.perf_counter() - t0) * 1e3
        ids = []
        for _ in range(W):
            lm.generate_token(batch)
        step_ms = []
        t0 = time.perf_counter()
        for _ in range(K):
            ts = time.perf_counter()
            toks = lm.generate_token(batch)[0]
            step_ms.append((time.perf_counter() - ts) * 1e3)
            ids.append([t.token_id for t in toks])
        elapsed = time.perf_counter() - t0
    out = {"metric": f"decode tokens/sec (GPT-2 small fp32 CPU causal_lm, batch {B}) + p50 step latency",
           "value": round(B * K / elapsed, 2), "unit": "tokens/s", "n_gpus": 0, "steps": K, "warmup": W,
           "ms_per_step": round(elapsed / K * 1e3, 4),
           "p50_step_ms": round(sorted(step_ms)[len(step_ms) // 2], 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic (seeded random-init GPT-2 small, seeded token ids)",
           "config": {"workload": f"gpt2-small fp32 CPU causal_lm (padded batch), B={B}, L_in={L_in}, prefill + {W} warm-up + "
                                  f"{K} timed decode steps, greedy", "global_batch": B, "seq_len": L_in + W + K,
                      "parallelism": "cpu", "threads": threads, "prefill_ms": round(prefill_ms, 2)},
           "roofline": None}
    if not args.no_cpu_baseline:
        from oracle.gpt2_ref import GPT2Ref

        class _Cfg:
            n_embd, n_head, n_layer, layer_norm_epsilon, activation_function = c["n_embd"], c["n_head"], c["n_layer"], 1e-5, "gelu_new"

        sd = {k: v for k, v in hf.state_dict().items()}
        ref = GPT2Ref(_Cfg, sd)
        n = min(8, K)
        t0 = time.perf_counter()
        want = ref.generate_greedy(prompts, n + 1)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(B * (n + 1) / dt, 2), "unit": "tokens/s", "cores": threads, "kind": "port",
                               "sample": f"oracle/gpt2_ref.py greedy loop, prefill + {n} decode steps, B={B}, fp32"}
    print(json.dumps(out), flush=True)


def bench_churn(args, lm, tok, B, ctx_mean, make_batch_pb):
    """the decode path under the membership changes the reference's servicer makes between steps
    (server.py:183-231 -> flash_causal_lm.py:196-353 concatenate / prune; router side batcher.rs:497-518).  Same weights and
    kernels as the headline line; every 4-8 decode steps 1-2 requests complete (prune) and 0-2 new ones are prefilled and
    concatenated, the batch wandering in [3B/4, B].  The same seeded schedule runs on (A) a pristine pool, (B) a pool aged by
    random allocations and frees (B2: in pieces of one or two pages, the finest fragmentation) and (C) the aged pool with a decode graph captured per exact batch size instead of per
    bucket.  (Round 6 also ran the aged pool under a channel-aware free list, one heap per page-id residue class: p50 4.149 vs
    4.144 ms for this allocator — profiles/r06a_churn_cfg3_classes.json, r06_page_classes.log; not kept.)  Reported per run: p50 / p99 / mean of the host time
    of the decode steps (generate_token incl. the copy of the ids), tokens/s, graph captures and what each cost."""
    import random

    import tgis_amd.models.flash_causal_lm as fcl

    cache = lm.kv_cache
    lo_b = max(1, 3 * B // 4)
    steps = args.churn_steps
    max_new = 256

    def age_pool(seed, largest):
        rng = random.Random(seed)
        held = []
        while cache.free_pages > cache.num_pages // 4:
            held.append(cache.alloc(rng.randrange(1, largest + 1)))
        rng.shuffle(held)
        while cache.free_pages < cache.num_pages // 2:
            cache.free(held.pop())
        return held

    def prefill(n, rng, first_id, batch_id):
        lens = [rng.randrange(max(1, ctx_mean - 192), max(2, ctx_mean - 32)) for _ in range(n)]
        pb = make_batch_pb(lens, max_new=max_new, first_request_id=first_id, batch_id=batch_id)
        b, errs = lm.batch_type.from_pb(pb, tok, lm.dtype, lm.device, lm.word_embeddings, None, True)
        assert not errs
        lm.generate_token(b, first=True, for_concat=True)
        return b

    def run(name, aged, buckets):
        # aged = largest piece of the random allocations that fragment the pool (48: holes of up to a sequence's worth of
        # pages; 2: holes of one or two pages — the finest fragmentation a pool can have), 0 = pristine
        assert cache.free_pages == cache.num_pages
        held = age_pool(11, aged) if aged else []
        fcl.GRAPH_BUCKETS = buckets
        lm._graphs.clear()
        lm.graph_captures.clear()
        lm.graph_pool = torch.cuda.graph_pool_handle()
        rng = random.Random(1234)
        with lm.context_manager():
            batch = prefill(B, rng, 0, 0)
            next_id, next_event = B, rng.randrange(4, 9)
            step_ms, sizes, ctxs, member_ms = [], [], [], []
            for s in range(steps):
                if s == next_event:
                    t0 = time.perf_counter()
                    ids = [r.id for r in batch.requests]
                    n_done = min(rng.randrange(1, 3), len(ids) - 1)
                    batch = lm.batch_type.prune(batch, rng.sample(ids, n_done))
                    n_new = rng.randrange(0, 3)
                    n_new = max(n_new, lo_b - len(batch))
                    n_new = min(n_new, B - len(batch))
                    if n_new > 0:
                        nb = prefill(n_new, rng, next_id, 1 + s)
                        next_id += n_new
                        batch = lm.batch_type.concatenate([batch, nb])
                    torch.cuda.synchronize()
                    member_ms.append((time.perf_counter() - t0) * 1e3)
                    next_event = s + rng.randrange(4, 9)
                t0 = time.perf_counter()
                lm.generate_token(batch)
                step_ms.append((time.perf_counter() - t0) * 1e3)
                sizes.append(len(batch))
                ctxs.append(sum(batch.input_lengths) / len(batch))
            torch.cuda.synchronize()
            batch.release()
        for h in held:
            cache.free(h)
        caps = list(lm.graph_captures)
        srt = sorted(step_ms)
        # steps that captured a graph are reported separately AND are part of p99 / mean: a serving step pays them
        return {"run": name, "pool": f"aged (pieces of 1-{aged} pages)" if aged else "pristine",
                "graph_rows": "bucket" if buckets else "exact batch size", "decode_steps": steps,
                "p50_ms": round(srt[len(srt) // 2], 4), "p99_ms": round(srt[min(len(srt) - 1, int(len(srt) * 0.99))], 4),
                "mean_ms": round(sum(step_ms) / len(step_ms), 4), "max_ms": round(srt[-1], 3),
                "tokens_per_s": round(sum(sizes) / (sum(step_ms) * 1e-3), 1),
                "mean_batch": round(sum(sizes) / len(sizes), 2), "mean_ctx": round(sum(ctxs) / len(ctxs), 1),
                "membership_events": len(member_ms),
                "membership_ms_mean": round(sum(member_ms) / max(len(member_ms), 1), 2),
                "graph_captures": len(caps), "graph_capture_ms": [round(c[2], 1) for c in caps],
                "graph_keys": sorted({(c[0], c[1]) for c in caps})}

    runs = [run("A", 0, True), run("B", 48, True), run("B2", 2, True), run("C", 48, False)]
    fcl.GRAPH_BUCKETS = True
    a, b, b2, c = runs
    out = {"metric": f"decode ms/step under batch churn ({args.config}, B in [{lo_b}, {B}], ctx ~{ctx_mean})",
           "unit": "ms", "higher_is_better": False, "n_gpus": 1, "data": "synthetic",
           "config": {"workload": f"{args.config} decode under churn: {steps} decode steps, a prune + 0-2 request Prefill + "
                                  f"concatenate every 4-8 steps, seeded schedule, greedy", "pool_pages": cache.num_pages,
                      "page_bytes": cache.num_kv_heads * 32 * cache.head_dim * 2},
           "runs": runs,
           "aged_over_pristine_p50": round(b["p50_ms"] / a["p50_ms"], 4),
           "finely_aged_over_pristine_p50": round(b2["p50_ms"] / a["p50_ms"], 4),
           "p99_over_p50": {r["run"]: round(r["p99_ms"] / r["p50_ms"], 3) for r in runs}}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="llama2-7b-gptq", choices=sorted(list(CONFIGS) + list(CPU_CONFIGS)))
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--ctx", type=int, default=None, help="mean context length over the timed steps")
    ap.add_argument("--blocks", type=int, default=3, help="timed blocks of --steps steps each; the median block is reported")
    ap.add_argument("--dump-steps", action="store_true", help="stderr: the host-side time of every timed step, per block")
    ap.add_argument("--churn", action="store_true",
                    help="the path the way the router drives it: requests leave (prune) and join (Prefill + concatenate) every "
                         "4-8 steps, B wandering in [3B/4, B], on a pristine and on an aged page pool; prints ONE JSON line of "
                         "its own (p50 / p99 of the decode steps, graph captures) — not the driver's contract line")
    ap.add_argument("--churn-steps", type=int, default=240)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()
    if args.config in CPU_CONFIGS:
        return bench_causal_lm_cpu(args)

    rank = int(os.getenv("RANK", "0"))
    world = int(os.getenv("WORLD_SIZE", "1"))
    os.environ.setdefault("TGIS_DIST_TIMEOUT_S", "600")  # cold boxes: first imports and weight set-up skew the ranks
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started bare (`python bench.py --gpus N`): become the launcher of N ranks, one per GPU, on this node
        import socket

        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                                  f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port",
                                  str(port), os.path.abspath(__file__), *sys.argv[1:]])
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU: the hot path has no CPU fallback"

    from tgis_amd import native

    if not os.path.exists(native.LIB_PATH):  # a checkout without built artefacts
        if rank == 0:
            import __graft_entry__

            __graft_entry__.build()
        if world > 1:
            import time as _t

            while not os.path.exists(native.LIB_PATH):  # the other ranks wait for rank 0's build
                _t.sleep(1.0)
            _t.sleep(2.0)
    from tgis_amd.inference_engine.synthetic import BigCodeConfig, InferenceEngine, bigcode_tensors, llama_tensors
    from tgis_amd.models.custom_modeling.flash_llama_modeling import LlamaConfig
    from tgis_amd.models.flash_causal_lm import FlashCausalLM
    from tgis_amd.testing import SyntheticTokenizer, make_batch_pb
    from tgis_amd.utils.kv_cache import PagedKVCache

    kw, quantize, dtype_s, B, ctx_mean = CONFIGS[args.config]
    B = args.batch or B
    ctx_mean = args.ctx or ctx_mean
    bigcode = "n_inner" in kw
    cfg = BigCodeConfig(**kw) if bigcode else LlamaConfig(**kw)
    dtype = getattr(torch, dtype_s)
    K, W = args.steps, args.warmup
    # context grows by one per step; centre the timed steps on ctx_mean
    L_in = max(1, ctx_mean - W - K // 2 - 1)
    total_len = L_in + W + K + 8

    device = torch.device("cuda", rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    if bigcode:
        tensors = bigcode_tensors(cfg, seed=1234, device=device, dtype=dtype)
    else:
        tensors = llama_tensors(cfg, quantize, seed=1234, device=device, dtype=dtype)
    tok = SyntheticTokenizer(cfg.vocab_size)
    eng = InferenceEngine(tensors, cfg, dtype, quantize, tokenizer=tok)
    del tensors
    pages = B * PagedKVCache.pages_for(total_len) + 8
    if args.churn:
        pages = 4 * B * PagedKVCache.pages_for(ctx_mean + 256) + 64  # room for an aged pool's holes
    lm = FlashCausalLM("synthetic", None, "synthetic", dtype, quantize, engine=eng, kv_cache_pages=pages)
    torch.cuda.empty_cache()
    tp = eng.world_size
    graphs_used = bool(lm.use_graphs)

    def sync():
        torch.cuda.synchronize()
        if tp > 1:
            torch.distributed.barrier()

    def fresh_batch():
        pb = make_batch_pb([L_in] * B, max_new=W + K + 8)
        batch, errs = lm.batch_type.from_pb(pb, tok, lm.dtype, lm.device, lm.word_embeddings, None, True)
        assert not errs
        lm.generate_token(batch, first=True)  # prefill (untimed): fills the KV pages
        return batch

    from tgis_amd.utils import graph_segments

    if args.churn:
        assert tp == 1, "--churn is a single-GPU measurement"
        return bench_churn(args, lm, tok, B, ctx_mean, make_batch_pb)

    with lm.context_manager():
        # Three timed blocks of exactly K steps each, every one on a fresh batch of the same shape (prefill + W warm-up
        # steps untimed, barrier + synchronize on both sides, max over ranks): the MEDIAN block is the reported
        # ms_per_step / value, the fastest and slowest are reported as ms_per_step_range.  One pass of 20 steps moved by
        # as much between boxes (and between runs on one box) as a round of kernel work did.
        blocks = []
        batch = None
        for blk in range(args.blocks):
            if batch is not None:
                batch.release()
                del batch
            batch = fresh_batch()
            for _ in range(W):
                lm.generate_token(batch)
            sync()
            blk_ms = []
            t0 = time.perf_counter()
            for _ in range(K):
                ts = time.perf_counter()
                lm.generate_token(batch)
                blk_ms.append((time.perf_counter() - ts) * 1e3)
            sync()
            blk_elapsed = time.perf_counter() - t0
            if tp > 1:
                t = torch.tensor([blk_elapsed], device=device, dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                blk_elapsed = float(t.item())
            blocks.append((blk_elapsed, blk_ms))
        order = sorted(range(len(blocks)), key=lambda i: blocks[i][0])
        elapsed, step_ms = blocks[order[len(order) // 2]]
        block_ms_per_step = [round(b[0] / K * 1e3, 4) for b in blocks]
        if args.dump_steps and rank == 0:
            for i, b in enumerate(blocks):
                print(f"block {i}: " + " ".join(f"{v:.3f}" for v in b[1]), file=sys.stderr)
        coll_us = []
        if tp > 1:
            # eagerly issued collectives bracketed with events in a pass of their own (timing them inside one of
            # the timed blocks slowed that block only, and described a block that was not the reported median)
            batch.release()
            batch = fresh_batch()
            for _ in range(W):
                lm.generate_token(batch)
            sync()
            graph_segments.time_collectives(True)
            for _ in range(K):
                lm.generate_token(batch)
            sync()
            coll_us = graph_segments.collective_times_us()
            graph_segments.time_collectives(False)
        ctx_timed_mean = L_in + 1 + W + (K - 1) / 2.0  # keys attended per step, averaged over the timed steps
        graphs_kept = bool(lm.use_graphs)
        last = next(iter(lm._graphs.values()), None)
        logits_finite = bool(torch.isfinite(last.logits).all()) if last is not None and last.logits is not None else None

        # the captured step replayed back to back (same inputs again; nothing reads this batch afterwards): GPU time of a
        # step without the host's per-step work (three small copies in, token bookkeeping, one copy out)
        graph_ms = None
        if tp == 1 and last is not None and isinstance(getattr(last, "graph", None), torch.cuda.CUDAGraph):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            last.graph.replay()
            e0.record()
            for _ in range(10):
                last.graph.replay()
            e1.record()
            sync()
            graph_ms = e0.elapsed_time(e1) / 10.0

        roofline = roofline_gemm = None
        if not args.no_roofline:  # every rank runs it (the forward contains collectives when tp > 1)
            # Instrumented pass over the same workload: eager launches (HIP events cannot bracket nodes of a
            # replayed graph), event pairs recorded by libtgis_hip.so on the launch stream around every attention
            # launch.  Its own timed region of K steps, same batch shape and context range.
            batch.release()
            del batch
            lm.use_graphs = False
            batch = fresh_batch()
            for _ in range(W):
                lm.generate_token(batch)
            sync()
            native.timing_reset()
            native.timing_enable(True)
            for _ in range(K):
                lm.generate_token(batch)
            sync()
            native.timing_enable(False)
            n_attn, ms_attn = native.timing_read(native.OP_ATTN)
            n_gemm, ms_gemm = native.timing_read(native.OP_GPTQ_GEMM if quantize == "gptq" else native.OP_DENSE_GEMM)
            Hkv_rank = max(1, cfg.num_key_value_heads // tp)
            D = cfg.hidden_size // cfg.num_attention_heads
            bytes_per_launch = B * ctx_timed_mean * 2 * Hkv_rank * D * 2 + B * 2 * Hkv_rank * D * 2
            avg_s = ms_attn * 1e-3 / max(n_attn, 1)
            achieved = bytes_per_launch / avg_s / 1e9
            # the committed counter pass is of the single-GPU workload: a rank of a TP group streams other bytes -> null
            traffic, traffic_src = pmc_traffic(args.config, B, ctx_mean) if tp == 1 else (None, None)
            roofline = {"bound": "hbm", "kernel": "attn_paged_kernel (decode)", "achieved": round(achieved, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": traffic,
                        # counters cannot be collected inside a timed run: the figure is the committed --pmc pass of this
                        # command (tools/profile_round.sh), not a measurement of THIS run
                        "traffic_source": traffic_src and f"{traffic_src} (committed rocprofv3 --pmc pass, not this run)",
                        # avg_launch_us comes from a second, eager pass of the same workload (HIP events around every
                        # launch); ms_per_step is the captured-graph pass
                        "pass": "eager, HIP events on the launch stream", "launches": int(n_attn),
                        "avg_launch_us": round(avg_s * 1e6, 2),
                        "algorithmic_bytes_per_launch": int(bytes_per_launch),
                        "gemm_launches": int(n_gemm), "gemm_avg_launch_us": round(ms_gemm * 1e3 / max(n_gemm, 1), 2)}
            # the weight-streaming GEMMs (the kernels furthest below their roofline): algorithmic bytes = the packed
            # weights + scales/zeros they stream (SURVEY.md §8d W_q + W_sz; dense models: W_dense incl. the head),
            # averaged over the launches of one step
            abw = algorithmic_bytes_per_step(cfg, quantize, B, ctx_timed_mean, tp)
            gemm_bytes_step = abw["weights"] + (0 if quantize == "gptq" else abw["head"])
            per_step = n_gemm / max(K, 1)
            if n_gemm:
                g_avg_s = ms_gemm * 1e-3 / n_gemm
                g_bytes = gemm_bytes_step / max(per_step, 1)
                g_traffic, g_traffic_src = pmc_gemm_traffic(args.config, B, ctx_mean) if tp == 1 else (None, None)
                roofline_gemm = {"bound": "hbm", "kernel": ("gptq_wide_kernel / gptq_gemm_kernel" if quantize == "gptq"
                                                            else "dense_gemm_kernel") + f" ({per_step:.0f} launches per step)",
                                 "achieved": round(g_bytes / g_avg_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(g_bytes / g_avg_s / 1e9 / HBM_PEAK_GBS, 4),
                                 "traffic": g_traffic,
                                 "traffic_source": g_traffic_src and f"{g_traffic_src} (committed rocprofv3 --pmc pass, not this run)",
                                 "pass": "eager, HIP events on the launch stream",
                                 "launches": int(n_gemm), "avg_launch_us": round(g_avg_s * 1e6, 2),
                                 "algorithmic_bytes_per_launch": int(g_bytes),
                                 # one GEMM launch per layer may carry the rotary embedding + cache write in its epilogue
                                 # (round 3: the former rope_kv_write launch is inside this average now)
                                 "launches_with_rope_epilogue_per_step": fused_rope_launches_per_step(lm)}

    toks_per_s = B * K / elapsed
    ab = algorithmic_bytes_per_step(cfg, quantize, B, ctx_timed_mean, tp)
    step_roof_ms = ab["total"] / (HBM_PEAK_GBS * 1e9) * 1e3
    out = {
        "metric": ("decode tokens/sec (Llama-7B int4 GPTQ, batch 32, ctx 1024) + p50 step latency"
                   if (args.config, B, ctx_mean) == ("llama2-7b-gptq", 32, 1024)
                   else f"decode tokens/sec ({args.config}, batch {B}, ctx {ctx_mean}) + p50 step latency"),
        "value": round(toks_per_s, 2), "unit": "tokens/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(elapsed / K * 1e3, 4),
        # timed blocks of `steps` steps each (fresh batch, same shape); ms_per_step / value are the median block
        "timed_blocks": len(block_ms_per_step), "ms_per_step_blocks": block_ms_per_step,
        "ms_per_step_range": [min(block_ms_per_step), max(block_ms_per_step)],
        "p50_step_ms": round(sorted(step_ms)[len(step_ms) // 2], 4),
        "graph_ms_per_step": None if graph_ms is None else round(graph_ms, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f16" if dtype == torch.float16 else "bf16",
        "data": "synthetic (seeded weights at the real shapes; KV from a real prefill of seeded token ids)",
        "config": {"workload": f"{args.config} decode, B={B}, mean ctx {ctx_timed_mean:.1f} "
                               f"(L_in={L_in}, {W} warm-up + {K} timed steps), greedy",
                   "global_batch": B, "seq_len": int(round(ctx_timed_mean)), "parallelism": f"tp{tp}",
                   "weights": "int4 GPTQ g128" if quantize == "gptq" else dtype_s,
                   "logits_finite": logits_finite,
                   "hip_graph": (lm.graph_mode if tp > 1 else True) if (graphs_used and graphs_kept) else False,
                   "rccl_world": (torch.distributed.get_world_size() if tp > 1 else 1),
                   "collective_backend": (torch.distributed.get_backend() if tp > 1 else None),
                   # eagerly issued collectives only (segments / eager mode); inside one captured graph they cannot be
                   # bracketed and are part of ms_per_step
                   "collectives_per_step": (round(len(coll_us) / K, 1) if coll_us else None),
                   "collective_avg_us": (round(sum(coll_us) / len(coll_us), 2) if coll_us else None),
                   # what ONE rank streams per step (weights / tp, its kv heads' pages, the head / tp)
                   "algorithmic_bytes_per_rank_step": int(ab["total"])},
        "step_roofline": {"algorithmic_bytes_per_step": int(ab["total"]), "ms_at_hbm_peak": round(step_roof_ms, 4),
                          "frac_of_hbm_peak": round(step_roof_ms / (elapsed / K * 1e3), 4)},
    }
    if roofline is not None:
        out["roofline"] = roofline
    if roofline_gemm is not None:
        out["roofline_gemm"] = roofline_gemm
    if rank == 0 and not args.no_cpu_baseline:
        # next to every line (N > 1 too: rank 0 times it while the others wait at the final barrier): the whole,
        # unsharded model on this host's cores
        out["cpu_baseline"] = cpu_baseline(cfg, quantize, B, int(round(ctx_timed_mean)))
    if rank == 0:
        print(json.dumps(out), flush=True)
    if tp > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
