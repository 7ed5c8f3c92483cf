"""Decode step time at cfg3 with sampling requests (temperature / top-k / top-p / typical-p / repetition penalty,
seeded) vs plain greedy: what the next-token chooser costs on top of the forward."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "text-generation-inference_amd")]
import torch
import bench
from tgis_amd.inference_engine.synthetic import InferenceEngine, llama_tensors
from tgis_amd.models.custom_modeling.flash_llama_modeling import LlamaConfig
from tgis_amd.models.flash_causal_lm import FlashCausalLM
from tgis_amd.testing import SyntheticTokenizer, make_batch_pb
from tgis_amd.utils.kv_cache import PagedKVCache

B, K, L_in = 32, 32, 990
kw, quantize, dtype_s, _, _ = bench.CONFIGS["llama2-7b-gptq"]
cfg = LlamaConfig(**kw)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
tok = SyntheticTokenizer(cfg.vocab_size)
eng = InferenceEngine(llama_tensors(cfg, quantize, seed=1234, device=dev, dtype=torch.float16, head_scale=float(os.getenv("HEAD_SCALE", "1"))), cfg, torch.float16, quantize, tokenizer=tok)
lm = FlashCausalLM("synthetic", None, "synthetic", torch.float16, quantize, engine=eng,
                   kv_cache_pages=B * PagedKVCache.pages_for(L_in + 3 * K) + 8)


ONLY = os.getenv("CASES")



def run(label, setup):
    if ONLY and not any(c.strip() == label for c in ONLY.split(";")):
        return
    pb = make_batch_pb([L_in] * B, max_new=K + 8)
    for i, r in enumerate(pb.requests):
        setup(i, r)
    with lm.context_manager():
        batch, errs = lm.batch_type.from_pb(pb, tok, lm.dtype, lm.device, lm.word_embeddings, None, True)
        assert not errs
        lm.generate_token(batch, first=True)
        if not run.checked:
            run.checked = True
            g = next(iter(lm._graphs.values()), None)
            lm.generate_token(batch)
            g = next(iter(lm._graphs.values()))
            print("decode logits finite:", bool(torch.isfinite(g.logits).all()), "std", float(g.logits.std()), flush=True)
        for _ in range(3):
            lm.generate_token(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            lm.generate_token(batch)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / K * 1e3
        batch.release()
    print(f"{label:44s} {ms:7.3f} ms/step  {B / ms * 1e3:8.1f} tok/s", flush=True)


run.checked = False


def sampled(i, r, **kw):
    p = r.parameters
    p.temperature = kw.get("t", 0.8)
    p.seed = 1000 + i
    for k in ("top_k", "top_p", "typical_p", "repetition_penalty"):
        if k in kw:
            setattr(p, k, kw[k])


run("greedy", lambda i, r: None)
run("greedy + logprobs", lambda i, r: setattr(r.details, "logprobs", True))
run("greedy + repetition_penalty 1.2", lambda i, r: setattr(r.parameters, "repetition_penalty", 1.2))
run("sample t=0.8", lambda i, r: sampled(i, r))
run("sample t=0.8 top_k=50", lambda i, r: sampled(i, r, top_k=50))
run("sample t=0.8 top_p=0.9", lambda i, r: sampled(i, r, top_p=0.9))
run("sample t=0.8 top_k=50 top_p=0.9 rep=1.2", lambda i, r: sampled(i, r, top_k=50, top_p=0.9, repetition_penalty=1.2))
run("sample t=0.8 typical_p=0.9", lambda i, r: sampled(i, r, typical_p=0.9))
run("half greedy, half sample top_k/top_p", lambda i, r: sampled(i, r, top_k=50, top_p=0.9) if i % 2 else None)
