"""Prefill (first generate_token) time for the cfg3 model: B sequences of L tokens -> tokens/s and ms (TTFT side)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
import torch
import bench
from tgis_amd.inference_engine.synthetic import InferenceEngine, llama_tensors
from tgis_amd.models.custom_modeling.flash_llama_modeling import LlamaConfig
from tgis_amd.models.flash_causal_lm import FlashCausalLM
from tgis_amd.testing import SyntheticTokenizer, make_batch_pb
from tgis_amd.utils.kv_cache import PagedKVCache

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1004
kw, quantize, dtype_s, _, _ = bench.CONFIGS["llama2-7b-gptq"]
cfg = LlamaConfig(**kw)
dev = torch.device("cuda:0")
tensors = llama_tensors(cfg, quantize, seed=1234, device=dev, dtype=torch.float16)
tok = SyntheticTokenizer(cfg.vocab_size)
eng = InferenceEngine(tensors, cfg, torch.float16, quantize, tokenizer=tok)
del tensors
lm = FlashCausalLM("synthetic", None, "synthetic", torch.float16, quantize, engine=eng,
                   kv_cache_pages=B * PagedKVCache.pages_for(L + 8) + 8)
times = []
with lm.context_manager():
    for rep in range(4):
        pb = make_batch_pb([L] * B, max_new=4)
        batch, errs = lm.batch_type.from_pb(pb, tok, lm.dtype, lm.device, lm.word_embeddings, None, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lm.generate_token(batch, first=True)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        batch.release()
best = min(times[1:])
print(f"prefill B={B} L={L}: {best*1e3:.1f} ms  {B*L/best/1e3:.1f} k tok/s  (runs: {[round(t*1e3,1) for t in times]})")
