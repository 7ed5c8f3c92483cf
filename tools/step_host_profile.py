"""Where the host's share of a decode step goes (TinyLlama shapes, plain greedy): wall-clock stamps around the parts of
FlashCausalLM.generate_token, averaged over the timed steps.  GPU only.   python tools/step_host_profile.py [config]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
import bench  # noqa: E402
from tgis_amd import native  # noqa: E402
from tgis_amd.inference_engine.synthetic import BigCodeConfig, InferenceEngine, bigcode_tensors, llama_tensors  # noqa: E402
from tgis_amd.models.custom_modeling.flash_llama_modeling import LlamaConfig  # noqa: E402
from tgis_amd.models import flash_causal_lm as fcl  # noqa: E402
from tgis_amd.testing import SyntheticTokenizer, make_batch_pb  # noqa: E402
from tgis_amd.utils.kv_cache import PagedKVCache  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "tinyllama-1.1b"
kw, quantize, dtype_s, B, ctx = bench.CONFIGS[name]
cfg = BigCodeConfig(**kw) if "n_inner" in kw else LlamaConfig(**kw)
dtype = getattr(torch, dtype_s)
dev = torch.device("cuda:0")
tensors = bigcode_tensors(cfg, seed=1, device=dev, dtype=dtype) if "n_inner" in kw else llama_tensors(cfg, quantize, seed=1, device=dev, dtype=dtype)
tok = SyntheticTokenizer(cfg.vocab_size)
eng = InferenceEngine(tensors, cfg, dtype, quantize, tokenizer=tok)
del tensors
K, W = 40, 5
L_in = ctx - W - K // 2
lm = fcl.FlashCausalLM("synthetic", None, "synthetic", dtype, quantize, engine=eng, kv_cache_pages=B * PagedKVCache.pages_for(L_in + W + K + 8) + 8)

stamps = {}


def wrap(obj, attr, label):
    f = getattr(obj, attr)

    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        stamps[label] = stamps.get(label, 0.0) + time.perf_counter() - t
        return r
    setattr(obj, attr, g)


wrap(fcl.FlashCausalLMBatch, "grow_pages", "grow_pages")
wrap(fcl._DecodeGraph, "run", "graph.run (stage + replay call)")
wrap(fcl._DecodeGraph, "fetch_greedy", "fetch_greedy (enqueue copy)")
wrap(fcl._DecodeGraph, "_read_host", "read_host (wait + tolist)")
wrap(native, "decode_advance", "decode_advance (launch)")
wrap(lm, "_process_new_tokens", "_process_new_tokens (all)")
wrap(lm, "_decode_forward", "_decode_forward (all)")

with lm.context_manager():
    pb = make_batch_pb([L_in] * B, max_new=W + K + 8)
    batch, _ = lm.batch_type.from_pb(pb, tok, lm.dtype, lm.device, lm.word_embeddings, None, True)
    lm.generate_token(batch, first=True)
    for _ in range(W):
        lm.generate_token(batch)
    torch.cuda.synchronize()
    stamps.clear()
    t0 = time.perf_counter()
    for _ in range(K):
        lm.generate_token(batch)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
print(f"{name}: {el / K * 1e3:.4f} ms per step; host time per step inside (us):")
for k, v in stamps.items():
    print(f"  {k:40s} {v / K * 1e6:8.1f}")
