"""Decode attention at multi-query / wide-group shapes (Starcoder: 48 q heads on 1 kv head; Llama-70B TP shard)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
import microbench as mb
for B, H, Hkv, ctx in ((32, 48, 1, 4096), (32, 48, 1, 1024), (8, 48, 1, 8192), (32, 32, 1, 2048), (32, 64, 8, 2048), (64, 8, 1, 2048)):
    for ns in (None, 4, 8, 16, 32):
        try:
            mb.bench_attn(B, H, Hkv, 128, ctx, sets=4, ns=ns)
        except Exception as e:
            print("failed", B, H, Hkv, ctx, ns, e)
