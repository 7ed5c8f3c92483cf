"""Decode attention on small problems (TinyLlama cfg2 and TP shards): is the key split + combine launch worth it?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
import microbench as mb
for B, H, Hkv, D, ctx in ((16, 32, 4, 64, 512), (32, 4, 4, 128, 1024), (32, 8, 8, 128, 1024), (32, 16, 16, 128, 1024), (8, 32, 32, 128, 1024), (1, 32, 32, 128, 1024), (4, 32, 4, 64, 2048)):
    for ns in (None, 1, 2, 4, 8):
        mb.bench_attn(B, H, Hkv, D, ctx, sets=8, ns=ns)
