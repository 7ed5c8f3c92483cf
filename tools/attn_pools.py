"""Decode attention time vs the number of distinct KV pools cycled through (address-translation / cache reach)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
import microbench as mb
from tgis_amd import native as nat
dev = mb.dev
B, H, Hkv, D, ctx = 32, 32, 32, 128, 1024
pages_per = (ctx + 31) // 32
total = B * pages_per
q = torch.randn(B, H * D, device=dev).half()
ctxl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
cu = torch.arange(B + 1, dtype=torch.int32, device=dev)
out = torch.empty(B, H * D, device=dev, dtype=torch.float16)
for sets in (2, 8, 32):
    big_k = torch.randn(sets, total, Hkv, 32 * D, device=dev, dtype=torch.float16)
    big_v = torch.randn(sets, total, Hkv, 32 * D, device=dev, dtype=torch.float16)
    for mode in ("random pages", "sequential pages"):
        bt = (torch.randperm(total, device=dev) if mode.startswith("random") else torch.arange(total, device=dev)).int().view(B, pages_per).contiguous()
        t = mb.timeit(lambda i: nat.attn_paged(q, H * D, big_k[i], big_v[i], bt, ctxl, cu, out, B, H, Hkv, D, 1, ctx, D ** -0.5, 1, None), sets)
        print(f"pools={sets} ({sets * 1.07:.1f} GB) {mode}: {t*1e6:.1f} us")
    del big_k, big_v
