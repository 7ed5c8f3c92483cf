"""Prefill attention over the paged cache: time per launch and effective TFLOP/s (B sequences of L tokens, causal)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
import microbench as mb
from tgis_amd import native as nat
dev = mb.dev
for (B, H, Hkv, D, L) in [(32, 32, 32, 128, 1004), (8, 32, 32, 128, 4096), (16, 32, 4, 64, 512), (4, 12, 1, 128, 2048)]:
    pages_per = (L + 31) // 32
    total = B * pages_per
    kpool = torch.randn(total, Hkv, 32 * D, device=dev).half()
    vpool = torch.randn(total, Hkv, 32 * D, device=dev).half()
    bt = torch.randperm(total, device=dev).int().view(B, pages_per).contiguous()
    q = torch.randn(B * L, H * D, device=dev).half()
    ctxl = torch.full((B,), L, dtype=torch.int32, device=dev)
    cu = (torch.arange(B + 1, device=dev) * L).int()
    out = torch.empty(B * L, H * D, device=dev, dtype=torch.float16)
    t = mb.timeit(lambda i: nat.attn_paged(q, H * D, kpool, vpool, bt, ctxl, cu, out, B, H, Hkv, D, L, L, D ** -0.5, 1, None), 1, iters=4)
    flops = 4.0 * B * H * D * L * (L + 1) / 2
    print(f"prefill attn B={B} H={H} Hkv={Hkv} D={D} L={L}: {t*1e3:.3f} ms  {flops/t/1e12:.1f} TFLOP/s")
