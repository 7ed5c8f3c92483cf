"""qkv projection + rotary embedding + cache write at the cfg3 shapes: GEMM (deferred split-K) + tgis_rope_kv_write vs the
fused launch tgis_gptq_gemm_rope_f16, GPU time from a captured graph.  TGIS_GPTQ_PLAN="KR,S,WK,TN" overrides the plan."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from tgis_amd import native as nat  # noqa: E402
from microbench import timeit  # noqa: E402

dev = torch.device("cuda:0")
H = Hkv = 32
D, K, B = 128, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = (H + 2 * Hkv) * D
sets = 6
plain, roped = [], []
for i in range(sets):
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
    qz = torch.randint(-2**31, 2**31 - 1, (K // 128, N // 8), dtype=torch.int32, device=dev)
    sc = (torch.rand(K // 128, N, device=dev) * 0.002 + 0.001).half()
    plain.append(nat.GptqWeight(qw, qz, sc, None, 4, 128))
    roped.append(nat.GptqWeight(qw, qz, sc, None, 4, 128, rope=(D, H + Hkv)))
x = torch.randn(B, K, device=dev).half()
cos = torch.randn(2048, D // 2, device=dev).half()
sin = torch.randn(2048, D // 2, device=dev).half()
pos = torch.randint(0, 2048, (B,), device=dev).int()
pages = 64 * B
slots = (torch.randperm(pages, device=dev)[:B] * 32 + torch.randint(0, 32, (B,), device=dev)).int()
kpool = torch.zeros((pages, Hkv, 32 * D), dtype=torch.float16, device=dev)
vpool = torch.zeros_like(kpool)
ws = nat.Workspace(plain[0].workspace_bytes(B), dev)
t_gemm = timeit(lambda i: nat.gptq_gemm_partial(x, plain[i]), sets)
t_pair = timeit(lambda i: nat.rope_kv_write(nat.gptq_gemm_partial(x, plain[i]), cos, sin, pos, slots, kpool, vpool, H, Hkv, D, D), sets)
t_fused = timeit(lambda i: nat.gptq_gemm_rope(x, roped[i], None, cos, sin, pos, slots, kpool, vpool, H, Hkv, D), sets)
print(f"B={B}: GEMM (partial) {t_gemm*1e6:6.2f} us   GEMM + rope_kv_write {t_pair*1e6:6.2f} us   fused {t_fused*1e6:6.2f} us"
      f"   plan {os.environ.get('TGIS_GPTQ_PLAN', 'default')}")
