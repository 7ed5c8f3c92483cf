"""qkv + rotary + cache write with a fragment-order operand at the cfg3 shape: plain unsplit GEMM vs the fused launch, GPU
time from a captured graph; TGIS_HIP_LIB selects an ablation build (csrc/gptq_wide_body.h, WIDE_ABL)."""
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
D, K, B = 128, 4096, 32
N = (H + 2 * Hkv) * D
sets = 24
plain, roped = [], []
for i in range(sets):
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
    qz = torch.randint(-2**31, 2**31 - 1, (K // 128, N // 8), dtype=torch.int32, device=dev)
    sc = (torch.rand(K // 128, N, device=dev) * 0.002 + 0.001).half()
    plain.append(nat.GptqWeight(qw, qz, sc, None, 4, 128))
    roped.append(nat.GptqWeight(qw, qz, sc, None, 4, 128, rope=(D, H + Hkv)))
x = torch.randn(B, K, device=dev).half()
xf = nat.FragAct.from_rows(x)
cos = torch.randn(2048, D // 2, device=dev).half()
sin = torch.randn(2048, D // 2, device=dev).half()
pos = torch.randint(0, 2048, (B,), device=dev).int()
pages = 64 * B
slots = (torch.randperm(pages, device=dev)[:B] * 32 + torch.randint(0, 32, (B,), device=dev)).int()
kpool = torch.zeros((pages, Hkv, 32 * D), dtype=torch.float16, device=dev)
vpool = torch.zeros_like(kpool)
ws = nat.Workspace(plain[0].workspace_bytes(B), dev)
out = torch.empty((B, N), dtype=torch.float16, device=dev)
t_plain = timeit(lambda i: nat.gptq_gemm(xf, plain[i], ws, out=out), sets)
t_fused = timeit(lambda i: nat.gptq_gemm_rope(xf, roped[i], None, cos, sin, pos, slots, kpool, vpool, H, Hkv, D, out=out), sets)
t_fused_row = timeit(lambda i: nat.gptq_gemm_rope(x, roped[i], None, cos, sin, pos, slots, kpool, vpool, H, Hkv, D, out=out), sets)
print(f"{os.path.basename(os.environ.get('TGIS_HIP_LIB', 'libtgis_hip.so'))}: frag GEMM (plan {os.environ.get('TGIS_GPTQ_WIDE_PLAN', 'default')}) "
      f"{t_plain*1e6:6.2f} us   frag fused rope {t_fused*1e6:6.2f} us   row-major fused rope {t_fused_row*1e6:6.2f} us")
