"""MLP of one TP shard at M=32: SiLU in the gate_up epilogue (act=2, whole K per block) + plain down, vs a split gate_up
with its reduce + SiLU applied while down stages its operand (act=1).  Shard widths of Llama-2-7B (tp 1..8) and 70B."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
import microbench as mb
from tgis_amd import native as nat
dev = mb.dev


def weights(K, N, gs, gate_up, sets=6):
    out = []
    for _ in range(sets):
        G = K // gs
        qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
        qz = torch.randint(-2**31, 2**31 - 1, (G, N // 8), dtype=torch.int32, device=dev)
        sc = (torch.rand(G, N, device=dev) * 0.002 + 0.001).half()
        out.append(nat.GptqWeight(qw, qz, sc, None, 4, gs, gate_up=gate_up))
    return out


for E, I, M in ((4096, 11008, 32), (4096, 5504, 32), (4096, 2752, 32), (4096, 1376, 32), (8192, 3584, 64), (8192, 3584, 32)):
    gs_down = 128 if I % 128 == 0 else (64 if I % 64 == 0 else 32)
    x = torch.randn(M, E, device=dev).half()
    ws = nat.Workspace(64 << 20, dev)
    gu_f, gu_p = weights(E, 2 * I, 128, True), weights(E, 2 * I, 128, False)
    dn = weights(I, E, gs_down, False)
    mid = torch.empty(M, I, device=dev, dtype=torch.float16)
    wide = torch.empty(M, 2 * I, device=dev, dtype=torch.float16)
    out = torch.empty(M, E, device=dev, dtype=torch.float16)

    def fused(i):
        nat.gptq_gemm(x, gu_f[i], ws, act=2, out=mid)
        nat.gptq_gemm(mid, dn[i], ws, out=out)

    def split(i):
        nat.gptq_gemm(x, gu_p[i], ws, out=wide)
        nat.gptq_gemm(wide, dn[i], ws, act=1, out=out)

    tf, tsp = mb.timeit(fused, 6), mb.timeit(split, 6)
    print(f"E={E} I={I} M={M}: fused epilogue {tf*1e6:6.1f} us   split + act-on-load {tsp*1e6:6.1f} us", flush=True)
