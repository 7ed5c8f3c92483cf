"""Does the row stride of x matter for the GEMM's x staging (L2 channel conflicts at power-of-two strides)?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
import microbench as mb
from tgis_amd import native as nat
dev = mb.dev
os.environ["TGIS_GPTQ_NOREDUCE"] = "1"
for (K, N) in [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)]:
    G = K // 128
    wl = []
    for i in range(8):
        qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
        qz = torch.randint(-2**31, 2**31 - 1, (G, N // 8), dtype=torch.int32, device=dev)
        sc = (torch.rand(G, N, device=dev) * 0.002 + 0.001).half()
        wl.append(nat.GptqWeight(qw, qz, sc, None, 4, 128))
    out = torch.empty(32, N, device=dev, dtype=torch.float16)
    ws = nat.Workspace(wl[0].workspace_bytes(32), dev)
    for pad in (0, 8, 64, 72, 520):
        xb = torch.randn(32, K + pad, device=dev).half()
        x = xb[:, :K]
        t = mb.timeit(lambda i: nat.gptq_gemm(x, wl[i], ws, out=out), 8)
        print(f"K={K} N={N} ldx=K+{pad}: {t*1e6:.1f} us")
