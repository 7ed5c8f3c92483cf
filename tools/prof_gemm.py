"""Tiny driver for rocprofv3: runs each cfg3 GPTQ GEMM shape a few times (eager launches)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
from tgis_amd import native as nat
dev = torch.device("cuda:0")
shapes = [(4096, 22016), (4096, 12288), (11008, 4096), (4096, 4096)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in sys.argv[1].split("x"))]
for (K, N) in shapes:
    G = K // 128
    ws_list = []
    for i in range(4):
        qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
        qz = torch.randint(-2**31, 2**31 - 1, (G, N // 8), dtype=torch.int32, device=dev)
        sc = (torch.rand(G, N, device=dev) * 0.002 + 0.001).half()
        ws_list.append(nat.GptqWeight(qw, qz, sc, None, 4, 128))
    x = torch.randn(32, K, device=dev).half()
    out = torch.empty(32, N, device=dev, dtype=torch.float16)
    ws = nat.Workspace(ws_list[0].workspace_bytes(32), dev)
    for it in range(8):
        nat.gptq_gemm(x, ws_list[it % 4], ws, out=out)
    torch.cuda.synchronize()
