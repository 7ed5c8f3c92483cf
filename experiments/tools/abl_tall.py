"""Tall int4 GEMM ablations at M = 4096 (qkv and down shapes): build variants of the library with -DABL_T_NODEQ /
-DABL_T_NOLDS / -DABL_T_NOSTAGE (wrong results by design) into lib/abl_<variant>.so, then
    TGIS_TALL_MAX_M=100000 TGIS_TALL_TW=2 TGIS_HIP_LIB=$PWD/text-generation-inference_amd/lib/abl_T_NODEQ.so python tools/abl_tall.py
GPU box only.
NOTE (round 5): the -DABL_T_* switches live in experiments/csrc/r04_ablations/gptq.hip.txt (round-4 tree, commit 1892754)."""
import os, sys, torch
sys.path.insert(0, "tools"); sys.path.insert(0, "text-generation-inference_amd")
import microbench as mb
from tgis_amd import native as nat
dev = torch.device("cuda:0")
for (K, N, act) in [(4096, 12288, 0), (11008, 4096, 0)]:
    G = K // 128
    lin = []
    for i in range(3):
        qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
        qz = torch.randint(-2**31, 2**31 - 1, (G, N // 8), dtype=torch.int32, device=dev)
        sc = (torch.rand(G, N, device=dev) * 0.002 + 0.001).half()
        lin.append(nat.GptqWeight(qw, qz, sc, None, 4, 128, gate_up=(act == 2)))
    for M in (4096,):
        x = torch.randn(M, K, device=dev).half()
        ws = nat.Workspace(lin[0].workspace_bytes(M), dev)
        t = mb.timeit(lambda i: nat.gptq_gemm(x, lin[i], ws, act=act), 3, iters=6)
        print(f"{os.getenv('TAG','base'):18s} TW={os.getenv('TGIS_TALL_TW','-')} K={K:5d} N={N:5d} M={M}: {t*1e6:8.1f} us {2*M*K*N/t/1e12:7.1f} TFLOP/s", flush=True)
