"""GPTQ linear at M > 64: the tall fused kernel (default rule, one and two column tiles per wave) vs the 64-row streaming
passes (TGIS_TALL_MIN_M=huge) vs dequantise + library GEMM, cfg3 shapes.  GPU time per call from a captured graph.  GPU box only."""
import os
import subprocess
import sys

CODE = r'''
import os, sys, torch
sys.path.insert(0, "tools"); sys.path.insert(0, "text-generation-inference_amd")
import microbench as mb
from tgis_amd import native as nat
from tgis_amd.utils import layers
dev = torch.device("cuda:0")
mode = sys.argv[1]
for (K, N, act) in [(4096, 12288, 0), (4096, 4096, 0), (4096, 22016, 2), (11008, 4096, 0)]:
    G = K // 128
    lin = []
    for i in range(3):
        qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
        qz = torch.randint(-2**31, 2**31 - 1, (G, N // 8), dtype=torch.int32, device=dev)
        sc = (torch.rand(G, N, device=dev) * 0.002 + 0.001).half()
        l = layers.Ex4bitLinearV2(qw, qz, sc, None, None, 4, 128)
        l.gate_up = act == 2
        l.post_init()
        lin.append(l)
    for M in [int(m) for m in sys.argv[2].split(",")]:
        x = torch.randn(M, K, device=dev).half()
        if mode == "lib":  # (TGIS_TALL_MAX_M=0 in the environment: _large_m never routes back to the tall kernel)
            f = lambda i: (nat.act_mul(lin[i]._large_m(x), N // 2) if act == 2 else lin[i]._large_m(x))
        else:
            ws = nat.Workspace(lin[0].q_handle.workspace_bytes(M), dev)
            f = lambda i: nat.gptq_gemm(x, lin[i].q_handle, ws, act=act)
        t = mb.timeit(f, 3, iters=6)
        print(f"{mode:7s} K={K:5d} N={N:5d} act={act} M={M:5d}: {t*1e6:9.1f} us  {2*M*K*N/t/1e12:7.1f} TFLOP/s", flush=True)
'''
ms = sys.argv[1] if len(sys.argv) > 1 else "96,128,256,512,1024,2048,4096"
for mode, env in (("tall", {"TGIS_TALL_MAX_M": "1000000"}), ("tall/1", {"TGIS_TALL_MAX_M": "1000000", "TGIS_TALL_TW": "1"}),
                  ("tall/2", {"TGIS_TALL_MAX_M": "1000000", "TGIS_TALL_TW": "2"}),
                  ("passes", {"TGIS_TALL_MIN_M": "1000000"}), ("lib", {"TGIS_TALL_MAX_M": "0"})):
    e = dict(os.environ, **env)
    m = ms if mode != "passes" else ",".join(x for x in ms.split(",") if int(x) <= 512)
    r = subprocess.run([sys.executable, "-c", CODE, mode, m], env=e, capture_output=True, text=True)
    print(r.stdout, end="")
    if r.returncode:
        print(r.stderr[-1500:])
