"""Randomised int4 GEMM check: streaming kernel (32- and 64-row passes), tall kernel, SiLU*up epilogue, deferred split-K
through the norm consumer — against x @ dequantised W (the library's own dequantisation kernel, which the test suite pins
bit-exactly to the GPTQ formula).  GPU box only.   python tools/fuzz_gptq.py [cases] [seed]"""
import sys

import torch

sys.path.insert(0, "text-generation-inference_amd")
from tgis_amd import native as nat  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
g = torch.Generator(device="cpu").manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def rnd(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))


for case in range(cases):
    gs = (64, 128, 128, 256)[rnd(0, 3)]
    K = gs * rnd(1, 11008 // gs)
    act = (0, 0, 2)[rnd(0, 2)]
    N = 32 * rnd(1, 700)
    M = (1, 5, 32, 33, 64, 100, 256, 300, 700, 1500)[rnd(0, 9)]
    G = K // gs
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, generator=g).to(dev)
    qz = torch.randint(-2**31, 2**31 - 1, (G, N // 8), dtype=torch.int32, generator=g).to(dev)
    sc = (torch.rand(G, N, generator=g) * 0.02 + 0.005).half().to(dev)
    w = nat.GptqWeight(qw, qz, sc, None, 4, gs, gate_up=(act == 2))
    wd = nat.gptq_dequant(w).float()  # [K, N] in checkpoint column order
    x = (torch.randn(M, K, generator=g) * 0.5).half().to(dev)
    bias = (torch.randn(N, generator=g) * 0.05).half().to(dev) if rnd(0, 1) else None
    ws = nat.Workspace(w.workspace_bytes(M), dev)
    got = nat.gptq_gemm(x, w, ws, bias=bias, act=act).float()
    lin = x.float() @ wd + (bias.float() if bias is not None else 0)
    if act == 2:
        lin = lin.half().float()
        I = N // 2
        want = torch.nn.functional.silu(lin[:, :I]).half().float() * lin[:, I:]
    else:
        want = lin
    scale = float(lin.abs().max()) + 1e-3
    tol = 2.0 ** -10 * scale * (4 * scale if act == 2 else 2) + 1e-4
    err = float((got - want).abs().max())
    flag = "" if err <= tol else "   <-- FAIL"
    print(f"case {case:3d} act={act} M={M:4d} K={K:5d} N={N:5d} g={gs:3d} bias={bias is not None!s:5s} err {err:.2e} tol {tol:.2e}{flag}",
          flush=True)
    if flag:
        sys.exit(1)
print(f"all {cases} cases within tolerance")
