"""Randomised dense skinny-GEMM check (tgis_dense_gemm act 0 / 1 / 2, f32 logits, partial + norm consumer) over shapes
that hit every launch plan, against torch on the GPU.  GPU box only.   python tools/fuzz_dense.py [cases] [seed]"""
import sys

import torch

sys.path.insert(0, "text-generation-inference_amd")
from tgis_amd import native as nat  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def rnd(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))


for case in range(cases):
    dtype = (torch.float16, torch.bfloat16)[rnd(0, 1)]
    act = (0, 0, 1, 2)[rnd(0, 3)]
    M = (1, 3, 16, 32, 33, 64, 100, 200)[rnd(0, 7)]
    K = 8 * rnd(4, 1400)
    N = (8 * rnd(1, 1600)) if act != 2 else 32 * rnd(1, 700)
    x = (torch.randn(M, (2 * K if act == 1 else K), generator=g) * 0.5).to(dtype).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dtype).to(dev)
    bias = (torch.randn(N, generator=g) * 0.05).to(dtype).to(dev) if rnd(0, 1) else None
    dw = nat.DenseWeight(w, gate_up=(act == 2))
    ws = nat.Workspace(dw.workspace_bytes(M), dev)
    got = nat.dense_gemm(x, dw, ws, bias=bias, act=act).float()
    xf = x.float()
    if act == 1:
        xf = (torch.nn.functional.silu(xf[:, :K]).to(dtype).float() * xf[:, K:]).to(dtype).float()
    lin = xf @ w.float().t() + (bias.float() if bias is not None else 0)
    if act == 2:
        lin = lin.to(dtype).float()
        I = N // 2
        want = torch.nn.functional.silu(lin[:, :I]).to(dtype).float() * lin[:, I:]
    else:
        want = lin
    eps = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    scale = float(lin.abs().max()) + 1e-3
    tol = eps * scale * (4 if act == 2 else 2) * (scale if act == 2 else 1) + 1e-4
    err = float((got - want).abs().max())
    flag = "" if err <= tol else "   <-- FAIL"
    print(f"case {case:3d} {str(dtype)[6:]:9s} act={act} M={M:3d} K={K:5d} N={N:5d} bias={bias is not None!s:5s} err {err:.2e} tol {tol:.2e}{flag}",
          flush=True)
    if flag:
        sys.exit(1)
print(f"all {cases} cases within tolerance")
