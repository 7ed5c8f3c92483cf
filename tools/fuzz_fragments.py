"""Randomised check of the fragment-order int4 GEMM (gptq_wide_body.h): M 1..64, K and N over the whole range of the plans
(column tiles per wave 2 - 4, k splits, ragged last tiles, one- and two-row-block forms), act 0 / 2, row-major, fragment-order
and deferred (fp32 slab) outputs — against x @ dequantised W (the library's dequantisation kernel, pinned bit-exactly to the
GPTQ formula by the test suite).  GPU box only.   python tools/fuzz_fragments.py [cases] [seed]"""
import sys

import torch

sys.path.insert(0, "text-generation-inference_amd")
from tgis_amd import native as nat  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 80
g = torch.Generator(device="cpu").manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def rnd(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))


served = 0
for case in range(cases):
    gs = (64, 128, 128, 256)[rnd(0, 3)]
    K = gs * rnd(1, 28672 // gs if rnd(0, 5) == 0 else 11008 // gs)
    act = (0, 0, 2)[rnd(0, 2)]
    N = 64 * rnd(1, 900) if act == 2 else 32 * rnd(1, 700)
    if act == 2 and (N // 2) % 64:
        N = (N // 128 + 1) * 128
    M = (1, 3, 16, 31, 32, 33, 40, 63, 64)[rnd(0, 8)]
    G = K // gs
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, generator=g).to(dev)
    qz = torch.randint(-2**31, 2**31 - 1, (G, N // 8), dtype=torch.int32, generator=g).to(dev)
    sc = (torch.rand(G, N, generator=g) * 0.02 + 0.005).half().to(dev)
    w = nat.GptqWeight(qw, qz, sc, None, 4, gs, gate_up=(act == 2))
    if not nat.gptq_fragments_ok(M, w, act):
        print(f"case {case:3d} act={act} M={M:2d} K={K:5d} N={N:5d} g={gs:3d}: not served in fragment order (plan / shape)")
        continue
    served += 1
    wd = nat.gptq_dequant(w).float()
    x = (torch.randn(M, K, generator=g) * 0.5).half().to(dev)
    xf = nat.FragAct.from_rows(x)
    bias = (torch.randn(N, generator=g) * 0.05).half().to(dev) if rnd(0, 1) else None
    ws = nat.Workspace(w.workspace_bytes(M), dev)
    mode = rnd(0, 2)  # 0 row-major out, 1 fragment-order out (act 2), 2 deferred slabs through the consumer (act 0)
    lin = x.float() @ wd + (bias.float() if bias is not None else 0)
    if act == 2:
        lin = lin.half().float()
        I = N // 2
        want = torch.nn.functional.silu(lin[:, :I]).half().float() * lin[:, I:]
        if mode == 1:
            got = nat.gptq_gemm(xf, w, ws, bias=bias, act=2, out_frag=True).to_rows().float()
        else:
            got = nat.gptq_gemm(xf, w, ws, bias=bias, act=2).float()
    else:
        want = lin
        if mode == 2 and bias is None and N <= 16384:
            part = nat.gptq_gemm_partial(xf, w)
            ones = torch.ones(N, dtype=torch.float16, device=dev)
            # the consumer (add + RMSNorm with a unit weight) finishes the split-K sum; undo the normalisation on the host
            res = torch.zeros(M, N, dtype=torch.float16, device=dev)
            y, summed = nat.rmsnorm_residual(part, res, ones, 1e-6)
            got = summed.float()
        else:
            got = nat.gptq_gemm(xf, w, ws, bias=bias, act=0).float()
    scale = float(lin.abs().max()) + 1e-3
    tol = 2.0 ** -10 * scale * (4 * scale if act == 2 else 2) + 1e-4
    err = float((got - want).abs().max())
    flag = "" if err <= tol else "   <-- FAIL"
    print(f"case {case:3d} act={act} M={M:2d} K={K:5d} N={N:5d} g={gs:3d} bias={bias is not None!s:5s} mode={mode} err {err:.2e} tol {tol:.2e}{flag}",
          flush=True)
    if flag:
        sys.exit(1)
print(f"all {served} served cases (of {cases}) within tolerance")
