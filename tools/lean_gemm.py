"""Lean int4 GEMM (tgis_gptq_gemm_f16_lean) vs the round-1/2 streaming kernel at the cfg3 shapes: results against an
fp32 torch reference of the exact (q - z) * s arithmetic, and GPU time per launch from a captured graph (weight sets
rotated to defeat the Infinity Cache).  GPU only.   python tools/lean_gemm.py [M] [check|time|both]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from tgis_amd import native as nat  # noqa: E402
from microbench import timeit  # noqa: E402

dev = torch.device("cuda:0")
_main = __name__ == "__main__"
M = int(sys.argv[1]) if _main and len(sys.argv) > 1 else 32
MODE = sys.argv[2] if _main and len(sys.argv) > 2 else "both"
SHAPES = [("qkv", 4096, 12288, 0), ("o", 4096, 4096, 0), ("gate_up", 4096, 22016, 2), ("down", 11008, 4096, 0)]
if os.environ.get("LEAN_SHAPES") == "70b":  # Llama-2-70B (whole matrices, one GPU)
    SHAPES = [("qkv", 8192, 10240, 0), ("o", 8192, 8192, 0), ("gate_up", 8192, 57344, 2), ("down", 28672, 8192, 0)]


def make(K, N, gs, seed, gate_up=False):
    g = torch.Generator(device="cpu").manual_seed(seed)
    G = K // gs
    q = torch.randint(0, 16, (K, N), generator=g, dtype=torch.int32)
    z = torch.randint(0, 14, (G, N), generator=g, dtype=torch.int32)
    sc = (torch.rand(G, N, generator=g) * 0.002 + 0.001).half()
    qw = torch.zeros(K // 8, N, dtype=torch.int32)
    for j in range(8):
        qw |= q[j::8] << (4 * j)
    qz = torch.zeros(G, N // 8, dtype=torch.int32)
    for j in range(8):
        qz |= z[:, j::8] << (4 * j)
    w = (q.float() - (z.float() + 1).repeat_interleave(gs, 0)) * sc.float().repeat_interleave(gs, 0)
    h = nat.GptqWeight(qw.to(dev), qz.to(dev), sc.to(dev), None, 4, gs, gate_up=gate_up)
    return h, w


def check():
    torch.manual_seed(0)
    for name, K, N, act in SHAPES:
        Kc, Nc = (K, N) if name != "gate_up" else (K, 2048)  # keep the CPU reference cheap
        if name == "qkv":
            Nc = 1536
        h, w = make(Kc, Nc, 128, 1, gate_up=act == 2)
        for m in sorted({M, 1, 7}):
            x = (torch.randn(m, Kc) * (1.0 if name != "down" else 3.0)).half()
            x[:, 5] *= 30.0  # an outlier channel
            xd = x.to(dev)
            xs = nat.xsum(xd)
            ws = nat.Workspace(h.workspace_bytes(m), dev)
            ref = x.double() @ w.double()
            if act == 2:
                I = Nc // 2
                gte = ref[:, :I].half().float()
                ref_o = (torch.nn.functional.silu(gte).half().float() * ref[:, I:].half().float())
            else:
                ref_o = ref.float()
            old = nat.gptq_gemm(xd, h, ws, act=act).float().cpu()
            new_t = nat.gptq_gemm_lean(xd, xs, h, ws, act=act, want_xs=act == 2)
            new = new_t.float().cpu()
            scale = ref_o.abs().max().item()
            e_old = (old - ref_o).abs().max().item() / scale
            e_new = (new - ref_o).abs().max().item() / scale
            line = f"{name:8s} M={m:2d} K={Kc} N={Nc}: max|err|/max|ref| old {e_old:.2e} lean {e_new:.2e}"
            if act == 2:
                xs_o = nat.xs_of(new_t).cpu()
                xs_ref = nat.xsum(new_t).cpu()
                line += f"  xs_out max diff {(xs_o - xs_ref).abs().max().item():.2e} (|xs| {xs_ref.abs().max().item():.2e})"
            else:
                p = nat.gptq_gemm_partial_lean(xd, xs, h)
                sl = p.slabs[: p.S * 32 * p.ld].view(p.S, 32, p.ld).sum(0)[:m, :Nc].cpu()
                line += f"  partial(S={p.S}) {(sl - ref_o).abs().max().item() / scale:.2e}"
            st = nat.gptq_lean_status(reset=True)
            print(line + (f"  STATUS {st}" if st else ""), flush=True)
            assert st == 0 and e_new < 2e-3, "lean GEMM disagrees with the fp32 reference"


def bench():
    for name, K, N, act in SHAPES:
        sets = 6 if K * N < (1 << 27) else 3
        hs = []
        for i in range(sets):
            G = K // 128
            qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
            qz = torch.randint(-2**31, 2**31 - 1, (G, N // 8), dtype=torch.int32, device=dev)
            sc = (torch.rand(G, N, device=dev) * 0.002 + 0.001).half()
            hs.append(nat.GptqWeight(qw, qz, sc, None, 4, 128, gate_up=act == 2))
        x = torch.randn(M, K, device=dev).half()
        xs = nat.xsum(x)
        ws = nat.Workspace(hs[0].workspace_bytes(M), dev)
        out = torch.empty(M, N // 2 if act == 2 else N, device=dev, dtype=torch.float16)
        byts = K * N / 2 + (K // 128) * N * 4
        if act == 2:
            t_old = timeit(lambda i: nat.gptq_gemm(x, hs[i], ws, act=2, out=out), sets)
            t_new = timeit(lambda i: nat.gptq_gemm_lean(x, xs, hs[i], ws, act=2, out=out, want_xs=True), sets)
        else:
            t_old = timeit(lambda i: nat.gptq_gemm_partial(x, hs[i]), sets)
            t_new = timeit(lambda i: nat.gptq_gemm_partial_lean(x, xs, hs[i]), sets)
        st = nat.gptq_lean_status(reset=True)
        assert st == 0, f"in-kernel wait gave up: code {st}"
        print(f"{name:8s} M={M} {K}x{N}: old {t_old*1e6:6.2f} us ({byts/t_old/1e12:.2f} TB/s)   lean {t_new*1e6:6.2f} us "
              f"({byts/t_new/1e12:.2f} TB/s)", flush=True)


if _main and MODE in ("check", "both"):
    check()
if _main and MODE in ("time", "both"):
    bench()
