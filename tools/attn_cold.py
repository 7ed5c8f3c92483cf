"""Round 6: the product's decode attention launch takes 83.4 us inside a cfg3 step and 80.0 us in a
microbenchmark that launches it back to back on the same tables.  What state of the predecessor costs the difference?
The same launch (cfg3 shape, page-major tables, six pool sets in rotation), timed with the library's own HIP events
(tgis_timing_*, events on the launch stream around the launch), eager:
  hot        attention launches back to back
  qkv        each behind the real qkv + rotary + cache-write launch that feeds it (its q is written by that launch on other XCDs,
             the new token's k / v lines are dirty)
  flush      each behind a 192 MB device copy (the L2s and part of the Infinity Cache hold nothing of the launch's small
             operands: lengths, block table, q)
  layer      each inside the layer's seven launches (norm, qkv, ATTENTION, o, norm, gate_up, down) on six weight sets"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
from tgis_amd import native as nat  # noqa: E402

dev = torch.device("cuda:0")
B, H, Hkv, D, ctx, E, I, SETS = 32, 32, 32, 128, 1023, 4096, 11008, 6
P = (ctx + 31) // 32


def gptq(K, N, **kw):
    G = K // 128
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
    qz = torch.randint(0, 2**31 - 1, (G, N // 8), dtype=torch.int32, device=dev) & 0x55555555
    sc = (torch.rand(G, N, device=dev) * 0.002 + 0.001).half()
    return nat.GptqWeight(qw, qz, sc, None, 4, 128, **kw)


def main():
    total = B * P
    pools = [(torch.randn(total, Hkv, 32 * D, device=dev).half(), torch.randn(total, Hkv, 32 * D, device=dev).half())
             for _ in range(SETS)]
    bt = torch.arange(total, device=dev).int().view(P, B).t().contiguous()  # page-major
    ctxl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
    cu = torch.arange(B + 1, dtype=torch.int32, device=dev)
    pos = torch.full((B,), ctx - 1, dtype=torch.int32, device=dev)
    slots = (bt[:, P - 1] * 32 + (ctx - 1) % 32).int().contiguous()
    cos = torch.randn(2048, D // 2, device=dev).half()
    sin = torch.randn(2048, D // 2, device=dev).half()
    wq = [gptq(E, (H + 2 * Hkv) * D, rope=(D, H + Hkv)) for _ in range(SETS)]
    wo = [gptq(E, E) for _ in range(SETS)]
    wg = [gptq(E, 2 * I, gate_up=True) for _ in range(SETS)]
    wd = [gptq(I, E) for _ in range(SETS)]
    nw = torch.ones(E, device=dev).half()
    x0 = (torch.randn(B, E, device=dev) * 0.1).half()
    ws = nat.Workspace(64 << 20, dev)
    big_a = torch.empty(192 << 20, dtype=torch.uint8, device=dev)
    big_b = torch.empty(192 << 20, dtype=torch.uint8, device=dev)
    scale = D ** -0.5

    def attn(i, qkv, out):
        nat.attn_paged(qkv, qkv.stride(0), pools[i][0], pools[i][1], bt, ctxl, cu, out, B, H, Hkv, D, 1, ctx, scale, 1, None)

    def qkv_launch(i, xf):
        return nat.gptq_gemm_rope(xf, wq[i], None, cos, sin, pos, slots, pools[i][0], pools[i][1], H, Hkv, D)

    xf0, _ = nat.rmsnorm_residual(x0, None, nw, 1e-5, frag=True)
    qkv0 = qkv_launch(0, xf0)
    out_f = nat.FragAct.empty(B, H * D, dev)

    def run(mode, iters=48):
        res, x = x0, x0
        for it in range(iters + 6):
            if it == 6:
                torch.cuda.synchronize()
                nat.timing_reset()
                nat.timing_enable(True)
            i = it % SETS
            if mode == "hot":
                attn(i, qkv0, out_f)
            elif mode == "qkv":
                attn(i, qkv_launch(i, xf0), out_f)
            elif mode == "flush":
                big_b.copy_(big_a)
                attn(i, qkv0, out_f)
            elif mode == "layer":
                xf, res = nat.rmsnorm_residual(x, res if it else None, nw, 1e-5, frag=True)
                q = qkv_launch(i, xf)
                attn(i, q, out_f)
                o = nat.gptq_gemm_partial(out_f, wo[i])
                xf2, res = nat.rmsnorm_residual(o, res, nw, 1e-5, frag=True)
                act = nat.gptq_gemm(xf2, wg[i], ws, act=2, out_frag=True)
                x = nat.gptq_gemm_partial(act, wd[i])
        torch.cuda.synchronize()
        nat.timing_enable(False)
        n, ms = nat.timing_read(nat.OP_ATTN)
        return ms * 1e3 / max(n, 1), n

    print(nat.version(), torch.cuda.get_device_name(0))
    for rnd in range(3):
        for mode in ("hot", "qkv", "flush", "layer"):
            us, n = run(mode)
            print(f"  round {rnd} {mode:6s} attention {us:7.2f} us per launch ({n} launches, HIP events)")


if __name__ == "__main__":
    main()
