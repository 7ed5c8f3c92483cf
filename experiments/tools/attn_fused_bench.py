"""Decode attention with the rotary embedding + cache write in its prologue (tgis_attn_decode_rope) against the two
launches it replaces (tgis_rope_kv_write_partial + tgis_attn_paged), GPU time per layer-step from a captured graph.
    python experiments/build.py && python experiments/tools/attn_fused_bench.py B H Hkv D ctx [S]"""
import sys

import torch

sys.path.insert(0, "experiments")
import native_experiments as nat  # noqa: E402  (binds libtgis_experiments.so; must come before anything loads the product library)
sys.path.insert(0, "tools")
import microbench as mb  # noqa: E402

dev = mb.dev
B, H, Hkv, D, ctx = (int(v) for v in sys.argv[1:6]) if len(sys.argv) >= 6 else (32, 32, 32, 128, 1024)
S = int(sys.argv[6]) if len(sys.argv) > 6 else 2
dt = torch.float16
N = (H + 2 * Hkv) * D
pages_per = (ctx + 31) // 32
total = B * pages_per
sets = 4
pools = [(torch.randn(total, Hkv, 32 * D, device=dev).to(dt), torch.randn(total, Hkv, 32 * D, device=dev).to(dt))
         for _ in range(sets)]
bt = torch.randperm(total, device=dev).int().view(B, pages_per).contiguous()
ctxl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
cu = torch.arange(B + 1, dtype=torch.int32, device=dev)
pos = torch.full((B,), ctx - 1, dtype=torch.int32, device=dev)
slots = (bt[:, (ctx - 1) // 32] * 32 + (ctx - 1) % 32).int().contiguous()
cos = torch.rand(ctx + 1, D // 2, device=dev).to(dt)
sin = torch.rand(ctx + 1, D // 2, device=dev).to(dt)
part = nat.Partial(torch.randn(S * 32 * N, device=dev), S, N, B, N, None)
part.dtype = dt
out = torch.empty(B, H * D, device=dev, dtype=dt)
ns = nat.attn_num_splits(B, Hkv, H, 1, ctx)
ws = nat.Workspace(max(4096, nat.attn_workspace_bytes(B, H, Hkv, D, ns)), dev)
qrot = torch.randn(B, N, device=dev).to(dt)


def separate(i):
    q = nat.rope_kv_write(part, cos, sin, pos, slots, pools[i][0], pools[i][1], H, Hkv, D, D)
    nat.attn_paged(q, q.stride(0), pools[i][0], pools[i][1], bt, ctxl, cu, out, B, H, Hkv, D, 1, ctx, D ** -0.5, ns, ws)


def fused(i):
    nat.attn_decode_rope(part, cos, sin, pos, slots, pools[i][0], pools[i][1], bt, ctxl, cu, out, B, H, Hkv, D, D, ctx,
                         D ** -0.5, ns, ws)


def only_attn(i):
    nat.attn_paged(qrot, qrot.stride(0), pools[i][0], pools[i][1], bt, ctxl, cu, out, B, H, Hkv, D, 1, ctx, D ** -0.5, ns,
                   ws)


for name, fn in (("rope launch + attention launch", separate), ("one launch", fused), ("attention launch alone", only_attn)):
    t = mb.timeit(fn, sets)
    print(f"B={B} H={H} Hkv={Hkv} D={D} ctx={ctx} splits={ns}  {name:32s} {t * 1e6:8.1f} us")
