"""Randomised decode-attention check: shapes drawn over batch, GQA ratio, head size, context and dtype (so that every
launch rule is hit: 2/4/8-wave blocks, key splits merged in the launch, multi-chunk MQA blocks, partial last pages)
against an fp32 torch reference on the GPU.  GPU box only.   python tools/fuzz_attention.py [cases] [seed]"""
import sys

import torch

sys.path.insert(0, "text-generation-inference_amd")
from tgis_amd import native as nat  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def rnd(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))


worst = 0.0
for case in range(cases):
    dtype = (torch.float16, torch.bfloat16)[rnd(0, 1)]
    D = (64, 128)[rnd(0, 1)]
    Hkv = (1, 1, 2, 4, 8, 32)[rnd(0, 5)]
    G = (1, 2, 4, 8, 12, 16, 48)[rnd(0, 6)]
    if Hkv * G > 64:
        G = max(1, 64 // Hkv)
    H = Hkv * G
    B = rnd(1, 40)
    cmax = (40, 300, 1200, 3000)[rnd(0, 3)]
    lens = [rnd(1, cmax) for _ in range(B)]
    pages_per = [(l + 31) // 32 for l in lens]
    total_pages = sum(pages_per)
    bt = torch.zeros((B, max(pages_per)), dtype=torch.int32)
    perm = torch.randperm(total_pages, generator=g)
    o = 0
    for b in range(B):
        bt[b, :pages_per[b]] = perm[o:o + pages_per[b]].int()
        o += pages_per[b]
    T = sum(lens)
    kv = torch.randn(T, 2 * Hkv * D, generator=g).to(dtype)
    dummy = torch.zeros((T, (H + 2 * Hkv) * D), dtype=dtype)
    dummy[:, H * D:] = kv
    slots = torch.cat([bt[b, torch.arange(l) // 32].long() * 32 + torch.arange(l) % 32 for b, l in enumerate(lens)]).int()
    kpool = torch.zeros((total_pages, Hkv, 32 * D), dtype=dtype, device=dev)
    vpool = torch.zeros_like(kpool)
    nat.rope_kv_write(dummy.to(dev), None, None, None, slots.to(dev), kpool, vpool, H, Hkv, D, D)
    q = torch.randn(B, H * D, generator=g).to(dtype).to(dev)
    ns = nat.attn_num_splits(B, Hkv, H, 1, max(lens))
    ws = nat.Workspace(max(4096, nat.attn_workspace_bytes(B, H, Hkv, D, ns)), dev)
    out = torch.empty((B, H * D), dtype=dtype, device=dev)
    nat.attn_paged(q, H * D, kpool, vpool, bt.to(dev), torch.tensor(lens, dtype=torch.int32, device=dev),
                   torch.arange(B + 1, dtype=torch.int32, device=dev), out, B, H, Hkv, D, 1, max(lens), D ** -0.5, ns, ws)
    # reference
    kvd = kv.to(dev).float()
    err = 0.0
    off = 0
    for b, l in enumerate(lens):
        k = kvd[off:off + l, :Hkv * D].view(l, Hkv, D).repeat_interleave(G, dim=1)   # [l, H, D]
        v = kvd[off:off + l, Hkv * D:].view(l, Hkv, D).repeat_interleave(G, dim=1)
        qq = q[b].float().view(H, D)
        s = torch.einsum("hd,lhd->hl", qq, k) * D ** -0.5
        p = torch.softmax(s, dim=-1)
        ref = torch.einsum("hl,lhd->hd", p, v)
        err = max(err, float((out[b].float().view(H, D) - ref).abs().max()))
        off += l
    tol = 3e-3 if dtype == torch.float16 else 2.5e-2
    worst = max(worst, err / tol)
    flag = "" if err <= tol else "   <-- FAIL"
    print(f"case {case:3d} {str(dtype)[6:]:9s} B={B:2d} H={H:2d} Hkv={Hkv:2d} D={D:3d} ctx<= {max(lens):4d} splits={ns:2d}  max err {err:.2e}{flag}",
          flush=True)
    if flag:
        sys.exit(1)
print(f"all {cases} cases within tolerance (worst {worst:.2f} of it)")
