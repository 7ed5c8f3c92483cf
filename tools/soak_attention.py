"""Soak of the in-launch merge of decode key splits: thousands of launches on the same workspace, three alternating q
inputs, every result compared bit for bit with the first one for that q (a stale read of another block's record would
show as a difference).  GPU box only.   python tools/soak_attention.py"""
import sys, torch
sys.path.insert(0, "text-generation-inference_amd")
from tgis_amd import native as nat
dev = torch.device("cuda:0")
def case(dtype, B, H, Hkv, D, ctx, iters):
    g = torch.Generator().manual_seed(B + H + ctx)
    lens = [ctx - 7 * i for i in range(B)]
    pages_per = [(l + 31) // 32 for l in lens]
    total_pages = sum(pages_per)
    bt = torch.zeros((B, max(pages_per)), dtype=torch.int32)
    perm = torch.randperm(total_pages, generator=g)
    o = 0
    for b in range(B):
        bt[b, :pages_per[b]] = perm[o:o + pages_per[b]].int(); o += pages_per[b]
    T = sum(lens)
    dummy = torch.zeros((T, (H + 2 * Hkv) * D), dtype=dtype)
    dummy[:, H * D:] = torch.randn(T, 2 * Hkv * D, generator=g).to(dtype)
    slots = torch.cat([bt[b, torch.arange(l) // 32].long() * 32 + torch.arange(l) % 32 for b, l in enumerate(lens)]).int()
    kpool = torch.zeros((total_pages, Hkv, 32 * D), dtype=dtype, device=dev); vpool = torch.zeros_like(kpool)
    nat.rope_kv_write(dummy.to(dev), None, None, None, slots.to(dev), kpool, vpool, H, Hkv, D, D)
    qs = [torch.randn(B, H * D, generator=g).to(dtype).to(dev) for _ in range(3)]
    ns = max(2, nat.attn_num_splits(B, Hkv, H, 1, max(lens)))
    ws = nat.Workspace(nat.attn_workspace_bytes(B, H, Hkv, D, ns), dev)
    btd, ctxd = bt.to(dev), torch.tensor(lens, dtype=torch.int32).to(dev)
    cuq = torch.arange(B + 1, dtype=torch.int32, device=dev)
    def run(q):
        out = torch.empty((B, H * D), dtype=dtype, device=dev)
        nat.attn_paged(q, H * D, kpool, vpool, btd, ctxd, cuq, out, B, H, Hkv, D, 1, max(lens), D ** -0.5, ns, ws)
        return out
    first = [run(q) for q in qs]
    bad = 0
    for i in range(iters):
        if not torch.equal(run(qs[i % 3]), first[i % 3]): bad += 1
    print(f"{dtype} B={B} H={H} Hkv={Hkv} D={D} ctx={ctx} splits={ns}: {iters} launches, {bad} differing")
    return bad
bad = 0
bad += case(torch.bfloat16, 16, 32, 4, 64, 2048, 4000)
bad += case(torch.float16, 4, 32, 32, 128, 700, 4000)
bad += case(torch.float16, 2, 48, 1, 128, 1500, 4000)
bad += case(torch.float16, 32, 48, 1, 128, 4096, 1500)
bad += case(torch.bfloat16, 1, 32, 8, 128, 4000, 4000)
sys.exit(1 if bad else 0)
