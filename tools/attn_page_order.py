"""Round 5: does the ORDER of a batch's pages in the pool move the decode attention launch?  (bench.py's first timed block — the
first batch of a fresh pool, pages in allocation order — is always 0.1-0.2 ms per step slower than the blocks after it, whose
batches take recycled pages.)  cfg3 shape: B 32, 32 heads, D 128, ctx 1023; block tables: random, sequential (b * P + p),
page-major (p * B + b), reversed sequential.  -> stdout"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "text-generation-inference_amd"))
from tgis_amd import native as nat  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=60, reps=5):
    for i in range(6):
        fn(i)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best * 1e3


def main():
    B, H, Hkv, D, ctx, sets = 32, 32, 32, 128, 1023, 6
    P = (ctx + 31) // 32
    total = B * P
    cap = 2 * total  # pool twice the batch: fragmented orders below
    pools = [(torch.randn(cap, Hkv, 32 * D, device=dev).half(), torch.randn(cap, Hkv, 32 * D, device=dev).half())
             for _ in range(sets)]
    q = torch.randn(B, H * D, device=dev).half()
    ctxl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
    cu = torch.arange(B + 1, dtype=torch.int32, device=dev)
    out = torch.empty(B, H * D, device=dev, dtype=torch.float16)
    ns = nat.attn_num_splits(B, Hkv, H, 1, ctx)
    ws = nat.Workspace(nat.attn_workspace_bytes(B, H, Hkv, D, ns), dev)
    seq = torch.arange(total, device=dev).int().view(B, P)
    orders = {
        "random": torch.randperm(total, device=dev).int().view(B, P),
        "sequential b*P+p": seq,
        "page-major p*B+b": torch.arange(total, device=dev).int().view(P, B).t(),
        "reversed": (total - 1 - seq),
        "random rows, sequential inside": seq[torch.randperm(B, device=dev)],
        "page-major, rows shuffled per p": torch.stack([p * B + torch.randperm(B, device=dev) for p in range(P)], 1).int(),
        "page-major in pairs of pages": ((torch.arange(P, device=dev) // 2) * 2 * B + torch.arange(P, device=dev) % 2)[None, :].int()
                                        + 2 * torch.arange(B, device=dev)[:, None].int(),
        "page-major, every second id": 2 * torch.arange(total, device=dev).int().view(P, B).t(),
        "page-major over a sorted random half": torch.sort(torch.randperm(cap, device=dev)[:total]).values.int().view(P, B).t(),
        "random over the 2x pool": torch.randperm(cap, device=dev)[:total].int().view(B, P),
    }
    print(nat.version(), torch.cuda.get_device_name(0), f"splits {ns}")
    for rnd in range(2):
        print(f' round {rnd}')
        for name, bt in orders.items():
            bt = bt.contiguous()
            t = timeit(lambda i: nat.attn_paged(q, H * D, pools[i % sets][0], pools[i % sets][1], bt, ctxl, cu, out, B, H, Hkv, D, 1,
                                                ctx, D ** -0.5, ns, ws))
            print(f"  {name:34s} {t:7.2f} us   {B * ctx * 2 * Hkv * D * 2 / t / 1e6:6.2f} TB/s")

    # where the V pool sits relative to the K pool (page-major tables): the product's pool is [layer][K | V][page], i.e. V at
    # num_pages * 256 KiB behind K; "interleaved" emulates a [page][K | V] pool (V one page behind K, even page ids only)
    print(" V pool offset (page-major tables)")
    page = Hkv * 32 * D
    bigs = [torch.randn((2 * cap + 64) * page, device=dev).half() for _ in range(sets)]
    pm = torch.arange(total, device=dev).int().view(P, B).t().contiguous()
    cases = {"V = K + cap pages (as the product)": (cap * page, pm),
             "V = K + cap pages + 4 KiB": (cap * page + 2048, pm),
             "V = K + cap pages + 64 KiB": (cap * page + 32768, pm),
             "V = K + cap pages + 128 KiB": (cap * page + 65536, pm),
             "V = K + total pages (dense K, then dense V)": (total * page, pm),
             "interleaved [page][K | V]": (page, (2 * pm).contiguous())}
    for rnd in range(2):
        for name, (voff, bt) in cases.items():
            t = timeit(lambda i: nat.attn_paged(q, H * D, bigs[i % sets], bigs[i % sets][voff:], bt, ctxl, cu, out, B, H, Hkv, D, 1,
                                                ctx, D ** -0.5, ns, ws))
            print(f"  {name:46s} {t:7.2f} us   {B * ctx * 2 * Hkv * D * 2 / t / 1e6:6.2f} TB/s")


def grouped(B, P, g, dev_):
    """page-major inside groups of g sequences (group after group in the pool)"""
    b = torch.arange(B, device=dev_)[:, None]
    p = torch.arange(P, device=dev_)[None, :]
    return ((b // g) * (g * P) + p * g + b % g).int().contiguous()


def more_rounds(B=64, H=32, Hkv=32, D=128, ctx=1023, sets=4):
    """B = 64 at the cfg3 width: 2048 blocks = two rounds of the 1024 that fit; which sequences' pages should be neighbours?"""
    P = (ctx + 31) // 32
    total = B * P
    pools = [(torch.randn(total, Hkv, 32 * D, device=dev).half(), torch.randn(total, Hkv, 32 * D, device=dev).half())
             for _ in range(sets)]
    q = torch.randn(B, H * D, device=dev).half()
    ctxl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
    cu = torch.arange(B + 1, dtype=torch.int32, device=dev)
    out = torch.empty(B, H * D, device=dev, dtype=torch.float16)
    ns = nat.attn_num_splits(B, Hkv, H, 1, ctx)
    ws = nat.Workspace(nat.attn_workspace_bytes(B, H, Hkv, D, ns), dev)
    orders = {"random": torch.randperm(total, device=dev).int().view(B, P).contiguous(),
              "page-major over all B": grouped(B, P, B, dev)}
    for g in (32, 16, 8, 4, 2):
        if g < B:
            orders[f"page-major in groups of {g} sequences"] = grouped(B, P, g, dev)
    print(f" B {B} H {H} Hkv {Hkv} ctx {ctx} splits {ns}")
    for rnd in range(2):
        for name, bt in orders.items():
            t = timeit(lambda i: nat.attn_paged(q, H * D, pools[i % sets][0], pools[i % sets][1], bt, ctxl, cu, out, B, H, Hkv, D, 1,
                                                ctx, D ** -0.5, ns, ws), iters=30)
            print(f"  {name:42s} {t:7.2f} us   {B * ctx * 2 * Hkv * D * 2 / t / 1e6:6.2f} TB/s")


class ClassPool:
    """Round 6, VERDICT r05 item 2a — measured, not kept in the product: a free list with one heap per residue class of the
    page id (C classes); page p of the sequence in lane l is taken from class (l + p) mod C, lowest id first, an empty class
    is replaced by the fullest one.  C = 1 is the product's allocator (utils/kv_cache.py: lowest free id first)."""

    def __init__(self, num_pages, C):
        import heapq

        self.hq, self.C, self.num_pages = heapq, C, num_pages
        self.heaps = [list(range(c, num_pages, C)) for c in range(C)]
        self.free_pages = num_pages

    def alloc_classes(self, wants):
        out = []
        for w in wants:
            h = self.heaps[w % self.C]
            if not h:
                h = max(self.heaps, key=len)
            out.append(self.hq.heappop(h))
        self.free_pages -= len(out)
        return out

    def free(self, pages):
        for p in pages:
            self.hq.heappush(self.heaps[p % self.C], p)
        self.free_pages += len(pages)


def aged_tables(C, B, P, cap, seed=0):
    """Block tables [B, P] a C-class free list hands a batch (dealt page-major, sequence b in lane b) on a pool of `cap` pages
    that has churned: random allocations and frees first, then enough of what is held freed again."""
    import random

    rng = random.Random(seed)
    cache = ClassPool(cap, C)
    held = []
    while cache.free_pages > cap // 4:
        held.append(cache.alloc_classes(range(rng.randrange(1, 48))))
    rng.shuffle(held)
    while cache.free_pages < B * P + cap // 8:
        cache.free(held.pop())
    flat = cache.alloc_classes([b + p for p in range(P) for b in range(B)])
    return torch.tensor(flat, dtype=torch.int32).view(P, B).t().contiguous()


def classes_sweep(B=32, H=32, Hkv=32, D=128, ctx=1023, sets=4):
    """Round 6: how many residue classes does the free list need so that a churned pool reads like a pristine one?"""
    P = (ctx + 31) // 32
    total = B * P
    cap = 4 * total
    pools = [(torch.randn(cap, Hkv, 32 * D, device=dev).half(), torch.randn(cap, Hkv, 32 * D, device=dev).half())
             for _ in range(sets)]
    q = torch.randn(B, H * D, device=dev).half()
    ctxl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
    cu = torch.arange(B + 1, dtype=torch.int32, device=dev)
    out = torch.empty(B, H * D, device=dev, dtype=torch.float16)
    ns = nat.attn_num_splits(B, Hkv, H, 1, ctx)
    ws = nat.Workspace(nat.attn_workspace_bytes(B, H, Hkv, D, ns), dev)
    orders = {"page-major p*B+b (pristine)": torch.arange(total).int().view(P, B).t().contiguous(),
              "random over the 4x pool": torch.randperm(cap)[:total].int().view(B, P).contiguous()}
    for C in (1, 2, 4, 8, 16, 32, 64, 128):
        orders[f"churned pool, {C:3d} classes"] = aged_tables(C, B, P, cap)
    print(f" B {B} H {H} Hkv {Hkv} D {D} ctx {ctx} splits {ns} page {Hkv * 32 * D * 2 // 1024} KiB")
    for rnd in range(2):
        for name, bt in orders.items():
            bt = bt.to(dev)
            t = timeit(lambda i: nat.attn_paged(q, H * D, pools[i % sets][0], pools[i % sets][1], bt, ctxl, cu, out, B, H, Hkv, D, 1,
                                                ctx, D ** -0.5, ns, ws), iters=40)
            print(f"  {name:34s} {t:7.2f} us   {B * ctx * 2 * Hkv * D * 2 / t / 1e6:6.2f} TB/s")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "classes":
        classes_sweep()                                  # cfg3: 256 KiB pages
        classes_sweep(H=4, Hkv=4)                        # a cfg3 rank at TP 8: 32 KiB pages
        classes_sweep(B=64, H=64, Hkv=8, ctx=2047, sets=2)   # cfg4 on one GPU: 64 KiB pages
        classes_sweep(B=16, H=32, Hkv=4, D=64, ctx=511)  # cfg2: 16 KiB pages
        classes_sweep(B=32, H=48, Hkv=1, ctx=4095)       # cfg5: 8 KiB pages
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "rounds":
        more_rounds(B=32)                            # cfg3
        more_rounds(B=64)                            # cfg3 at B = 64: two rounds of blocks
        more_rounds(B=16)
        more_rounds(B=8)
        more_rounds(B=32, H=16, Hkv=16)              # a cfg3 rank at TP 2
        more_rounds(B=32, H=4, Hkv=4)                # ... at TP 8
        more_rounds(B=64, H=64, Hkv=8, ctx=2047, sets=2)   # cfg4 on one GPU
        more_rounds(B=64, H=8, Hkv=1, ctx=2047)      # a cfg4 rank at TP 8
        more_rounds(B=32, H=48, Hkv=1, ctx=4095)     # cfg5
        more_rounds(B=16, H=32, Hkv=4, ctx=511)      # cfg2
        sys.exit(0)
    main()
