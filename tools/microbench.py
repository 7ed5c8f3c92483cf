"""Per-kernel achieved bandwidth at the cfg3 (Llama-2-7B int4, B=32, ctx 1024) shapes.  GPU only.
Rotates over enough distinct weight / KV sets to defeat the 256 MiB Infinity Cache."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
from tgis_amd import native as nat  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n_sets, iters=20, warm=3):
    """GPU time per call: the calls are captured into one HIP graph (no host launch cost in the number)."""
    for i in range(warm):
        fn(i % n_sets)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            fn(i % n_sets)
    g.replay()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    reps = 5
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (iters * reps) * 1e-3


def bench_gptq(M, K, N, gs=128, sets=8):
    G = K // gs
    ws_list = []
    for i in range(sets):
        qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
        qz = torch.randint(-2**31, 2**31 - 1, (G, N // 8), dtype=torch.int32, device=dev)
        sc = (torch.rand(G, N, device=dev) * 0.002 + 0.001).half()
        ws_list.append(nat.GptqWeight(qw, qz, sc, None, 4, gs))
    x = torch.randn(M, K, device=dev).half()
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    ws = nat.Workspace(ws_list[0].workspace_bytes(M), dev)
    t = timeit(lambda i: nat.gptq_gemm(x, ws_list[i], ws, out=out), sets)
    byts = K * N / 2 + G * N * 2.5
    print(f"gptq_gemm M={M} K={K} N={N}: {t*1e6:8.1f} us  {byts/t/1e9:8.1f} GB/s (algorithmic)")
    return t


def bench_dense(M, K, N, sets=2):
    wl = [nat.DenseWeight((torch.randn(N, K, device=dev) * 0.02).half()) for _ in range(sets)]
    x = torch.randn(M, K, device=dev).half()
    ws = nat.Workspace(wl[0].workspace_bytes(M), dev)
    out = torch.empty(M, N, device=dev, dtype=torch.float32)
    t = timeit(lambda i: nat.dense_gemm(x, wl[i], ws, out_f32=True, out=out), sets)
    print(f"dense_gemm M={M} K={K} N={N}: {t*1e6:8.1f} us  {N*K*2/t/1e9:8.1f} GB/s")
    return t


def bench_attn(B, H, Hkv, D, ctx, sets=2, ns=None):
    pages_per = (ctx + 31) // 32
    total = B * pages_per
    pools = [(torch.randn(total, Hkv, 32 * D, device=dev).half(), torch.randn(total, Hkv, 32 * D, device=dev).half())
             for _ in range(sets)]
    bt = torch.randperm(total, device=dev).int().view(B, pages_per).contiguous()
    q = torch.randn(B, H * D, device=dev).half()
    ctxl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
    cu = torch.arange(B + 1, dtype=torch.int32, device=dev)
    out = torch.empty(B, H * D, device=dev, dtype=torch.float16)
    if ns is None:
        ns = nat.attn_num_splits(B, Hkv, H, 1, ctx)
    ws = nat.Workspace(nat.attn_workspace_bytes(B, H, Hkv, D, ns), dev)
    t = timeit(lambda i: nat.attn_paged(q, H * D, pools[i][0], pools[i][1], bt, ctxl, cu, out, B, H, Hkv, D, 1, ctx,
                                        D ** -0.5, ns, ws), sets)
    byts = B * ctx * 2 * Hkv * D * 2
    print(f"attn_decode B={B} H={H} Hkv={Hkv} D={D} ctx={ctx} splits={ns}: {t*1e6:8.1f} us  {byts/t/1e9:8.1f} GB/s")
    return t


def bench_small(B=32, E=4096, I=11008):
    x = torch.randn(B, E, device=dev).half()
    r = torch.randn(B, E, device=dev).half()
    w = torch.ones(E, device=dev).half()
    y = torch.empty_like(x)
    ro = torch.empty_like(x)
    t = timeit(lambda i: nat.rmsnorm_residual(x, r, w, 1e-5, y=y, res_out=ro), 1, iters=200)
    print(f"rmsnorm B={B} E={E}: {t*1e6:6.2f} us")
    gu = torch.randn(B, 2 * I, device=dev).half()
    o = torch.empty(B, I, device=dev).half()
    t = timeit(lambda i: nat.act_mul(gu, I, out=o), 1, iters=200)
    print(f"silu_mul B={B} I={I}: {t*1e6:6.2f} us")
    lg = torch.randn(B, 32000, device=dev)
    t = timeit(lambda i: nat.argmax_logprob(lg), 1, iters=200)
    print(f"argmax_logprob B={B} V=32000: {t*1e6:6.2f} us")


if __name__ == "__main__":
    print(nat.version(), torch.cuda.get_device_name(0))
    tot = 0.0
    tot += 32 * bench_attn(32, 32, 32, 128, 1024)
    for (K, N) in [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)]:
        tot += 32 * bench_gptq(32, K, N)
    tot += bench_dense(32, 4096, 32000)
    bench_small()
    print(f"sum of big kernels for one cfg3 decode step: {tot*1e3:.3f} ms  -> {32/tot:.0f} tok/s upper bound")
    bench_attn(64, 8, 1, 128, 2048)
    bench_attn(16, 32, 4, 64, 512)
    bench_attn(32, 12, 1, 128, 4096)
