"""-m gpu: every HIP operator, called through the C ABI (tgis_amd.native -> libtgis_hip.so), against the
CPU oracle (oracle/ops_ref.py) on the same seeded inputs.  Tolerances are stated per test: outputs are
f16/bf16 roundings of fp32-accumulated results, so the bound is a few ulps of the output dtype relative to
the magnitude of the result (integer outputs — token ids, slots — are compared bit-exact)."""
import math

import numpy as np
import pytest
import torch

from oracle import ops_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat(gpu_device):
    from tgis_amd import native

    native.load_library()
    return native


def _close(got, want, rtol, atol, what=""):
    got = got.float().cpu()
    want = want.float().cpu()
    err = (got - want).abs()
    bound = atol + rtol * want.abs()
    bad = err > bound
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} out of tolerance, max err {err.max():.4g} " \
                          f"(want max {want.abs().max():.4g})"


# ---- GPTQ -----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,K,N,gs,act_order", [
    (32, 4096, 4096, 128, False),      # o_proj, cfg3
    (32, 4096, 12288, 128, False),     # fused qkv
    (32, 11008, 4096, 128, False),     # down_proj (K not a power of two)
    (1, 256, 64, 64, False),           # minimum sizes, single row
    (7, 512, 96, 32, False),           # group size 32: scale-in-weight path
    (33, 1024, 160, 128, False),       # two row slabs, ragged last slab
    (16, 1024, 256, 1024, False),      # per-channel (one group)
    (5, 1024, 128, 128, True),         # act-order g_idx
    (70, 2048, 2752, 128, False),      # TP-shard width (22016/8), M > 64
    (40, 1024, 128, 128, True),        # act-order g_idx with a 64-row pass
    (64, 4096, 4096, 128, False),      # one full 64-row pass
    (100, 512, 96, 32, False),         # two 64-row passes, ragged, scale-in-weight path
    (5, 224, 64, 32, False),           # K % 64 == 32 (a regrouped row-parallel shard): half-valid last step
    (32, 1376, 4096, 32, False),       # llama-7B down_proj shard at tp=8 after regrouping 128 -> 32
    (32, 2752, 4096, 64, False),       # ... at tp=4 (groups of 64)
])
def test_gptq_gemm(nat, gpu_device, M, K, N, gs, act_order):
    qw, qz, sc, gi = ops_ref.make_gptq_tensors(K, N, gs, seed=K + N + M, act_order=act_order)
    g = torch.Generator().manual_seed(M * 7 + 1)
    x = (torch.randn(M, K, generator=g) * 0.5).half()
    bias = (torch.randn(N, generator=g) * 0.1).half()
    want = ops_ref.gptq_linear(x, qw, qz, sc, gi, gs, bias)
    w = nat.GptqWeight(torch.from_numpy(qw).to(gpu_device), torch.from_numpy(qz).to(gpu_device),
                       torch.from_numpy(sc).to(gpu_device), torch.from_numpy(gi), 4, gs)
    ws = nat.Workspace(w.workspace_bytes(M), gpu_device)
    got = nat.gptq_gemm(x.to(gpu_device), w, ws, bias=bias.to(gpu_device))
    # f16 output of an fp32-accumulated sum of K terms: 2 ulp(f16) of the result + accumulated-order noise
    _close(got, want, rtol=2e-3, atol=2e-3 * float(want.abs().mean()) + 1e-4, what="gptq_gemm")
    # second call reuses workspace (arrival counters must have been left at zero)
    got2 = nat.gptq_gemm(x.to(gpu_device), w, ws, bias=bias.to(gpu_device))
    assert torch.equal(got, got2), "gptq_gemm is not deterministic / counters not reset"
    # the full-dequant path must reproduce the formula bit-for-bit up to the f16 rounding of W
    wd = nat.gptq_dequant(w).float().cpu()
    wref = ops_ref.gptq_dequant(qw, qz, sc, gi, gs)
    if act_order:
        wref = wref[w.perm.cpu().long()]
    assert torch.equal(wd, wref.half().float()), "gptq_dequant differs from the reference formula"


def test_gptq_gemm_fused_silu(nat, gpu_device):
    M, K, N, gs = 32, 1024, 512, 128
    qw, qz, sc, gi = ops_ref.make_gptq_tensors(K, N, gs, seed=5)
    g = torch.Generator().manual_seed(3)
    gu = torch.randn(M, 2 * K, generator=g).half()
    act = (torch.nn.functional.silu(gu[:, :K].float()).half().float() * gu[:, K:].float()).half()
    want = ops_ref.gptq_linear(act, qw, qz, sc, gi, gs)
    w = nat.GptqWeight(torch.from_numpy(qw).to(gpu_device), torch.from_numpy(qz).to(gpu_device),
                       torch.from_numpy(sc).to(gpu_device), None, 4, gs)
    ws = nat.Workspace(w.workspace_bytes(M), gpu_device)
    got = nat.gptq_gemm(gu.to(gpu_device), w, ws, act=1)
    _close(got, want, rtol=3e-3, atol=3e-3 * float(want.abs().mean()) + 1e-4, what="gptq_gemm+silu")


# ---- dense skinny GEMM -------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,K,N,f32out", [(32, 4096, 32000, True), (16, 2048, 2560, False), (3, 200, 72, False),
                                          (64, 2048, 5632, False), (100, 1024, 160, True),
                                         (40, 512, 100, True),
                                         # batches of up to 16 rows stage 16 rows of x (round 6, R16) — every plan family
                                         (16, 5632, 2048, False), (9, 2048, 32000, True), (1, 2048, 2048, False),
                                         (13, 6144, 6400, False), (16, 24576, 6144, False), (17, 2048, 2048, False)])
def test_dense_gemm(nat, gpu_device, dtype, M, K, N, f32out):
    g = torch.Generator().manual_seed(M + K + N)
    x = (torch.randn(M, K, generator=g) * 0.5).to(dtype)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dtype)
    want = x.float() @ w.float().t()
    if K % 8:
        pytest.skip("x rows must be 16-byte aligned")
    dw = nat.DenseWeight(w.to(gpu_device))
    ws = nat.Workspace(dw.workspace_bytes(M), gpu_device)
    got = nat.dense_gemm(x.to(gpu_device), dw, ws, out_f32=f32out)
    eps = 1e-5 if f32out else (1e-3 if dtype == torch.float16 else 8e-3)
    _close(got, want, rtol=2 * eps, atol=2 * eps * float(want.abs().mean()) + 1e-5, what="dense_gemm")


# ---- norms ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows,hidden", [(32, 4096), (1, 64), (5, 2048), (3, 8192)])
@pytest.mark.parametrize("with_res", [False, True])
def test_rmsnorm(nat, gpu_device, dtype, rows, hidden, with_res):
    g = torch.Generator().manual_seed(rows + hidden)
    x = torch.randn(rows, hidden, generator=g).to(dtype)
    r = torch.randn(rows, hidden, generator=g).to(dtype) if with_res else None
    w = (1 + 0.1 * torch.randn(hidden, generator=g)).to(dtype)
    wy, wres = ops_ref.rmsnorm_residual(x, r, w, 1e-5)
    y, res = nat.rmsnorm_residual(x.to(gpu_device), None if r is None else r.to(gpu_device), w.to(gpu_device), 1e-5)
    ulp = 1e-3 if dtype == torch.float16 else 8e-3
    _close(y, wy, rtol=ulp, atol=ulp, what="rmsnorm y")
    assert torch.equal(res.cpu(), wres.to(dtype)), "residual stream must be the rounded fp32 sum, bit-exact"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_layernorm(nat, gpu_device, dtype):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(9, 6144, generator=g).to(dtype)
    r = torch.randn(9, 6144, generator=g).to(dtype)
    w = (1 + 0.1 * torch.randn(6144, generator=g)).to(dtype)
    b = (0.1 * torch.randn(6144, generator=g)).to(dtype)
    wy, wres = ops_ref.layernorm_residual(x, r, w, b, 1e-5)
    y, res = nat.layernorm_residual(x.to(gpu_device), r.to(gpu_device), w.to(gpu_device), b.to(gpu_device), 1e-5)
    ulp = 1e-3 if dtype == torch.float16 else 8e-3
    _close(y, wy, rtol=ulp, atol=ulp, what="layernorm y")
    assert torch.equal(res.cpu(), wres.to(dtype))


# ---- RoPE + KV write + paged attention -------------------------------------------------------------------------
def _paged_setup(gpu_device, dtype, lens_ctx, lens_q, H, Hkv, D, seed, num_pages_extra=3):
    """Build q/k/v for sequences with ctx tokens of which the last q_len are 'new', scatter the whole
    context into a shuffled page pool through tgis_rope_kv_write, return everything needed."""
    g = torch.Generator().manual_seed(seed)
    B = len(lens_ctx)
    pages_per = [(c + 31) // 32 for c in lens_ctx]
    total_pages = sum(pages_per) + num_pages_extra
    perm = torch.randperm(total_pages, generator=g).tolist()
    max_pages = max(pages_per)
    bt = torch.zeros((B, max_pages), dtype=torch.int32)
    pi = 0
    for b in range(B):
        for j in range(pages_per[b]):
            bt[b, j] = perm[pi]
            pi += 1
    return B, bt, total_pages, max_pages, g


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("H,Hkv,D", [(32, 32, 128), (32, 4, 64), (8, 1, 128), (12, 1, 128), (48, 1, 128), (6, 2, 64),
                                      (20, 1, 128), (80, 1, 64), (64, 2, 128)])  # groups of 2, 5 and 2 16-head chunks
def test_rope_kv_attention_prefill_then_decode(nat, gpu_device, dtype, H, Hkv, D):
    """Prefill (ragged lengths incl. 1 and non-multiples of 32) then one decode step, against the oracle's
    rotary + varlen attention.  Also checks the cache contents bit-exactly against the rotated k / v."""
    lens = [45, 1, 32, 97, 64]
    B, bt, total_pages, max_pages, g = _paged_setup(gpu_device, dtype, [l + 1 for l in lens], lens, H, Hkv, D, seed=H * D)
    T = sum(lens)
    W = (H + 2 * Hkv) * D
    qkv = (torch.randn(T, W, generator=g) * 0.7).to(dtype)
    cos, sin = ops_ref.rope_tables(D, 10000.0, 256, dtype)
    pos = torch.cat([torch.arange(l) for l in lens]).int()
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    slots = torch.cat([bt[b, torch.arange(l) // 32].long() * 32 + torch.arange(l) % 32 for b, l in enumerate(lens)]).int()
    # oracle
    q3 = qkv[:, :H * D].view(T, H, D)
    k3 = qkv[:, H * D:(H + Hkv) * D].view(T, Hkv, D)
    v3 = qkv[:, (H + Hkv) * D:].view(T, Hkv, D)
    qr = ops_ref.apply_rope(q3, cos[pos.long()], sin[pos.long()]).to(dtype)
    kr = ops_ref.apply_rope(k3, cos[pos.long()], sin[pos.long()]).to(dtype)
    want = ops_ref.attention_varlen(qr, kr, v3, cu, cu, D ** -0.5)
    # device
    kpool = torch.zeros((total_pages, Hkv, 32 * D), dtype=dtype, device=gpu_device)
    vpool = torch.zeros_like(kpool)
    dq = qkv.to(gpu_device)
    nat.rope_kv_write(dq, cos.to(gpu_device), sin.to(gpu_device), pos.to(gpu_device), slots.to(gpu_device), kpool,
                      vpool, H, Hkv, D, D)
    # rotation parity: one rounding of an fp32 result
    ulp = 1e-3 if dtype == torch.float16 else 8e-3
    _close(dq[:, :H * D].view(T, H, D), qr, rtol=ulp, atol=ulp, what="rope q")
    # cache contents == rotated k (as produced on the device) and v, bit-exact, at the right slots
    kdev = dq[:, H * D:(H + Hkv) * D].view(T, Hkv, D).cpu()
    for b, l in enumerate(lens):
        for j in range((l + 31) // 32):
            K, V = ops_ref.kv_page_unpack(kpool.cpu(), vpool.cpu(), int(bt[b, j]), Hkv, D)
            n = min(32, l - j * 32)
            t0 = int(cu[b]) + j * 32
            assert torch.equal(K[:n], kdev[t0:t0 + n]), "K page content"
            assert torch.equal(V[:n], v3[t0:t0 + n]), "V page content"
    out = torch.empty((T, H * D), dtype=dtype, device=gpu_device)
    ctx = torch.tensor(lens, dtype=torch.int32)
    nat.attn_paged(dq, dq.stride(0), kpool, vpool, bt.to(gpu_device), ctx.to(gpu_device), cu.to(gpu_device), out, B, H,
                   Hkv, D, max(lens), max(lens), D ** -0.5, 1, None)
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2
    _close(out.view(T, H, D), want, rtol=tol, atol=tol, what="prefill attention")

    # ---- one decode step: a new token per sequence at position len ----------------------------------------
    qkv1 = (torch.randn(B, W, generator=g) * 0.7).to(dtype)
    pos1 = torch.tensor(lens, dtype=torch.int32)
    slots1 = torch.empty(B, dtype=torch.int32, device=gpu_device)
    ctx1 = torch.empty(B, dtype=torch.int32, device=gpu_device)
    nat.decode_slots(pos1.to(gpu_device), bt.to(gpu_device), slots1, ctx1)
    want_slots = torch.tensor([int(bt[b, l // 32]) * 32 + l % 32 for b, l in enumerate(lens)], dtype=torch.int32)
    assert torch.equal(slots1.cpu(), want_slots) and torch.equal(ctx1.cpu(), pos1 + 1)
    dq1 = qkv1.to(gpu_device)
    nat.rope_kv_write(dq1, cos.to(gpu_device), sin.to(gpu_device), pos1.to(gpu_device), slots1, kpool, vpool, H, Hkv,
                      D, D)
    q1 = ops_ref.apply_rope(qkv1[:, :H * D].view(B, H, D), cos[pos1.long()], sin[pos1.long()]).to(dtype)
    k1 = ops_ref.apply_rope(qkv1[:, H * D:(H + Hkv) * D].view(B, Hkv, D), cos[pos1.long()], sin[pos1.long()]).to(dtype)
    v1 = qkv1[:, (H + Hkv) * D:].view(B, Hkv, D)
    # full key set per sequence = prefill keys + the new one
    ks, vs, cuk = [], [], [0]
    for b, l in enumerate(lens):
        ks += [kr[int(cu[b]):int(cu[b + 1])], k1[b:b + 1]]
        vs += [v3[int(cu[b]):int(cu[b + 1])], v1[b:b + 1]]
        cuk.append(cuk[-1] + l + 1)
    want1 = ops_ref.attention_varlen(q1, torch.cat(ks), torch.cat(vs), torch.arange(B + 1), cuk, D ** -0.5)
    cuq1 = torch.arange(B + 1, dtype=torch.int32, device=gpu_device)
    for ns in (1, 2, 3):
        out1 = torch.empty((B, H * D), dtype=dtype, device=gpu_device)
        ws = nat.Workspace(nat.attn_workspace_bytes(B, H, Hkv, D, ns), gpu_device)
        nat.attn_paged(dq1, dq1.stride(0), kpool, vpool, bt.to(gpu_device), ctx1, cuq1, out1, B, H, Hkv, D, 1,
                       max(lens) + 1, D ** -0.5, ns, ws)
        _close(out1.view(B, H, D), want1, rtol=tol, atol=tol, what=f"decode attention splits={ns}")


def test_attention_decode_long_context(nat, gpu_device):
    """cfg3-shaped decode at reduced batch: ctx 1024..1040, MHA D=128, property check vs oracle."""
    dtype, H, Hkv, D, B = torch.float16, 32, 32, 128, 3
    lens = [1024, 1039, 1000]
    g = torch.Generator().manual_seed(9)
    pages_per = [(l + 31) // 32 for l in lens]
    total_pages = sum(pages_per)
    bt = torch.zeros((B, max(pages_per)), dtype=torch.int32)
    perm = torch.randperm(total_pages, generator=g)
    o = 0
    for b in range(B):
        bt[b, :pages_per[b]] = perm[o:o + pages_per[b]].int()
        o += pages_per[b]
    T = sum(lens)
    kv = (torch.randn(T, 2 * Hkv * D, generator=g)).to(dtype)
    dummy = torch.zeros((T, (H + 2 * Hkv) * D), dtype=dtype)
    dummy[:, H * D:] = kv
    slots = torch.cat([bt[b, torch.arange(l) // 32].long() * 32 + torch.arange(l) % 32 for b, l in enumerate(lens)]).int()
    kpool = torch.zeros((total_pages, Hkv, 32 * D), dtype=dtype, device=gpu_device)
    vpool = torch.zeros_like(kpool)
    nat.rope_kv_write(dummy.to(gpu_device), None, None, None, slots.to(gpu_device), kpool, vpool, H, Hkv, D, D)
    q = torch.randn(B, H * D, generator=g).to(dtype)
    cu = [0] + list(np.cumsum(lens))
    want = ops_ref.attention_varlen(q.view(B, H, D), kv[:, :Hkv * D].view(T, Hkv, D), kv[:, Hkv * D:].view(T, Hkv, D),
                                    torch.arange(B + 1), cu, D ** -0.5)
    out = torch.empty((B, H * D), dtype=dtype, device=gpu_device)
    ns = nat.attn_num_splits(B, Hkv, H, 1, max(lens))
    ws = nat.Workspace(nat.attn_workspace_bytes(B, H, Hkv, D, ns), gpu_device)
    nat.attn_paged(q.to(gpu_device), H * D, kpool, vpool, bt.to(gpu_device), torch.tensor(lens, dtype=torch.int32).to(gpu_device),
                   torch.arange(B + 1, dtype=torch.int32, device=gpu_device), out, B, H, Hkv, D, 1, max(lens), D ** -0.5,
                   ns, ws)
    _close(out.view(B, H, D), want, rtol=2e-3, atol=2e-3, what="long decode attention")


@pytest.mark.parametrize("H,Hkv", [(32, 32), (8, 2), (12, 1)])
def test_attention_decode_every_fill_of_the_last_page(nat, gpu_device, H, Hkv):
    """Round 5: the page a sequence is filling is requested only up to its last written token (K's second 16-token tile once
    the sequence has reached it, V's column groups 4 g .. 4 g + 3 up to the last token).  One sequence per fill level
    1 .. 32, against the oracle — with the regions that must NOT be requested poisoned with NaN after the scatter (an
    unrequested register never meets an MFMA), and the unwritten slots the kernel MAY still read (the rest of the tile that is
    being filled, the rest of a started V group) holding large finite garbage that an exact P = 0 has to silence."""
    dtype, D = torch.float16, 128
    lens = [64 + nv for nv in range(1, 33)]
    B = len(lens)
    g = torch.Generator().manual_seed(77)
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
    # every slot starts as large finite garbage, the scatter then writes the real tokens
    # (the extreme finite f16 values: what a reused page's stale tail can hold at worst — the pool's invariant, utils/kv_cache.py,
    # is that every value ever written to it is finite)
    kpool = torch.full((total_pages, Hkv, 32 * D), 65504.0, dtype=dtype, device=gpu_device)
    vpool = torch.full((total_pages, Hkv, 32 * D), -65504.0, dtype=dtype, device=gpu_device)
    nat.rope_kv_write(dummy.to(gpu_device), None, None, None, slots.to(gpu_device), kpool, vpool, H, Hkv, D, D)
    for b, l in enumerate(lens):
        nv, pg = l - 64, int(bt[b, 2])
        if nv <= 16:  # K tile 1 = the second half of the page's [tile][D / 8][16][8] image
            kpool[pg, :, 16 * D:] = float("nan")
        vp = vpool[pg].view(Hkv, 4, D, 8)  # V image [4 column groups][D][8]: group g holds tokens 4 g .. 4 g + 3 of both tiles
        for grp in range(4):
            if 4 * grp >= nv:
                vp[:, grp] = float("nan")
    q = torch.randn(B, H * D, generator=g).to(dtype)
    cu = [0] + list(np.cumsum(lens))
    want = ops_ref.attention_varlen(q.view(B, H, D), kv[:, :Hkv * D].view(T, Hkv, D), kv[:, Hkv * D:].view(T, Hkv, D),
                                    torch.arange(B + 1), cu, D ** -0.5)
    out = torch.empty((B, H * D), dtype=dtype, device=gpu_device)
    ns = nat.attn_num_splits(B, Hkv, H, 1, max(lens))
    ws = nat.Workspace(nat.attn_workspace_bytes(B, H, Hkv, D, ns), gpu_device)
    nat.attn_paged(q.to(gpu_device), H * D, kpool, vpool, bt.to(gpu_device), torch.tensor(lens, dtype=torch.int32).to(gpu_device),
                   torch.arange(B + 1, dtype=torch.int32, device=gpu_device), out, B, H, Hkv, D, 1, max(lens), D ** -0.5,
                   ns, ws)
    assert torch.isfinite(out).all(), "a region that must not be requested reached an MFMA"
    _close(out.view(B, H, D), want, rtol=2e-3, atol=2e-3, what="decode attention over every fill of the last page")


@pytest.mark.parametrize("dtype,B,H,Hkv,D,ctx", [
    (torch.bfloat16, 16, 32, 4, 64, 512),    # cfg2: GQA 8:1, four key splits
    (torch.float16, 4, 32, 32, 128, 700),    # MHA: four (sequence, head) groups share a 128-byte line of {m, l}
    (torch.float16, 2, 48, 1, 128, 1500),    # MQA: three 16-head chunks per block
])
def test_split_decode_attention_merges_in_one_launch_every_time(nat, gpu_device, dtype, B, H, Hkv, D, ctx):
    """Key-split decode attention merges its partial results inside the launch (the last block of each group to arrive
    reads the others' records).  A stale read there — a line of another block's record served from this XCD's L2 or
    L1 — would show as a result that changes between launches that re-use the same workspace, so: 200 launches
    alternating two different q, each compared BIT for bit with the first result for that q, and the first results
    against the un-split launch."""
    g = torch.Generator().manual_seed(B + H + ctx)
    lens = [ctx - 7 * i for i in range(B)]
    pages_per = [(l + 31) // 32 for l in lens]
    total_pages = sum(pages_per)
    bt = torch.zeros((B, max(pages_per)), dtype=torch.int32)
    perm = torch.randperm(total_pages, generator=g)
    o = 0
    for b in range(B):
        bt[b, :pages_per[b]] = perm[o:o + pages_per[b]].int()
        o += pages_per[b]
    T = sum(lens)
    dummy = torch.zeros((T, (H + 2 * Hkv) * D), dtype=dtype)
    dummy[:, H * D:] = torch.randn(T, 2 * Hkv * D, generator=g).to(dtype)
    slots = torch.cat([bt[b, torch.arange(l) // 32].long() * 32 + torch.arange(l) % 32 for b, l in enumerate(lens)]).int()
    kpool = torch.zeros((total_pages, Hkv, 32 * D), dtype=dtype, device=gpu_device)
    vpool = torch.zeros_like(kpool)
    nat.rope_kv_write(dummy.to(gpu_device), None, None, None, slots.to(gpu_device), kpool, vpool, H, Hkv, D, D)
    qs = [torch.randn(B, H * D, generator=g).to(dtype).to(gpu_device) for _ in range(2)]
    ns = max(2, nat.attn_num_splits(B, Hkv, H, 1, max(lens)))
    ws = nat.Workspace(nat.attn_workspace_bytes(B, H, Hkv, D, ns), gpu_device)
    btd, ctxd = bt.to(gpu_device), torch.tensor(lens, dtype=torch.int32).to(gpu_device)
    cuq = torch.arange(B + 1, dtype=torch.int32, device=gpu_device)

    def run(q, splits):
        out = torch.empty((B, H * D), dtype=dtype, device=gpu_device)
        nat.attn_paged(q, H * D, kpool, vpool, btd, ctxd, cuq, out, B, H, Hkv, D, 1, max(lens), D ** -0.5, splits,
                       ws if splits > 1 else None)
        return out

    first = [run(q, ns) for q in qs]
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2
    for q, f in zip(qs, first):
        _close(f, run(q, 1).float().cpu(), rtol=tol, atol=tol, what=f"{ns} key splits against one")
    for i in range(200):
        assert torch.equal(run(qs[i & 1], ns), first[i & 1]), f"launch {i}: the merged result changed"


@pytest.mark.parametrize("dtype,B,H,Hkv,D,ctx,splits", [
    (torch.float16, 8, 64, 1, 128, 300, 1),    # 4 chunks = two blocks per group (3 + 1 chunks), 8 groups
    (torch.bfloat16, 8, 80, 1, 64, 200, 2),    # 5 chunks, two key splits: 16 groups, merged by the combine launch
    (torch.float16, 4, 128, 2, 128, 150, 1),   # two kv heads: group -> (sequence, kv head) decomposition
])
def test_decode_attention_chunk_blocks_share_an_xcd(nat, gpu_device, dtype, B, H, Hkv, D, ctx, splits):
    """Groups wider than 48 heads take several blocks per (sequence, split, kv head); when the number of groups is a
    multiple of 8 the launch remaps block ids so that the chunk blocks of one group run on one XCD (attention.hip,
    xcd_remap).  The remap is a permutation of block ids: every (group, chunk block) must still be computed exactly
    once — checked against the oracle on ragged contexts."""
    g = torch.Generator().manual_seed(B * H + ctx)
    lens = [ctx - 11 * i for i in range(B)]
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
    kpool = torch.zeros((total_pages, Hkv, 32 * D), dtype=dtype, device=gpu_device)
    vpool = torch.zeros_like(kpool)
    nat.rope_kv_write(dummy.to(gpu_device), None, None, None, slots.to(gpu_device), kpool, vpool, H, Hkv, D, D)
    q = torch.randn(B, H * D, generator=g).to(dtype)
    cu = [0] + list(np.cumsum(lens))
    want = ops_ref.attention_varlen(q.view(B, H, D), kv[:, :Hkv * D].view(T, Hkv, D), kv[:, Hkv * D:].view(T, Hkv, D),
                                    torch.arange(B + 1), cu, D ** -0.5)
    out = torch.zeros((B, H * D), dtype=dtype, device=gpu_device)
    ws = nat.Workspace(nat.attn_workspace_bytes(B, H, Hkv, D, splits), gpu_device) if splits > 1 else None
    nat.attn_paged(q.to(gpu_device), H * D, kpool, vpool, bt.to(gpu_device), torch.tensor(lens, dtype=torch.int32).to(gpu_device),
                   torch.arange(B + 1, dtype=torch.int32, device=gpu_device), out, B, H, Hkv, D, 1, max(lens), D ** -0.5,
                   splits, ws)
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2
    _close(out.view(B, H, D), want, rtol=tol, atol=tol, what="chunk blocks on one XCD")


# ---- elementwise / sampling ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_act_mul_gelu_embedding(nat, gpu_device, dtype):
    g = torch.Generator().manual_seed(2)
    gu = torch.randn(7, 2 * 11008, generator=g).to(dtype)
    got = nat.act_mul(gu.to(gpu_device), 11008)
    ulp = 2e-3 if dtype == torch.float16 else 1.6e-2  # two roundings (act, product)
    _close(got, ops_ref.silu_mul(gu, 11008), rtol=ulp, atol=1e-4, what="silu_mul")
    x = torch.randn(5, 256, generator=g).to(dtype)
    for tanh in (True, False):
        _close(nat.gelu(x.to(gpu_device), tanh), ops_ref.gelu(x, tanh), rtol=ulp, atol=1e-3, what="gelu")
    table = torch.randn(100, 64, generator=g).to(dtype)
    ids = torch.tensor([3, 99, 0, 57, 120, -1], dtype=torch.int64)
    got = nat.embedding(ids.to(gpu_device), table.to(gpu_device)).cpu()
    want = torch.zeros(6, 64, dtype=dtype)
    want[:4] = table[ids[:4]]
    assert torch.equal(got, want), "embedding gather (out-of-shard ids give the null row)"


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_argmax_logprob(nat, gpu_device, dtype):
    g = torch.Generator().manual_seed(4)
    logits = (torch.randn(32, 32000, generator=g) * 3).to(dtype)
    logits[5, 100] = logits[5, 7] = logits[5].max() + 1  # tie: lowest index wins (tokens.py Greedy)
    ids, lp = nat.argmax_logprob(logits.to(gpu_device))
    wi, wl = ops_ref.greedy(logits)
    assert torch.equal(ids.cpu(), wi), "greedy token ids must be bit-exact"
    assert int(ids[5]) == 7
    _close(lp, wl, rtol=1e-5, atol=1e-5, what="logprob")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,V", [(32, 32000), (16, 32000), (1, 49152), (3, 1000), (200, 32000), (7, 50257)])
def test_argmax_logprob_rows_split_over_workgroups(nat, gpu_device, dtype, B, V):
    """With the scratch buffer the rows of a small batch are split over several workgroups (two launches): same ids as the
    one-workgroup-per-row form and as the oracle, ties across a segment boundary included."""
    g = torch.Generator().manual_seed(B + V)
    logits = (torch.randn(B, V, generator=g) * 3).to(dtype)
    r = B // 2
    logits[r, V - 1] = logits[r, 3] = logits[r].max() + 1  # a tie between the first and the last segment
    if B > 2:
        logits[0, V // 2 + 11] = logits[0].max() + 2       # a winner in a middle segment
    dev = logits.to(gpu_device)
    ids0, lp0 = nat.argmax_logprob(dev)
    ids1, lp1 = nat.argmax_logprob(dev, scratch=nat.argmax_scratch(B, gpu_device))
    wi, wl = ops_ref.greedy(logits)
    assert torch.equal(ids1, ids0) and torch.equal(ids1.cpu(), wi)
    assert int(ids1[r]) == 3
    _close(lp1, wl, rtol=1e-5, atol=1e-5, what="logprob (split rows)")
    _close(lp1, lp0, rtol=1e-5, atol=1e-5, what="split vs one workgroup per row")


# ---- deferred split-K reduce: GEMM leaves fp32 slabs, the consumer kernel finishes the sum -----------------------
@pytest.mark.parametrize("M,K,N", [(32, 4096, 4096), (5, 11008, 4096), (1, 256, 64), (64, 4096, 4096), (40, 11008, 4096),
                                   (100, 4096, 4096)])
def test_gptq_partial_then_rmsnorm_is_bit_identical_to_unfused(nat, gpu_device, M, K, N):
    gs = 128 if K % 128 == 0 else 64
    qw, qz, sc, gi = ops_ref.make_gptq_tensors(K, N, gs, seed=K + N)
    g = torch.Generator().manual_seed(M)
    x = (torch.randn(M, K, generator=g) * 0.5).half().to(gpu_device)
    res = torch.randn(M, N, generator=g).half().to(gpu_device)
    bias = (torch.randn(N, generator=g) * 0.1).half().to(gpu_device)
    wn = (1 + 0.1 * torch.randn(N, generator=g)).half().to(gpu_device)
    w = nat.GptqWeight(torch.from_numpy(qw).to(gpu_device), torch.from_numpy(qz).to(gpu_device),
                       torch.from_numpy(sc).to(gpu_device), None, 4, gs)
    ws = nat.Workspace(w.workspace_bytes(M), gpu_device)
    y0, r0 = nat.rmsnorm_residual(nat.gptq_gemm(x, w, ws, bias=bias), res, wn, 1e-5)
    part = nat.gptq_gemm_partial(x, w, bias=bias)
    if K >= 4096:
        assert part.S > 1, "this shape is meant to exercise a real split"
    y1, r1 = nat.rmsnorm_residual(part, res, wn, 1e-5)
    assert torch.equal(y0, y1) and torch.equal(r0, r1)
    want_y, want_r = ops_ref.rmsnorm_residual(ops_ref.gptq_linear(x.cpu(), qw, qz, sc, gi, gs, bias.cpu()).half(),
                                              res.cpu(), wn.cpu(), 1e-5)
    _close(y1, want_y, rtol=4e-3, atol=4e-3, what="partial+rmsnorm vs oracle")


def test_gptq_partial_then_rope_kv_is_bit_identical_to_unfused(nat, gpu_device):
    H, Hkv, D, K, B = 8, 8, 128, 4096, 7
    N = (H + 2 * Hkv) * D
    qw, qz, sc, gi = ops_ref.make_gptq_tensors(K, N, 128, seed=17)
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(B, K, generator=g) * 0.5).half().to(gpu_device)
    w = nat.GptqWeight(torch.from_numpy(qw).to(gpu_device), torch.from_numpy(qz).to(gpu_device),
                       torch.from_numpy(sc).to(gpu_device), None, 4, 128)
    ws = nat.Workspace(w.workspace_bytes(B), gpu_device)
    cos, sin = ops_ref.rope_tables(D, 10000.0, 64, torch.float16)
    cos, sin = cos.to(gpu_device), sin.to(gpu_device)
    pos = torch.tensor([3, 0, 31, 32, 17, 5, 63], dtype=torch.int32, device=gpu_device)
    slots = torch.tensor([3, 32, 95, 96, 145, 165, 255], dtype=torch.int32, device=gpu_device)
    pools = [torch.zeros((8, Hkv, 32 * D), dtype=torch.float16, device=gpu_device) for _ in range(4)]
    q0 = nat.rope_kv_write(nat.gptq_gemm(x, w, ws), cos, sin, pos, slots, pools[0], pools[1], H, Hkv, D, D)
    part = nat.gptq_gemm_partial(x, w)
    assert part.S > 1
    q1 = nat.rope_kv_write(part, cos, sin, pos, slots, pools[2], pools[3], H, Hkv, D, D)
    assert torch.equal(q0, q1) and torch.equal(pools[0], pools[2]) and torch.equal(pools[1], pools[3])
    assert pools[0].abs().sum() > 0


@pytest.mark.parametrize("H,Hkv,D,K,B,gs,bias", [(8, 8, 128, 4096, 7, 128, False), (32, 32, 128, 4096, 32, 128, False),
                                                  (8, 1, 128, 1024, 32, 128, True), (4, 4, 64, 512, 5, 64, True),
                                                  (64, 8, 128, 1024, 1, 128, False), (8, 1, 128, 8192, 64, 128, False),
                                                  (32, 32, 128, 4096, 33, 128, True)])
def test_gptq_gemm_rope_equals_gemm_then_rope_kv_write(nat, gpu_device, H, Hkv, D, K, B, gs, bias):
    """tgis_gptq_gemm_rope_f16 (qkv GEMM with the rotary embedding + cache write in its epilogue, on the rope image of the
    weight) against tgis_gptq_gemm_f16 + tgis_rope_kv_write on the plain image: the same f16-rounded activation is rotated
    by the same arithmetic, so q and the cache pages may differ only where the two GEMMs' fp32 sums round differently
    (another split of the k range); both against the oracle's rotary embedding of the oracle's linear."""
    N = (H + 2 * Hkv) * D
    qw, qz, sc, gi = ops_ref.make_gptq_tensors(K, N, gs, seed=H + D + K)
    g = torch.Generator().manual_seed(B)
    x = (torch.randn(B, K, generator=g) * 0.5).half().to(gpu_device)
    bv = (torch.randn(N, generator=g) * 0.1).half().to(gpu_device) if bias else None
    t = [torch.from_numpy(a).to(gpu_device) for a in (qw, qz, sc)]
    w = nat.GptqWeight(t[0], t[1], t[2], None, 4, gs)
    wr = nat.GptqWeight(t[0], t[1], t[2], None, 4, gs, rope=(D, H + Hkv))
    ws = nat.Workspace(w.workspace_bytes(B), gpu_device)
    cos, sin = ops_ref.rope_tables(D, 10000.0, 80, torch.float16)
    cos, sin = cos.to(gpu_device), sin.to(gpu_device)
    pos = torch.randint(0, 80, (B,), generator=g).int().to(gpu_device)
    slots = torch.randperm(8 * 32, generator=g)[:B].int().to(gpu_device)
    pools = [torch.zeros((8, Hkv, 32 * D), dtype=torch.float16, device=gpu_device) for _ in range(4)]
    # (the plain GEMM may hand back a padded-row view: materialise q0 as the rope kernel's own tensor)
    q0 = nat.rope_kv_write(nat.gptq_gemm(x, w, ws, bias=bv), cos, sin, pos, slots, pools[0], pools[1], H, Hkv, D, D)
    q1 = nat.gptq_gemm_rope(x, wr, bv, cos, sin, pos, slots, pools[2], pools[3], H, Hkv, D)
    lin = ops_ref.gptq_linear(x.cpu(), qw, qz, sc, gi, gs, bv.cpu() if bias else None)
    scale = float(lin.abs().max())
    eps = 2.0 ** -11
    for name, a, b in (("q", q0[:, :H * D], q1[:, :H * D]), ("k pages", pools[0], pools[2]), ("v pages", pools[1], pools[3])):
        diff = (a.float() - b.float()).abs()
        assert float(diff.max()) <= 4 * eps * scale, f"{name}: fused and un-fused differ by more than an f16 rounding"
        assert float((diff > 0).float().mean()) < 0.05, f"{name}: more than summation-order noise"
    assert pools[2].abs().sum() > 0 and pools[3].abs().sum() > 0
    # untouched cache slots stay untouched (zeros), written ones match the oracle's rotated k and v
    qkv_ref = lin.half()
    want_q = ops_ref.apply_rope(qkv_ref[:, :H * D].view(B, H, D), cos.cpu()[pos.cpu().long()], sin.cpu()[pos.cpu().long()])
    want_k = ops_ref.apply_rope(qkv_ref[:, H * D:(H + Hkv) * D].view(B, Hkv, D), cos.cpu()[pos.cpu().long()],
                                sin.cpu()[pos.cpu().long()])
    _close(q1[:, :H * D].view(B, H, D), want_q, rtol=4e-3, atol=6 * eps * scale, what="fused rope q vs oracle")
    kp, vp = pools[2].cpu(), pools[3].cpu()
    for b in range(B):
        s_ = int(slots[b])
        Kp, Vp = ops_ref.kv_page_unpack(kp, vp, s_ >> 5, Hkv, D)
        _close(Kp[s_ & 31], want_k[b], rtol=4e-3, atol=6 * eps * scale, what="fused rope k page vs oracle")
        _close(Vp[s_ & 31], qkv_ref[b, (H + Hkv) * D:].view(Hkv, D), rtol=4e-3, atol=6 * eps * scale, what="fused v page")
    used = torch.zeros(8 * 32, dtype=torch.bool)
    used[slots.cpu().long()] = True
    for page in range(8):
        Kp, Vp = ops_ref.kv_page_unpack(kp, vp, page, Hkv, D)
        free = ~used[page * 32:(page + 1) * 32]
        assert float(Kp[free].abs().sum()) == 0 and float(Vp[free].abs().sum()) == 0


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("H,Hkv,D,K,B,bias", [(32, 4, 64, 2048, 16, False), (8, 8, 128, 1024, 7, True), (32, 4, 64, 2048, 40, False),
                                              (4, 1, 32, 256, 1, True)])
def test_dense_gemm_rope_equals_gemm_then_rope_kv_write(nat, gpu_device, dtype, H, Hkv, D, K, B, bias):
    """tgis_dense_gemm_rope (dense qkv GEMM with the rotary embedding + cache write in its epilogue, rope image) against
    tgis_dense_gemm + tgis_rope_kv_write on the plain image, and against the oracle's rotary embedding."""
    N = (H + 2 * Hkv) * D
    g = torch.Generator().manual_seed(H * D + B)
    x = (torch.randn(B, K, generator=g) * 0.5).to(dtype).to(gpu_device)
    wt = (torch.randn(N, K, generator=g) * 0.03).to(dtype).to(gpu_device)
    bv = (torch.randn(N, generator=g) * 0.1).to(dtype).to(gpu_device) if bias else None
    w = nat.DenseWeight(wt)
    wr = nat.DenseWeight(wt, rope=(D, H + Hkv))
    ws = nat.Workspace(w.workspace_bytes(B), gpu_device)
    cos, sin = ops_ref.rope_tables(D, 10000.0, 80, dtype)
    cos, sin = cos.to(gpu_device), sin.to(gpu_device)
    pos = torch.randint(0, 80, (B,), generator=g).int().to(gpu_device)
    slots = torch.randperm(8 * 32, generator=g)[:B].int().to(gpu_device)
    pools = [torch.zeros((8, Hkv, 32 * D), dtype=dtype, device=gpu_device) for _ in range(4)]
    q0 = nat.rope_kv_write(nat.dense_gemm(x, w, ws, bias=bv), cos, sin, pos, slots, pools[0], pools[1], H, Hkv, D, D)
    q1 = nat.dense_gemm_rope(x, wr, bv, cos, sin, pos, slots, pools[2], pools[3], H, Hkv, D)
    lin = x.float().cpu() @ wt.float().cpu().t() + (bv.float().cpu() if bias else 0.0)
    scale = float(lin.abs().max())
    eps = 2.0 ** -11 if dtype == torch.float16 else 2.0 ** -8
    for name, a, b in (("q", q0[:, :H * D], q1[:, :H * D]), ("k pages", pools[0], pools[2]), ("v pages", pools[1], pools[3])):
        diff = (a.float() - b.float()).abs()
        assert float(diff.max()) <= 4 * eps * scale, f"{name}: fused and un-fused differ by more than a rounding"
        assert float((diff > 0).float().mean()) < 0.05, f"{name}: more than summation-order noise"
    assert pools[2].abs().sum() > 0 and pools[3].abs().sum() > 0
    qkv_ref = lin.to(dtype)
    cp, sp = cos.cpu()[pos.cpu().long()], sin.cpu()[pos.cpu().long()]
    want_q = ops_ref.apply_rope(qkv_ref[:, :H * D].view(B, H, D), cp, sp)
    want_k = ops_ref.apply_rope(qkv_ref[:, H * D:(H + Hkv) * D].view(B, Hkv, D), cp, sp)
    _close(q1[:, :H * D].view(B, H, D), want_q, rtol=2 * eps, atol=6 * eps * scale, what="fused rope q vs oracle")
    kp, vp = pools[2].cpu(), pools[3].cpu()
    for b in range(B):
        s_ = int(slots[b])
        Kp, Vp = ops_ref.kv_page_unpack(kp, vp, s_ >> 5, Hkv, D)
        _close(Kp[s_ & 31], want_k[b], rtol=2 * eps, atol=6 * eps * scale, what="fused rope k page vs oracle")
        _close(Vp[s_ & 31], qkv_ref[b, (H + Hkv) * D:].view(Hkv, D), rtol=2 * eps, atol=6 * eps * scale, what="fused v page")


def test_per_op_timing_hooks_and_device_info(nat, gpu_device):
    """tgis_timing_* (HIP events on the op's own stream; bench.py's `roofline` is computed from them) and tgis_device_info."""
    import ctypes

    lib = nat.load_library()
    cus, hbm = ctypes.c_int(0), ctypes.c_int64(0)
    name = ctypes.create_string_buffer(128)
    lib.tgis_device_info.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int64), ctypes.c_char_p, ctypes.c_int]
    assert lib.tgis_device_info(0, ctypes.byref(cus), ctypes.byref(hbm), name, 128) == 0
    assert cus.value >= 64 and hbm.value > (16 << 30) and len(name.value) > 0
    x = torch.randn(32, 4096, device=gpu_device).half()
    w = torch.ones(4096, device=gpu_device).half()
    nat.timing_reset()
    n0, _ = nat.timing_read(nat.OP_NORM)
    nat.rmsnorm_residual(x, None, w, 1e-5)      # timing off: not counted
    nat.timing_enable(True)
    for _ in range(5):
        nat.rmsnorm_residual(x, None, w, 1e-5)
    torch.cuda.synchronize()
    nat.timing_enable(False)
    nat.rmsnorm_residual(x, None, w, 1e-5)
    n, ms = nat.timing_read(nat.OP_NORM)
    assert n0 == 0 and n == 5 and 0.0 < ms < 50.0, (n0, n, ms)
    n_attn, _ = nat.timing_read(nat.OP_ATTN)
    assert n_attn == 0
    nat.timing_reset()
    assert nat.timing_read(nat.OP_NORM)[0] == 0


@pytest.mark.parametrize("B", [1, 7, 64, 200])
def test_decode_advance_equals_the_reference_ops(nat, gpu_device, B):
    """tgis_decode_advance against the reference's statements after a decode step (flash_causal_lm.py:457,499,533-535)."""
    g = torch.Generator().manual_seed(B)
    L = 50
    ids = torch.randint(0, 32000, (B,), generator=g).to(gpu_device)
    pos = torch.randint(0, L - 1, (B,), generator=g).to(gpu_device)
    all_ids = torch.randint(0, 32000, (B + 3, L), generator=g).to(gpu_device)[:B]  # a view with more rows behind it
    cu_q = torch.arange(B + 1, dtype=torch.int32, device=gpu_device)
    cu = (torch.cumsum(torch.randint(1, 40, (B + 1,), generator=g), 0).int() - 1).to(gpu_device)
    want_pos = pos + 1
    want_all = all_ids.clone().scatter_(1, want_pos[:, None], ids[:, None])
    want_cu = cu + cu_q
    st_ids = torch.zeros(B, dtype=torch.int64, device=gpu_device)
    st_pos = torch.zeros(B, dtype=torch.int32, device=gpu_device)
    out = nat.decode_advance(ids, pos, all_ids, cu, cu_q, stage_ids=st_ids, stage_positions=st_pos)
    assert out.data_ptr() != ids.data_ptr() and torch.equal(out, ids)
    assert torch.equal(pos, want_pos) and torch.equal(all_ids, want_all) and torch.equal(cu, want_cu)
    assert torch.equal(st_ids, ids) and torch.equal(st_pos, want_pos.int())
    # optional outputs left out: only the positions move
    pos2 = pos.clone()
    out2 = nat.decode_advance(ids, pos2)
    assert torch.equal(out2, ids) and torch.equal(pos2, want_pos + 1)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("tanh", [False, True])
@pytest.mark.parametrize("M,K,N", [(32, 1024, 24576), (7, 512, 16384 + 40), (32, 4096, 256), (64, 2048, 6144), (1, 256, 72)])
def test_dense_gemm_gelu_equals_gemm_then_gelu(nat, gpu_device, dtype, tanh, M, K, N):
    """tgis_dense_gemm act 4 / 5 (GELU of the rounded sum + bias, in the epilogue of an unsplit plan or in the split-K
    reduce) against tgis_dense_gemm + tgis_gelu, bit for bit, and against the oracle's GELU of the fp32 product."""
    g = torch.Generator().manual_seed(M + K + N)
    x = (torch.randn(M, K, generator=g) * 0.7).to(dtype).to(gpu_device)
    wt = (torch.randn(N, K, generator=g) * 0.04).to(dtype).to(gpu_device)
    bv = (torch.randn(N, generator=g) * 0.2).to(dtype).to(gpu_device)
    w = nat.DenseWeight(wt)
    ws = nat.Workspace(w.workspace_bytes(M), gpu_device)
    two = nat.gelu(nat.dense_gemm(x, w, ws, bias=bv), tanh)
    one = nat.dense_gemm(x, w, ws, bias=bv, act=5 if tanh else 4)
    assert torch.equal(one, two), "GELU in the GEMM differs from GEMM + tgis_gelu"
    lin = (x.float().cpu() @ wt.float().cpu().t() + bv.float().cpu()).to(dtype)
    eps = 2.0 ** -11 if dtype == torch.float16 else 2.0 ** -8
    _close(one, ops_ref.gelu(lin, tanh), rtol=4 * eps, atol=8 * eps * float(lin.abs().max()), what="gelu(x W^T + b) vs oracle")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,K,I", [(16, 2048, 5632), (32, 4096, 11008), (3, 256, 48), (40, 512, 1376), (64, 2048, 5632)])
def test_dense_gemm_gate_up_epilogue(nat, gpu_device, dtype, M, K, I):
    """Dense act=2: [gate | up] projection with SiLU(gate)*up in the epilogue (pairs interleaved at prepare time) against
    the fp32 reference with the reference's rounding points, and against the un-fused pair of launches (plain GEMM, then
    the down projection's act=1 staging of the same values), which may differ only by the summation order."""
    g = torch.Generator().manual_seed(M + K + I)
    x = (torch.randn(M, K, generator=g) * 0.5).to(dtype)
    w = (torch.randn(2 * I, K, generator=g) * 0.03).to(dtype)
    bias = (torch.randn(2 * I, generator=g) * 0.05).to(dtype)
    fused = nat.DenseWeight(w.to(gpu_device), gate_up=True)
    ws = nat.Workspace(fused.workspace_bytes(M), gpu_device)
    got = nat.dense_gemm(x.to(gpu_device), fused, ws, bias=bias.to(gpu_device), act=2)
    assert got.shape == (M, I) and got.dtype == dtype
    lin = (x.float() @ w.float().t() + bias.float()).to(dtype)
    want = torch.nn.functional.silu(lin[:, :I].float()).to(dtype).float() * lin[:, I:].float()
    eps = 2.0 ** -11 if dtype == torch.float16 else 2.0 ** -8
    _close(got, want, rtol=8 * eps, atol=3 * eps * float(lin.float().abs().max()) + 1e-4, what="dense gate_up epilogue")
    plain = nat.DenseWeight(w.to(gpu_device))
    ws.ensure(plain.workspace_bytes(M))
    unfused = nat.act_mul(nat.dense_gemm(x.to(gpu_device), plain, ws, bias=bias.to(gpu_device)), I)
    diff = (got.float() - unfused.float()).abs()
    assert float((diff > 0).float().mean()) < 0.02, "more than summation-order noise between fused and un-fused"
    assert float(diff.max()) <= 4 * eps * float(lin.float().abs().max()) ** 2 + 1e-3


@pytest.mark.parametrize("M,K,I", [(32, 4096, 11008), (3, 256, 48), (40, 512, 1376), (64, 8192, 3584), (64, 2048, 1008),
                                   (33, 1024, 48), (64, 4096, 11008), (32, 4096, 1376), (7, 8192, 3584)])
def test_gptq_gemm_gate_up_epilogue(nat, gpu_device, M, K, I):
    """act=2: fused [gate | up] projection with SiLU(gate)*up in the epilogue (columns interleaved at prepare time),
    and the dequant path must still return the matrix in checkpoint column order.  64-row passes over a narrow matrix
    (the TP shards: (64, 8192, 3584) is a Llama-2-70B gate_up at TP = 8) run split over k with the activation in the
    split-K reduce; wide ones keep the epilogue."""
    gs = 128 if K % 128 == 0 else 64
    N = 2 * I
    qw, qz, sc, gi = ops_ref.make_gptq_tensors(K, N, gs, seed=K + I)
    g = torch.Generator().manual_seed(M + 1)
    x = (torch.randn(M, K, generator=g) * 0.5).half()
    bias = (torch.randn(N, generator=g) * 0.05).half()
    w = nat.GptqWeight(torch.from_numpy(qw).to(gpu_device), torch.from_numpy(qz).to(gpu_device),
                       torch.from_numpy(sc).to(gpu_device), None, 4, gs, gate_up=True)
    ws = nat.Workspace(w.workspace_bytes(M), gpu_device)
    got = nat.gptq_gemm(x.to(gpu_device), w, ws, bias=bias.to(gpu_device), act=2)
    assert got.shape == (M, I)
    lin = ops_ref.gptq_linear(x, qw, qz, sc, gi, gs, bias).half()
    want = torch.nn.functional.silu(lin[:, :I].float()).half().float() * lin[:, I:].float()
    # gate and up are each rounded to f16 (rel 2^-11 of |lin|) before silu(gate)*up: the absolute error of the
    # product scales with max|lin| (error in gate times |up|), not with the size of the (possibly tiny) product
    _close(got, want, rtol=4e-3, atol=1.5e-3 * float(lin.float().abs().max()) + 1e-4, what="gate_up epilogue")
    wd = nat.gptq_dequant(w).float().cpu()
    assert torch.equal(wd, ops_ref.gptq_dequant(qw, qz, sc, gi, gs).half().float())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,K,N", [(16, 2048, 2048), (32, 5632, 2048), (3, 256, 96), (64, 2048, 2048), (40, 5632, 2048)])
def test_dense_partial_then_rmsnorm_is_bit_identical_to_unfused(nat, gpu_device, dtype, M, K, N):
    g = torch.Generator().manual_seed(M + K)
    w = (torch.randn(N, K, generator=g) * 0.02).to(dtype)
    x = (torch.randn(M, K, generator=g) * 0.5).to(dtype).to(gpu_device)
    res = torch.randn(M, N, generator=g).to(dtype).to(gpu_device)
    bias = (torch.randn(N, generator=g) * 0.1).to(dtype).to(gpu_device)
    wn = (1 + 0.1 * torch.randn(N, generator=g)).to(dtype).to(gpu_device)
    dw = nat.DenseWeight(w.to(gpu_device))
    ws = nat.Workspace(dw.workspace_bytes(M), gpu_device)
    y0, r0 = nat.rmsnorm_residual(nat.dense_gemm(x, dw, ws, bias=bias), res, wn, 1e-5)
    part = nat.dense_gemm_partial(x, dw, bias=bias)
    if K >= 2048:
        assert part.S > 1, "this shape is meant to exercise a real split"
    y1, r1 = nat.rmsnorm_residual(part, res, wn, 1e-5)
    assert torch.equal(y0, y1) and torch.equal(r0, r1)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,K,N", [(32, 6144, 6144), (5, 2048, 2048), (3, 256, 96), (64, 5632, 2048)])
def test_dense_partial_then_layernorm_is_bit_identical_to_unfused(nat, gpu_device, dtype, M, K, N):
    """GPT-BigCode's c_proj -> add + LayerNorm with the split-K sum (and the linear's bias) left to the norm kernel."""
    g = torch.Generator().manual_seed(M + K + 1)
    w = (torch.randn(N, K, generator=g) * 0.02).to(dtype)
    x = (torch.randn(M, K, generator=g) * 0.5).to(dtype).to(gpu_device)
    res = torch.randn(M, N, generator=g).to(dtype).to(gpu_device)
    bias = (torch.randn(N, generator=g) * 0.1).to(dtype).to(gpu_device)
    wn = (1 + 0.1 * torch.randn(N, generator=g)).to(dtype).to(gpu_device)
    bn = (0.1 * torch.randn(N, generator=g)).to(dtype).to(gpu_device)
    dw = nat.DenseWeight(w.to(gpu_device))
    ws = nat.Workspace(dw.workspace_bytes(M), gpu_device)
    y0, r0 = nat.layernorm_residual(nat.dense_gemm(x, dw, ws, bias=bias), res, wn, bn, 1e-5)
    part = nat.dense_gemm_partial(x, dw, bias=bias)
    if K >= 2048:
        assert part.S > 1, "this shape is meant to exercise a real split"
    y1, r1 = nat.layernorm_residual(part, res, wn, bn, 1e-5)
    assert torch.equal(y0, y1) and torch.equal(r0, r1)
    want_y, want_r = ops_ref.layernorm_residual(
        (x.float().cpu() @ w.float().t() + bias.float().cpu()).to(dtype), res.cpu(), wn.cpu(), bn.cpu(), 1e-5)
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    _close(r1, want_r, rtol=tol, atol=tol, what="partial + layernorm residual vs oracle")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("H,Hkv,D,rope", [(8, 8, 128, True), (8, 2, 64, True), (12, 1, 128, False)])
def test_rope_kv_write_prefill_is_bit_identical_to_per_token_kernel(nat, gpu_device, dtype, H, Hkv, D, rope):
    """Page-wise prefill cache write == per-token kernel: rotated q rows, every valid cache slot, zeros in the tail of a
    last partial page (the per-token kernel leaves those untouched: compared against a zeroed pool)."""
    lens = [1, 31, 32, 33, 64, 100, 7]
    B, T = len(lens), sum(lens)
    g = torch.Generator().manual_seed(H * 7 + D)
    pages_per = (max(lens) + 31) // 32
    total = B * pages_per + 1
    bt = torch.randperm(total, generator=g)[: B * pages_per].int().view(B, pages_per).contiguous().to(gpu_device)
    qkv = torch.randn(T, (H + 2 * Hkv) * D, generator=g).to(dtype).to(gpu_device)
    cos = sin = None
    if rope:
        cos, sin = ops_ref.rope_tables(D, 10000.0, 128, dtype)
        cos, sin = cos.to(gpu_device), sin.to(gpu_device)
    pos = torch.cat([torch.arange(l) for l in lens]).int().to(gpu_device)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=gpu_device)
    slots = torch.cat([bt[b, torch.arange(l, device=gpu_device) // 32].long() * 32 + torch.arange(l, device=gpu_device) % 32
                       for b, l in enumerate(lens)]).int()
    pools = [(torch.zeros(total, Hkv, 32 * D, dtype=dtype, device=gpu_device),
              torch.zeros(total, Hkv, 32 * D, dtype=dtype, device=gpu_device)),
             (torch.full((total, Hkv, 32 * D), 7.0, dtype=dtype, device=gpu_device),   # stale data the page-wise kernel
              torch.full((total, Hkv, 32 * D), 7.0, dtype=dtype, device=gpu_device))]  # must overwrite in used pages
    q0 = qkv.clone()
    nat.rope_kv_write(q0, cos, sin, pos, slots, pools[0][0], pools[0][1], H, Hkv, D, D)
    q1 = qkv.clone()
    nat.rope_kv_write_prefill(q1, cos, sin, pos, cu, bt, pools[1][0], pools[1][1], max(lens), H, Hkv, D, D)
    torch.cuda.synchronize()
    assert torch.equal(q0[:, :H * D], q1[:, :H * D]), "rotated q"
    assert torch.equal(q1[:, H * D:], qkv[:, H * D:]), "k/v columns of qkv are left as they came"
    used = torch.cat([bt[b, :(l + 31) // 32] for b, l in enumerate(lens)]).long()
    assert torch.equal(pools[0][0][used], pools[1][0][used]), "K pages"
    assert torch.equal(pools[0][1][used], pools[1][1][used]), "V pages"
    unused = torch.tensor(sorted(set(range(total)) - set(used.tolist())), device=gpu_device)
    assert (pools[1][0][unused] == 7.0).all() and (pools[1][1][unused] == 7.0).all(), "pages of other sequences untouched"
