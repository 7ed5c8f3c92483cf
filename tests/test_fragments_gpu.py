"""-m gpu: the decode step's fragment-order activations (TGIS_LD_FRAGMENTS, csrc/gptq_wide_body.h).

The int4 GEMM of <= 32 rows that reads its operand in MFMA-fragment order against (a) the oracle's formula
(oracle/ops_ref.gptq_linear: utils/gptq/quant_linear.py:130-138,184-194) and (b) the row-major launch on the same
inputs — same arithmetic, another summation order over k, so the two may differ by an f16 rounding in a few outputs;
and the three producers (add + RMSNorm, decode attention, the SiLU * up epilogue) against their row-major forms,
which must be the same bits in another place."""
import pytest
import torch

from oracle import ops_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat(gpu_device):
    from tgis_amd import native

    native.load_library()
    return native


def _weight(nat, dev, K, N, gs, seed, **kw):
    qw, qz, sc, gi = ops_ref.make_gptq_tensors(K, N, gs, seed=seed)
    t = [torch.from_numpy(a).to(dev) for a in (qw, qz, sc)]
    return (qw, qz, sc, gi), nat.GptqWeight(t[0], t[1], t[2], None, 4, gs, **kw)


def test_fragment_order_round_trip_and_offsets(nat, gpu_device):
    """FragAct.from_rows / to_rows are inverse, and element (m, k) sits where include/tgis_hip.h says."""
    for M, K in ((7, 256), (45, 128)):
        x = (torch.arange(M * K, dtype=torch.float32) % 2039).view(M, K).half().to(gpu_device)
        f = nat.FragAct.from_rows(x)
        assert torch.equal(f.to_rows(), x)
        buf = f.buf.cpu()
        for m, k in ((0, 0), (3, 9), (6, 63), (5, 64 + 37), (2, K - 1), (M - 1, 70)):
            off = (m // 32) * 32 * K + ((k // 64 * 4 + k // 8 % 4) * 64 + 32 * (k // 32 % 2) + m % 32) * 8 + k % 8
            assert buf[off] == x[m, k].cpu()


@pytest.mark.parametrize("M,K,N,gs", [
    (32, 4096, 12288, 128),   # cfg3 qkv (plain image)
    (32, 4096, 4096, 128),    # cfg3 o_proj: four k splits
    (32, 11008, 4096, 128),   # cfg3 down_proj: 172 k64-steps over 4 splits x 8 waves (ragged shares)
    (1, 4096, 4096, 128),
    (7, 1024, 160, 64),       # five tiles: a ragged last column group
    (19, 512, 96, 512),       # one group
    (32, 1408, 4096, 64),     # a row-parallel shard: 22 steps
    (5, 64, 64, 64),          # one k64-step: seven of the eight waves have nothing to do
    (32, 8192, 1024, 128),    # 70B shard shapes
    (64, 4096, 4096, 128),    # two row blocks: every dequantised fragment feeds two MFMAs
    (40, 11008, 4096, 128),   # ragged second row block
    (33, 1024, 160, 64),
    (64, 8192, 7168, 128),    # 70B gate_up shard at TP = 8
])
def test_fragment_gemm_matches_the_oracle_and_the_row_major_launch(nat, gpu_device, M, K, N, gs):
    (qw, qz, sc, gi), w = _weight(nat, gpu_device, K, N, gs, seed=K + N + M)
    g = torch.Generator().manual_seed(M * 11 + 3)
    x = (torch.randn(M, K, generator=g) * 0.5).half()
    bias = (torch.randn(N, generator=g) * 0.1).half()
    want = ops_ref.gptq_linear(x, qw, qz, sc, gi, gs, bias)
    assert nat.gptq_fragments_ok(M, w, 0)
    ws = nat.Workspace(w.workspace_bytes(M), gpu_device)
    xd = x.to(gpu_device)
    xf = nat.FragAct.from_rows(xd)
    if M % 32:  # rows past M must not matter: poison them (lanes M % 32 .. 31 of both halves in the last row block)
        last = xf.buf.view((M + 31) // 32, -1, 2, 32, 8)[-1]
        last[:, :, M % 32:].fill_(float("nan"))
    got = nat.gptq_gemm(xf, w, ws, bias=bias.to(gpu_device))
    row = nat.gptq_gemm(xd, w, ws, bias=bias.to(gpu_device))
    tol = dict(rtol=2e-3, atol=2e-3 * float(want.abs().mean()) + 1e-4)
    err = (got.float().cpu() - want).abs()
    assert bool((err <= tol["atol"] + tol["rtol"] * want.abs()).all()), f"vs oracle: max err {float(err.max()):.4g}"
    d = (got.float() - row.float()).abs()
    assert float(d.max()) <= 2.0 ** -9 * float(want.abs().max()) + 1e-3, "fragment and row-major launches disagree"
    assert torch.equal(got, nat.gptq_gemm(xf, w, ws, bias=bias.to(gpu_device))), "not deterministic"
    # deferred reduce: the slabs' sum (+ bias) rounds to the same f16 tensor
    p = nat.gptq_gemm_partial(xf, w, bias=bias.to(gpu_device))
    rb = (M + 31) // 32
    sl = p.slabs[:rb * p.S * 32 * p.ld].view(rb, p.S, 32, p.ld).sum(1).reshape(rb * 32, p.ld)[:M, :N] + bias.to(gpu_device).float()
    dd = (sl.half().float() - got.float()).abs()
    assert float(dd.max()) <= 2.0 ** -9 * float(want.abs().max()) + 1e-3


@pytest.mark.parametrize("order", [(16, 48), (48, 16)])
def test_deferred_reduce_plan_follows_the_row_class(nat, gpu_device, order):
    """One GptqWeight serving a <= 32-row and then a 33 - 64-row decode batch (continuous batching: concatenate / prune move a
    batch across 32 rows): the split plan of the fragment-order kernel differs between the two row classes on long k ranges
    (plan_wide: 70B down_proj S 2 vs 4), so the cached (S, ld) of one class must never describe the slabs of the other."""
    K, N, gs = 28672, 1024, 128
    (qw, qz, sc, gi), w = _weight(nat, gpu_device, K, N, gs, seed=77)
    seen = []
    for M in order:
        g = torch.Generator().manual_seed(M)
        x = (torch.randn(M, K, generator=g) * 0.25).half()
        want = ops_ref.gptq_linear(x, qw, qz, sc, gi, gs, None)
        xf = nat.FragAct.from_rows(x.to(gpu_device))
        for _ in range(2):  # first call of a row class asks the library, the second one uses the cached plan
            p = nat.gptq_gemm_partial(xf, w)
            rb = (M + 31) // 32
            got = p.slabs[:rb * p.S * 32 * p.ld].view(rb, p.S, 32, p.ld).sum(1).reshape(rb * 32, p.ld)[:M, :N]
            err = (got.cpu() - want).abs()
            assert bool((err <= 2e-3 * float(want.abs().mean()) + 1e-4 + 2e-3 * want.abs()).all()), (M, p.S, float(err.max()))
        seen.append(p.S)
    assert len(w.partial_plan) == 2, "one cached plan per row class"


@pytest.mark.parametrize("M,K,I", [(32, 4096, 11008), (9, 1024, 1408), (32, 2048, 5632), (64, 4096, 11008), (45, 8192, 3584),
                                   (64, 8192, 3584),   # 70B shard at TP = 8: two k splits, SiLU * up in the reduce launch
                                   (32, 4096, 1376)])  # 7B shard at TP = 8: 86 one-tile blocks, unsplit
def test_fragment_gemm_silu_epilogue_and_fragment_output(nat, gpu_device, M, K, I):
    """gate_up on the interleaved image with a fragment-order operand: row-major and fragment-order outputs hold the same
    bits, and equal the row-major launch up to summation order."""
    (qw, qz, sc, gi), w = _weight(nat, gpu_device, K, 2 * I, 128, seed=K + I, gate_up=True)
    g = torch.Generator().manual_seed(I)
    x = (torch.randn(M, K, generator=g) * 0.5).half().to(gpu_device)
    ws = nat.Workspace(w.workspace_bytes(M), gpu_device)
    xf = nat.FragAct.from_rows(x)
    a_row = nat.gptq_gemm(xf, w, ws, act=2)
    ok_frag = I % 64 == 0
    if ok_frag:
        a_frag = nat.gptq_gemm(xf, w, ws, act=2, out_frag=True)
        assert isinstance(a_frag, nat.FragAct) and torch.equal(a_frag.to_rows(), a_row)
    base = nat.gptq_gemm(x, w, ws, act=2)
    lin = ops_ref.gptq_linear(x.cpu(), qw, qz, sc, gi, 128, None)
    want = ops_ref.silu_mul(lin.half().view(M, 2 * I), I)
    scale = float(want.abs().max())
    assert float((a_row.float().cpu() - want.float()).abs().max()) <= 6e-3 * scale
    assert float((a_row.float() - base.float()).abs().max()) <= 4e-3 * scale


@pytest.mark.parametrize("H,Hkv,D,K,B,gs,bias", [(32, 32, 128, 4096, 32, 128, False), (8, 8, 128, 4096, 7, 128, True),
                                                  (32, 4, 64, 2048, 16, 64, False), (64, 8, 128, 1024, 1, 128, False),
                                                  (32, 32, 128, 4096, 64, 128, False), (8, 1, 128, 8192, 50, 128, True)])
def test_fragment_gemm_rope_epilogue(nat, gpu_device, H, Hkv, D, K, B, gs, bias):
    """tgis_gptq_gemm_rope_f16 with a fragment-order operand against the same launch on the row-major operand: q and the
    cache pages agree up to the summation order of the two k partitions."""
    N = (H + 2 * Hkv) * D
    (qw, qz, sc, gi), wr = _weight(nat, gpu_device, K, N, gs, seed=H + D + K, rope=(D, H + Hkv))
    g = torch.Generator().manual_seed(B)
    x = (torch.randn(B, K, generator=g) * 0.5).half().to(gpu_device)
    bv = (torch.randn(N, generator=g) * 0.1).half().to(gpu_device) if bias else None
    cos, sin = ops_ref.rope_tables(D, 10000.0, 80, torch.float16)
    cos, sin = cos.to(gpu_device), sin.to(gpu_device)
    pos = torch.randint(0, 80, (B,), generator=g).int().to(gpu_device)
    slots = torch.randperm(8 * 32, generator=g)[:B].int().to(gpu_device)
    pools = [torch.zeros((8, Hkv, 32 * D), dtype=torch.float16, device=gpu_device) for _ in range(4)]
    q0 = nat.gptq_gemm_rope(x, wr, bv, cos, sin, pos, slots, pools[0], pools[1], H, Hkv, D)
    q1 = nat.gptq_gemm_rope(nat.FragAct.from_rows(x), wr, bv, cos, sin, pos, slots, pools[2], pools[3], H, Hkv, D)
    lin = ops_ref.gptq_linear(x.cpu(), qw, qz, sc, gi, gs, bv.cpu() if bias else None)
    scale = float(lin.abs().max())
    eps = 2.0 ** -11
    for name, a, b in (("q", q0[:, :H * D], q1[:, :H * D]), ("k pages", pools[0], pools[2]), ("v pages", pools[1], pools[3])):
        diff = (a.float() - b.float()).abs()
        assert float(diff.max()) <= 4 * eps * scale, f"{name}: more than an f16 rounding apart"
        assert float((diff > 0).float().mean()) < 0.05, f"{name}: more than summation-order noise"
    assert pools[2].abs().sum() > 0 and pools[3].abs().sum() > 0


@pytest.mark.parametrize("rows,hidden,partial", [(32, 4096, True), (5, 4096, False), (17, 2048, True), (1, 256, False),
                                                 (64, 8192, True), (47, 4096, False)])
def test_rmsnorm_writes_the_same_bits_in_fragment_order(nat, gpu_device, rows, hidden, partial):
    g = torch.Generator().manual_seed(rows + hidden)
    res = (torch.randn(rows, hidden, generator=g)).half().to(gpu_device)
    wn = (1 + 0.1 * torch.randn(hidden, generator=g)).half().to(gpu_device)
    if partial:
        S, ld = 3, hidden
        slabs = torch.randn((rows + 31) // 32, S, 32, ld, generator=g).to(gpu_device)
        src = lambda: nat.Partial(slabs.reshape(-1), S, ld, rows, hidden, None)  # noqa: E731
    else:
        x = torch.randn(rows, hidden, generator=g).half().to(gpu_device)
        src = lambda: x  # noqa: E731
    y0, r0 = nat.rmsnorm_residual(src(), res, wn, 1e-5)
    y1, r1 = nat.rmsnorm_residual(src(), res, wn, 1e-5, frag=True)
    assert isinstance(y1, nat.FragAct) and torch.equal(y1.to_rows(), y0) and torch.equal(r0, r1)


@pytest.mark.parametrize("B,H,Hkv,D,ctx", [(32, 32, 32, 128, 300), (16, 32, 4, 64, 512), (3, 32, 8, 128, 1500), (4, 48, 1, 128, 700),
                                           (64, 8, 1, 128, 600), (50, 32, 32, 128, 100)])
def test_decode_attention_writes_the_same_bits_in_fragment_order(nat, gpu_device, B, H, Hkv, D, ctx):
    """tgis_attn_paged with ld_out = TGIS_LD_FRAGMENTS (every combine path: in-block, in-launch merge, combine launch)."""
    g = torch.Generator().manual_seed(B + H + ctx)
    lens = [max(1, ctx - 7 * i) for i in range(B)]
    pages_per = [(l + 31) // 32 for l in lens]
    npages = sum(pages_per) + 2
    kp = (torch.randn(npages, Hkv, 32 * D, generator=g) * 0.5).half().to(gpu_device)
    vp = (torch.randn(npages, Hkv, 32 * D, generator=g) * 0.5).half().to(gpu_device)
    bt = torch.zeros((B, max(pages_per)), dtype=torch.int32)
    nxt = 1
    for b, n in enumerate(pages_per):
        bt[b, :n] = torch.arange(nxt, nxt + n)
        nxt += n
    q = (torch.randn(B, H * D, generator=g) * 0.5).half().to(gpu_device)
    ctxd = torch.tensor(lens, dtype=torch.int32, device=gpu_device)
    cuq = torch.arange(B + 1, dtype=torch.int32, device=gpu_device)
    btd = bt.to(gpu_device)
    for ns in sorted({1, nat.attn_num_splits(B, Hkv, H, 1, max(lens))}):
        ws = None
        if ns > 1:
            ws = nat.Workspace(nat.attn_workspace_bytes(B, H, Hkv, D, ns), gpu_device)
        o0 = torch.empty((B, H * D), dtype=torch.float16, device=gpu_device)
        nat.attn_paged(q, H * D, kp, vp, btd, ctxd, cuq, o0, B, H, Hkv, D, 1, max(lens), D ** -0.5, ns, ws)
        o1 = nat.FragAct.empty(B, H * D, gpu_device)
        nat.attn_paged(q, H * D, kp, vp, btd, ctxd, cuq, o1, B, H, Hkv, D, 1, max(lens), D ** -0.5, ns, ws)
        assert torch.equal(o1.to_rows(), o0), f"{ns} splits"
