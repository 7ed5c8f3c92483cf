"""Op-level tests of the experiments library (python experiments/build.py first; run with
`python -m pytest experiments/tests -m experimental`).  Not collected by the driver's `pytest tests/`: these kernels were
measured slower than the product path and are kept as evidence, bit-identical to it."""
import math
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "text-generation-inference_amd"), os.path.join(ROOT, "experiments")):
    if p not in sys.path:
        sys.path.insert(0, p)
from oracle import ops_ref  # noqa: E402

pytestmark = [pytest.mark.experimental, pytest.mark.gpu]


@pytest.fixture(scope="module")
def gpu_device():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def nat(gpu_device):
    import native_experiments as nx

    if not os.path.exists(os.environ["TGIS_HIP_LIB"]):
        pytest.skip("python experiments/build.py first")
    nx.load_library()
    return nx


def _close(got, want, rtol, atol, what=""):
    got = got.float().cpu()
    want = want.float().cpu()
    err = (got - want).abs()
    bound = atol + rtol * want.abs()
    bad = err > bound
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} out of tolerance, max err {err.max():.4g}"


@pytest.mark.parametrize("dtype,B,H,Hkv,D,ctx,rotary,partial", [
    (torch.float16, 32, 32, 32, 128, 300, True, True),     # cfg3 shape class: MHA, split-K qkv slabs
    (torch.bfloat16, 16, 32, 4, 64, 512, True, True),      # cfg2: GQA 8:1, D = 64, 8-wave blocks
    (torch.float16, 3, 32, 8, 128, 1500, True, False),     # key splits, qkv as a finished tensor (the TP path)
    (torch.bfloat16, 4, 48, 1, 128, 700, False, True),     # GPT-BigCode: MQA in three chunks, no rotation
    (torch.float16, 2, 16, 16, 64, 33, True, True),        # the new token opens a page
    (torch.float16, 5, 32, 8, 128, 1, True, False),        # first decode position: the cache holds only the new token
])
def test_decode_attention_with_rope_and_cache_write_in_its_prologue(nat, gpu_device, dtype, B, H, Hkv, D, ctx, rotary,
                                                                    partial):
    """tgis_attn_decode_rope against tgis_rope_kv_write[_partial] + tgis_attn_paged on the same inputs: same attention
    output and the same bytes in the KV pages (the new token's k / v at its slot, everything else untouched)."""
    g = torch.Generator().manual_seed(B * 7 + H + ctx)
    lens = [max(1, ctx - 5 * i) for i in range(B)]
    pages_per = [(l + 31) // 32 for l in lens]
    total_pages = sum(pages_per) + 1
    bt = torch.zeros((B, max(pages_per)), dtype=torch.int32)
    perm = torch.randperm(total_pages - 1, generator=g) + 1
    o = 0
    for b in range(B):
        bt[b, :pages_per[b]] = perm[o:o + pages_per[b]].int()
        o += pages_per[b]
    N = (H + 2 * Hkv) * D
    # cache contents of the earlier tokens (written through the stand-alone kernel, no rotation needed for the test)
    Tall = sum(l - 1 for l in lens)
    kpool = torch.zeros((total_pages, Hkv, 32 * D), dtype=dtype, device=gpu_device)
    vpool = torch.zeros_like(kpool)
    if Tall:
        old = torch.zeros((Tall, N), dtype=dtype)
        old[:, H * D:] = torch.randn(Tall, 2 * Hkv * D, generator=g).to(dtype)
        slots_old = torch.cat([bt[b, torch.arange(l - 1) // 32].long() * 32 + torch.arange(l - 1) % 32
                               for b, l in enumerate(lens)]).int()
        nat.rope_kv_write(old.to(gpu_device), None, None, None, slots_old.to(gpu_device), kpool, vpool, H, Hkv, D, D)
    # the decode step's qkv projection output: as split-K slabs (two partial sums + bias) or as a tensor
    qkv = torch.randn(B, N, generator=g)
    bias = (torch.randn(N, generator=g) * 0.1).to(dtype) if partial else None
    positions = torch.tensor([l - 1 for l in lens], dtype=torch.int32, device=gpu_device)
    slots = torch.tensor([int(bt[b, (l - 1) // 32]) * 32 + (l - 1) % 32 for b, l in enumerate(lens)], dtype=torch.int32,
                         device=gpu_device)
    cos = sin = None
    if rotary:
        ang = torch.rand(max(lens) + 1, D // 2, generator=g) * 6.28
        cos, sin = ang.cos().to(dtype).to(gpu_device), ang.sin().to(dtype).to(gpu_device)

    def make_input():
        if not partial:
            return qkv.to(dtype).to(gpu_device)
        S, ld = 2, N
        slabs = torch.zeros((1, S, 32, ld), dtype=torch.float32)
        part = torch.randn(B, N, generator=torch.Generator().manual_seed(1))
        slabs[0, 0, :B] = part
        slabs[0, 1, :B] = qkv - part
        p = nat.Partial(slabs.reshape(-1).to(gpu_device), S, ld, B, N, bias.to(gpu_device))
        p.dtype = dtype
        return p

    ns = nat.attn_num_splits(B, Hkv, H, 1, max(lens))
    ws = nat.Workspace(max(4096, nat.attn_workspace_bytes(B, H, Hkv, D, ns)), gpu_device)
    btd, ctxd = bt.to(gpu_device), torch.tensor(lens, dtype=torch.int32).to(gpu_device)
    cuq = torch.arange(B + 1, dtype=torch.int32, device=gpu_device)
    # separate launches
    k1, v1 = kpool.clone(), vpool.clone()
    rot = nat.rope_kv_write(make_input(), cos, sin, positions, slots, k1, v1, H, Hkv, D, D)
    out1 = torch.empty((B, H * D), dtype=dtype, device=gpu_device)
    nat.attn_paged(rot, rot.stride(0), k1, v1, btd, ctxd, cuq, out1, B, H, Hkv, D, 1, max(lens), D ** -0.5, ns, ws)
    # one launch
    k2, v2 = kpool.clone(), vpool.clone()
    out2 = torch.empty((B, H * D), dtype=dtype, device=gpu_device)
    nat.attn_decode_rope(make_input(), cos, sin, positions, slots, k2, v2, btd, ctxd, cuq, out2, B, H, Hkv, D, D,
                         max(lens), D ** -0.5, ns, ws)
    assert torch.equal(k1, k2) and torch.equal(v1, v2), "the KV pages written in the attention prologue differ"
    assert not torch.equal(k2, kpool), "nothing was written"
    assert torch.equal(out1, out2), f"attention output differs (max {(out1.float() - out2.float()).abs().max().item()})"


@pytest.mark.parametrize("B,partial,resid", [(32, True, True), (7, True, True), (1, False, True), (32, False, False)])
def test_norm_phase_gemms_are_bit_identical_to_the_separate_launches(nat, gpu_device, B, partial, resid):
    """tgis_gptq_norm_gate_up_f16 / tgis_gptq_norm_qkv_rope_f16 (add + RMSNorm as the first phase of the GEMM launch: norm
    rows -> grid barrier -> GEMM on L1-bypassing loads) against tgis_rmsnorm_residual[_partial] followed by the plain
    launches: the same arithmetic in the same order, so activation, residual stream, q and cache pages must be
    BIT-identical — a stale line read after the barrier would show up as a difference.  Repeated on re-used buffers."""
    K, I, H, Hkv, D = 4096, 11008, 32, 32, 128
    g = torch.Generator().manual_seed(B + 3)
    qw, qz, sc, gi = ops_ref.make_gptq_tensors(K, 2 * I, 128, seed=5)
    t = [torch.from_numpy(a).to(gpu_device) for a in (qw, qz, sc)]
    w_gu = nat.GptqWeight(t[0], t[1], t[2], None, 4, 128, gate_up=True)
    qw2, qz2, sc2, _ = ops_ref.make_gptq_tensors(K, (H + 2 * Hkv) * D, 128, seed=6)
    t2 = [torch.from_numpy(a).to(gpu_device) for a in (qw2, qz2, sc2)]
    w_qkv = nat.GptqWeight(t2[0], t2[1], t2[2], None, 4, 128, rope=(D, H + Hkv))
    if not (nat.gptq_norm_gemm_ok(B, w_gu, 2) and nat.gptq_norm_gemm_ok(B, w_qkv, 3)):
        pytest.skip("the two-phase launch is not available here (shared GPU or too few CUs)")
    qw3, qz3, sc3, _ = ops_ref.make_gptq_tensors(K, K, 128, seed=7)  # a GEMM that leaves split-K slabs in front of the norm
    w_prev = nat.GptqWeight(*[torch.from_numpy(a).to(gpu_device) for a in (qw3, qz3, sc3)], None, 4, 128)
    wn = (torch.rand(K, generator=g) + 0.5).half().to(gpu_device)
    cos, sin = ops_ref.rope_tables(D, 10000.0, 128, torch.float16)
    cos, sin = cos.to(gpu_device), sin.to(gpu_device)
    ws = nat.Workspace(w_gu.workspace_bytes(B), gpu_device)
    pools = [torch.zeros((8, Hkv, 32 * D), dtype=torch.float16, device=gpu_device) for _ in range(4)]
    for rep in range(6):
        xin = (torch.randn(B, K, generator=g) * 0.5).half().to(gpu_device)
        res = (torch.randn(B, K, generator=g)).half().to(gpu_device) if resid else None
        pos = torch.randint(0, 128, (B,), generator=g).int().to(gpu_device)
        slots = torch.randperm(8 * 32, generator=g)[:B].int().to(gpu_device)

        def src():
            return nat.gptq_gemm_partial(xin, w_prev) if partial else xin.clone()

        y0, r0 = nat.rmsnorm_residual(src(), res, wn, 1e-5)
        a0 = nat.gptq_gemm(y0, w_gu, ws, act=2)
        a1, r1 = nat.gptq_norm_gate_up(src(), res, wn, 1e-5, w_gu)
        assert torch.equal(r0, r1), f"residual rep {rep}: {int((r0 != r1).sum())} elements differ"
        assert torch.equal(a0, a1), (f"gate_up rep {rep}: {int((a0 != a1).sum())} of {a0.numel()} elements differ, max "
                                     f"{float((a0.float() - a1.float()).abs().max())}, rows {sorted(set((a0 != a1).nonzero()[:, 0].tolist()))[:40]}")
        q0 = nat.gptq_gemm_rope(y0, w_qkv, None, cos, sin, pos, slots, pools[0], pools[1], H, Hkv, D)
        q1, r2 = nat.gptq_norm_qkv_rope(src(), res, wn, 1e-5, w_qkv, None, cos, sin, pos, slots, pools[2], pools[3], H, Hkv, D)
        assert torch.equal(q0[:, :H * D], q1[:, :H * D]) and torch.equal(r0, r2), f"qkv rep {rep}"
        assert torch.equal(pools[0], pools[2]) and torch.equal(pools[1], pools[3]), f"cache pages rep {rep}"
    assert nat.gptq_norm_gemm_status() == 0, "a bounded wait of the two-phase launch gave up"


@pytest.mark.parametrize("M", [1, 7, 32])
@pytest.mark.parametrize("K,N,act", [(4096, 1536, 0), (4096, 2048, 2), (2048, 512, 0), (11008, 256, 0)])
def test_lean_gptq_gemm_against_the_exact_dequantised_product(nat, gpu_device, M, K, N, act):
    """The experimental lean int4 GEMM (tgis_gptq_gemm_f16_lean / _partial_lean with tgis_xsum_f16 row sums: nibbles go to
    the MFMA undequantised, zero points and scales are applied per group) against the fp32 product with the dequantised
    weights of the oracle: it never rounds (q - z) * s to f16, so it must be at least as close as the streaming kernel, and
    within 2e-3 of the largest output.  One activation channel is an outlier (30 x) on purpose."""
    gs = 128
    qw, qz, sc, gi = ops_ref.make_gptq_tensors(K, N, gs, seed=K + N)
    t = [torch.from_numpy(a).to(gpu_device) for a in (qw, qz, sc)]
    w = nat.GptqWeight(t[0], t[1], t[2], None, 4, gs, gate_up=act == 2)
    if not nat.gptq_lean_ok(M, w, act):
        pytest.skip("shape outside the lean kernel's plan")
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g).half()
    x[:, 5] *= 30.0
    xd = x.to(gpu_device)
    xs = nat.xsum(xd)
    want_xs = torch.stack([x.float().view(M, K // 16, 4, 4)[:, :, :, :2].sum((2, 3)),
                           x.float().view(M, K // 16, 4, 4)[:, :, :, 2:].sum((2, 3))], -1)
    _close(xs, want_xs, rtol=1e-5, atol=1e-3, what="row sums (k % 4 < 2 | k % 4 >= 2 of every 16 columns)")
    ws = nat.Workspace(w.workspace_bytes(M), gpu_device)
    wd = ops_ref.gptq_dequant(qw, qz, sc, gi, gs).double()
    ref = x.double() @ wd
    if act == 2:
        I = N // 2
        ref = torch.nn.functional.silu(ref[:, :I].half().float()).half().float() * ref[:, I:].half().float()
    ref = ref.float()
    scale = float(ref.abs().max())
    old = nat.gptq_gemm(xd, w, ws, act=act).float().cpu()
    new = nat.gptq_gemm_lean(xd, xs, w, ws, act=act).float().cpu()
    e_old, e_new = float((old - ref).abs().max()) / scale, float((new - ref).abs().max()) / scale
    assert nat.gptq_lean_status(reset=True) == 0, "a bounded wait inside the kernel gave up"
    assert e_new < 2e-3 and e_new <= 1.5 * e_old + 2.0 ** -11, f"lean {e_new:.2e} vs streaming {e_old:.2e}"
    if act == 0:
        p = nat.gptq_gemm_partial_lean(xd, xs, w)
        sl = p.slabs[: p.S * 32 * p.ld].view(p.S, 32, p.ld).sum(0)[:M, :N].cpu()
        assert float((sl - ref).abs().max()) / scale < 1e-4, "fp32 split-K slabs of the lean kernel"
        assert nat.gptq_lean_status(reset=True) == 0


