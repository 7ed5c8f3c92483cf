"""-m gpu: the add + RMSNorm around a decode GEMM folded INTO the GEMMs (round 6, tgis_fold in include/tgis_hip.h) against the
oracle's unfused chain — ops_ref.rmsnorm_residual, then the fp32 matmul / ops_ref.gptq_linear, with the reference's rounding
points (flash_llama_modeling.py:113-152 LlamaRMSNorm, :251-268 qkv + rotary, :332-335 SiLU * up).

The folded launches are NOT bit-identical to tgis_rmsnorm_residual + GEMM, by construction:
  * the statistics come from the rounded residual stream h = T(x + residual) (the reference's torch branch for hidden sizes
    above 8192 does the same; its dropout_layer_norm branch, and tgis_rmsnorm_residual, take them from the unrounded fp32 sum):
    a relative change of rstd of at most ~eps_T / sqrt(K);
  * the operand that reaches the MFMA is T(h * w), and rstd scales the fp32 accumulator — the reference rounds h * rstd * w
    once per element: one rounding per operand element either way, at a different place.
So the bound is the bound of any f16 / bf16 GEMM on a rounded operand: |err| <= c * eps_T * scale of the output, with eps_T =
2^-11 (f16) / 2^-8 (bf16); every test states its c.  The residual stream itself (the producer's output) IS exact: the same two
roundings as the reference, checked to one ulp of summation-order noise."""
import numpy as np
import pytest
import torch

from oracle import ops_ref

pytestmark = pytest.mark.gpu

EPS_T = {torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8}


@pytest.fixture(scope="module")
def nat(gpu_device):
    from tgis_amd import native

    native.load_library()
    return native


def _max_err(got, want):
    return float((got.float().cpu() - want.float().cpu()).abs().max())


def _inputs(M, K, N, dtype, seed, w_std=0.03):
    g = torch.Generator().manual_seed(seed)
    h = (torch.randn(M, K, generator=g) * 0.8).to(dtype)            # the residual stream
    nw = (1 + 0.2 * torch.randn(K, generator=g)).to(dtype)          # RMSNorm weight
    w = (torch.randn(N, K, generator=g) * w_std).to(dtype)
    bias = (torch.randn(N, generator=g) * 0.05).to(dtype)
    return h, nw, w, bias


# TinyLlama (cfg2) and Llama-7B widths, the batch sizes of the configs and ragged ones
SHAPES = [(16, 2048), (32, 4096), (1, 2048), (7, 1024), (32, 2048)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,K", SHAPES)
@pytest.mark.parametrize("bias", [False, True])
def test_producer_residual_epilogue(nat, gpu_device, dtype, M, K, bias):
    """o_proj / down_proj with the residual add in the epilogue: out = T(T(x W^T + b) + residual) — exactly the residual stream
    the reference's next fused add + RMSNorm would return (`res`), up to the summation order of the GEMM."""
    for N, KK in ((K, K), (K, 2 * K + 1536)):                       # o_proj (K x K) and a down projection (longer k range)
        x, _nw, w, b = _inputs(M, KK, N, dtype, seed=M + KK + N)
        g = torch.Generator().manual_seed(5)
        res = torch.randn(M, N, generator=g).to(dtype)
        dw = nat.DenseWeight(w.to(gpu_device))
        assert nat.dense_fold_ok(M, dw, 0, False)
        got = nat.dense_gemm_fold(x.to(gpu_device), dw, bias=b.to(gpu_device) if bias else None, out_residual=res.to(gpu_device))
        lin = (x.float() @ w.float().t() + (b.float() if bias else 0.0)).to(dtype)
        want = (lin.float() + res.float()).to(dtype)
        e = EPS_T[dtype]
        err = (got.float().cpu() - want.float()).abs()
        # one rounding of the sum may land on the other side (summation order), then one more rounding of the add: 2 ulps of
        # the larger of the two values involved (an ulp is up to 2 eps_T of the value)
        bound = 4 * e * torch.maximum(lin.float().abs(), want.float().abs()) + 1e-6
        assert bool((err <= bound).all()), f"N={N}: max err {float(err.max()):.4g}"
        assert float((err > 0).float().mean()) < 0.05, "more than summation-order noise"
        # and against the two launches it replaces (same kernel family, unsplit vs split plan): identical up to that noise
        ws = nat.Workspace(dw.workspace_bytes(M), gpu_device)
        two = (nat.dense_gemm(x.to(gpu_device), dw, ws, bias=b.to(gpu_device) if bias else None).float() + res.to(gpu_device).float()).to(dtype)
        assert float((two != got).float().mean()) < 0.05


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,K", SHAPES)
def test_consumer_gate_up_with_folded_norm(nat, gpu_device, dtype, M, K):
    """gate_up + SiLU * up on the residual stream: RMSNorm(h) formed while the operand is staged.  Bound: c = 12 on the scale of
    the pre-activation (the SiLU * up product of two such values doubles the relative error of each factor)."""
    I = {1024: 2816, 2048: 5632, 4096: 11008}[K]
    h, nw, w, b = _inputs(M, K, 2 * I, dtype, seed=M + K)
    dw = nat.DenseWeight(w.to(gpu_device), gate_up=True)
    assert nat.dense_fold_ok(M, dw, 2, True)
    got = nat.dense_gemm_fold(h.to(gpu_device), dw, bias=b.to(gpu_device), act=2, norm_weight=nw.to(gpu_device), eps=1e-5)
    y, _ = ops_ref.rmsnorm_residual(h, None, nw, 1e-5)
    lin = (y.to(dtype).float() @ w.float().t() + b.float()).to(dtype)
    want = torch.nn.functional.silu(lin[:, :I].float()).to(dtype).float() * lin[:, I:].float()
    e, scale = EPS_T[dtype], float(lin.float().abs().max())
    err = _max_err(got, want)
    assert err <= 12 * e * scale * max(1.0, scale), f"max err {err:.4g} (pre-activation scale {scale:.3g})"
    # the unfused product path (norm launch + GEMM) sits inside the same bound around the oracle
    ws = nat.Workspace(dw.workspace_bytes(M), gpu_device)
    yn, _ = nat.rmsnorm_residual(h.to(gpu_device), None, nw.to(gpu_device), 1e-5)
    two = nat.dense_gemm(yn, dw, ws, bias=b.to(gpu_device), act=2)
    assert _max_err(got, two) <= 12 * e * scale * max(1.0, scale)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,K,H,Hkv,D", [(16, 2048, 32, 4, 64), (32, 4096, 32, 32, 128), (5, 2048, 32, 4, 64), (1, 1024, 8, 2, 128)])
def test_consumer_qkv_rope_with_folded_norm(nat, gpu_device, dtype, M, K, H, Hkv, D):
    """qkv + rotary embedding + cache write on the residual stream.  q against the oracle (c = 8 on the scale of the projection),
    k / v pages against what the unfused product chain writes (same bound)."""
    N = (H + 2 * Hkv) * D
    h, nw, w, b = _inputs(M, K, N, dtype, seed=M + K + H)
    wr = nat.DenseWeight(w.to(gpu_device), rope=(D, H + Hkv))
    assert nat.dense_fold_ok(M, wr, 3, True)
    g = torch.Generator().manual_seed(3)
    cos, sin = ops_ref.rope_tables(D, 10000.0, 96, dtype)
    pos = torch.randint(0, 96, (M,), generator=g).int()
    slots = torch.randperm(8 * 32, generator=g)[:M].int()
    pools = [torch.zeros((8, Hkv, 32 * D), dtype=dtype, device=gpu_device) for _ in range(4)]
    dev = lambda t: t.to(gpu_device)
    got = nat.dense_gemm_rope_fold(dev(h), wr, dev(b), dev(cos), dev(sin), dev(pos), dev(slots), pools[0], pools[1], H, Hkv, D,
                                   norm_weight=dev(nw), eps=1e-5)
    y, _ = ops_ref.rmsnorm_residual(h, None, nw, 1e-5)
    lin = (y.to(dtype).float() @ w.float().t() + b.float()).to(dtype)
    cp, sp = cos[pos.long()], sin[pos.long()]
    want_q = ops_ref.apply_rope(lin[:, :H * D].view(M, H, D), cp, sp)
    e, scale = EPS_T[dtype], float(lin.float().abs().max())
    err = _max_err(got[:, :H * D].view(M, H, D), want_q)
    assert err <= 8 * e * scale, f"q: max err {err:.4g} (scale {scale:.3g})"
    yn, _ = nat.rmsnorm_residual(dev(h), None, dev(nw), 1e-5)
    nat.dense_gemm_rope(yn, wr, dev(b), dev(cos), dev(sin), dev(pos), dev(slots), pools[2], pools[3], H, Hkv, D)
    assert pools[0].abs().sum() > 0 and pools[1].abs().sum() > 0
    assert _max_err(pools[0], pools[2]) <= 8 * e * scale, "k pages"
    assert _max_err(pools[1], pools[3]) <= 8 * e * scale, "v pages"
    # untouched slots stay untouched
    written = torch.zeros(8 * 32, dtype=torch.bool)
    written[slots.long()] = True
    from oracle.ops_ref import kv_page_unpack
    for pg in range(8):
        K_, V_ = kv_page_unpack(pools[0].float().cpu(), pools[1].float().cpu(), pg, Hkv, D)
        free = ~written[pg * 32:(pg + 1) * 32]
        assert float(K_[free].abs().sum()) == 0.0 and float(V_[free].abs().sum()) == 0.0


def test_fold_argument_validation(nat, gpu_device):
    """The entry points refuse what they do not implement, with a message (no silent fallback)."""
    dtype = torch.bfloat16
    x = torch.zeros(40, 2048, dtype=dtype, device=gpu_device)
    w = nat.DenseWeight(torch.zeros(2048, 2048, dtype=dtype, device=gpu_device))
    assert not nat.dense_fold_ok(40, w, 0, False)                   # more than 32 rows
    with pytest.raises(nat.TgisHipError, match="no folded launch"):
        nat.dense_gemm_fold(x, w, out_residual=torch.zeros(40, 2048, dtype=dtype, device=gpu_device))
    small = nat.DenseWeight(torch.zeros(64, 512, dtype=dtype, device=gpu_device))
    assert not nat.dense_fold_ok(8, small, 0, False)                # k range too short for four k-parts
    f = nat.Fold(None, 0.0, None, None, None)
    rc = nat.load_library().tgis_dense_gemm_fold(x.data_ptr(), 2048, w.image.data_ptr(), None, x.data_ptr(), 2048, 8, 2048, 2048,
                                                 nat.BF16, 0, 0, f, None)
    assert rc != 0 and b"neither a norm nor a residual" in nat.load_library().tgis_last_error()
    nat.clear_error()
