"""CPU restatement (fp32 torch / numpy) of the arithmetic of every operator on the batched-decode hot path.

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as
the checker; the product path (text-generation-inference_amd/) never imports this package.

Each function cites the reference lines it restates (paths relative to
/root/reference/server/text_generation_server/).  Where the reference delegates to an un-vendored CUDA
extension (flash-attn v2.5.6: flash_attn_2_cuda / dropout_layer_norm / rotary_emb; auto-gptq 0.7.1:
exllamav2_kernels — Dockerfile:6,211-221,281) the published algorithm of that op is restated and anchored
on the reference's own call site; for those ops the reference holds no test or golden vector
 — parity at that boundary is pinned through the CPU `causal_lm` fixtures in
tests/golden/ (see oracle/llama_ref.py), not op by op.
"""
import math
from typing import Optional

import numpy as np
import torch


# ---- GPTQ ------------------------------------------------------------------------------------------
def gptq_pack(intweight: np.ndarray, zeros_true: np.ndarray, bits: int = 4):
    """Pack integer weights [K,N] (values 0..15) and true zero points [G,N] (values 1..16) exactly as
    QuantLinear.pack does (utils/gptq/quant_linear.py:311-345): row r of qweight holds rows 8r..8r+7 in
    nibbles 0..7; qzeros stores (zero - 1) with nibble t of word c = column 8c+t."""
    assert bits == 4
    K, N = intweight.shape
    iw = intweight.astype(np.uint32)
    qweight = np.zeros((K // 8, N), dtype=np.uint32)
    for j in range(8):
        qweight |= iw[j::8] << (4 * j)
    z = (zeros_true.astype(np.int64) - 1).astype(np.uint32)
    G = z.shape[0]
    qzeros = np.zeros((G, N // 8), dtype=np.uint32)
    for j in range(8):
        qzeros |= z[:, j::8] << (4 * j)
    return qweight.astype(np.int32), qzeros.astype(np.int32)


def gptq_dequant(qweight, qzeros, scales, g_idx: Optional[np.ndarray], groupsize: int) -> torch.Tensor:
    """W[k,n] = (q[k,n] - (z[g(k),n] + 1)) * s[g(k),n] in fp32 — the unpack of matmul_248_kernel
    (utils/gptq/quant_linear.py:130-138,159-192; note zeros+1 is NOT masked back to 4 bits)."""
    # (torch integer ops on all host cores: the numpy form of rounds 1-3 took 5 s for one 4096 x 22016 matrix, which made a
    # 32-layer oracle at width unaffordable; same integer unpack, same fp32 expression)
    qw = torch.from_numpy(np.ascontiguousarray(np.asarray(qweight))).to(torch.int32)
    qz = torch.from_numpy(np.ascontiguousarray(np.asarray(qzeros))).to(torch.int32)
    K = qw.shape[0] * 8
    N = qw.shape[1]
    sh = (torch.arange(8, dtype=torch.int32) * 4)
    q = (qw.unsqueeze(1) >> sh.view(1, 8, 1)).bitwise_and_(15).view(K, N)  # row k = pack k // 8, nibble k % 8
    z = ((qz.unsqueeze(2) >> sh.view(1, 1, 8)) & 15).reshape(qz.shape[0], N) + 1  # column n = pack n // 8, nibble n % 8
    if g_idx is None:
        gi = torch.arange(K, dtype=torch.int64) // groupsize
    else:
        gi = torch.from_numpy(np.asarray(g_idx).astype(np.int64))
    s = scales.float().cpu() if torch.is_tensor(scales) else torch.from_numpy(np.asarray(scales, dtype=np.float32))
    if g_idx is None and K % groupsize == 0 and K // groupsize == z.shape[0]:
        G = z.shape[0]
        w = q.view(G, groupsize, N).float()  # in place from here on (fresh 100-MB temporaries cost more than the arithmetic)
        w.sub_(z.float().view(G, 1, N)).mul_(s.view(G, 1, N))
        w = w.view(K, N)
    else:
        w = q.float()
        w.sub_(z.float()[gi]).mul_(s[gi])
    return w  # [K,N] fp32


def gptq_linear(x: torch.Tensor, qweight, qzeros, scales, g_idx, groupsize, bias=None) -> torch.Tensor:
    """y = x @ W (+bias), fp32 accumulate (quant_linear.py:171,194; exllamav2.py:139-144)."""
    w = gptq_dequant(qweight, qzeros, scales, g_idx, groupsize)
    y = x.float() @ w
    if bias is not None:
        y = y + bias.float()
    return y


def make_gptq_tensors(K: int, N: int, groupsize: int, seed: int, act_order: bool = False, w_std: float = 0.02):
    """Seeded synthetic GPTQ tensors at a given shape (the bench generator of SURVEY.md §8d):
    random nibbles, random stored zeros, scales U(0.5,1.5)*2/15*w_std, g_idx trivial or a random
    equal-size-group assignment (act-order)."""
    rng = np.random.default_rng(seed)
    G = K // groupsize
    intw = rng.integers(0, 16, size=(K, N), dtype=np.uint8)
    zeros_true = rng.integers(1, 17, size=(G, N), dtype=np.uint8)
    scales = (rng.uniform(0.5, 1.5, size=(G, N)) * (2.0 / 15.0) * w_std).astype(np.float16)
    if act_order:
        perm = rng.permutation(K)
        g_idx = np.empty(K, dtype=np.int32)
        g_idx[perm] = np.arange(K, dtype=np.int32) // groupsize
    else:
        g_idx = (np.arange(K) // groupsize).astype(np.int32)
    qweight, qzeros = gptq_pack(intw, zeros_true)
    return qweight, qzeros, scales, g_idx


# ---- norms ---------------------------------------------------------------------------------------------
def rmsnorm_residual(x: torch.Tensor, residual: Optional[torch.Tensor], weight: torch.Tensor, eps: float):
    """LlamaRMSNorm.forward (custom_modeling/flash_llama_modeling.py:113-152): res = x (+ residual);
    y = res * rsqrt(mean(res^2) + eps) * weight.  fp32 throughout; returns (y, res)."""
    res = x.float() if residual is None else x.float() + residual.float()
    var = res.pow(2).mean(-1, keepdim=True)
    y = res * torch.rsqrt(var + eps) * weight.float()
    return y, res


def layernorm_residual(x, residual, weight, bias, eps: float):
    """FastLayerNorm.forward (utils/layers.py:363-396): res = x (+ residual); y = LayerNorm(res)."""
    res = x.float() if residual is None else x.float() + residual.float()
    y = torch.nn.functional.layer_norm(res, (res.shape[-1],), weight.float(),
                                       None if bias is None else bias.float(), eps)
    return y, res


# ---- rotary ------------------------------------------------------------------------------------------------
def rope_tables(dim: int, base: float, max_pos: int, dtype: torch.dtype, scaling_factor: float = 1.0):
    """cos/sin caches of PositionRotaryEmbedding (utils/layers.py:419-451): inv_freq = base^(-2i/dim) in
    fp32, freqs = outer(t / factor, inv_freq) in fp32, cos/sin cast to the model dtype."""
    inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    t = torch.arange(max_pos, dtype=torch.float32)
    if scaling_factor != 1.0:
        t = t / scaling_factor
    freqs = torch.outer(t, inv_freq)
    return torch.cos(freqs).to(dtype), torch.sin(freqs).to(dtype)


def apply_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """rotary_emb.apply_rotary with conj=False on x1 = x[..., :r], x2 = x[..., r:2r] (utils/layers.py:466-472):
    x1' = x1 cos - x2 sin ; x2' = x1 sin + x2 cos.  x [T, heads, D]; cos/sin [T, r]."""
    r = cos.shape[-1]
    xf = x.float()
    c = cos.float()[:, None, :]
    s = sin.float()[:, None, :]
    x1, x2 = xf[..., :r], xf[..., r:2 * r]
    out = xf.clone()
    out[..., :r] = x1 * c - x2 * s
    out[..., r:2 * r] = x1 * s + x2 * c
    return out


# ---- attention ----------------------------------------------------------------------------------------------------
def attention_varlen(q, k, v, cu_seqlens_q, cu_seqlens_k, scale: float) -> torch.Tensor:
    """softmax(q k^T * scale) v per sequence with the flash-attn varlen semantics of
    utils/flash_attn.py:43-78: q [Tq,H,D], k/v [Tk,Hkv,D]; sequence b owns q rows cu_q[b]:cu_q[b+1] and key
    rows cu_k[b]:cu_k[b+1]; GQA/MQA by head ratio; the q rows are the LAST q_len positions of the
    sequence and attend causally (bottom-right aligned — identical to causal=True for prefill where
    q_len == k_len, and to causal=False for decode where q_len == 1)."""
    Tq, H, D = q.shape
    Hkv = k.shape[1]
    G = H // Hkv
    out = torch.zeros((Tq, H, D), dtype=torch.float32)
    B = len(cu_seqlens_q) - 1
    for b in range(B):
        q0, q1 = int(cu_seqlens_q[b]), int(cu_seqlens_q[b + 1])
        k0, k1 = int(cu_seqlens_k[b]), int(cu_seqlens_k[b + 1])
        ql, kl = q1 - q0, k1 - k0
        if ql == 0:
            continue
        qb = q[q0:q1].float().transpose(0, 1)  # [H,ql,D]
        kb = k[k0:k1].float().transpose(0, 1).repeat_interleave(G, dim=0)  # [H,kl,D]
        vb = v[k0:k1].float().transpose(0, 1).repeat_interleave(G, dim=0)
        s = torch.matmul(qb, kb.transpose(1, 2)) * scale  # [H,ql,kl]
        qpos = torch.arange(ql)[:, None] + (kl - ql)
        kpos = torch.arange(kl)[None, :]
        s = s.masked_fill(kpos > qpos, float("-inf"))
        p = torch.softmax(s, dim=-1)
        out[q0:q1] = torch.matmul(p, vb).transpose(0, 1)
    return out


# ---- activations / sampling ---------------------------------------------------------------------------------------
def silu_mul(gate_up: torch.Tensor, I: int) -> torch.Tensor:
    """LlamaMLP: act(gate_up[:, 0]) * gate_up[:, 1] on the [T,2,I] view (flash_llama_modeling.py:332-335)."""
    g = gate_up.float()[:, :I]
    u = gate_up.float()[:, I:2 * I]
    return torch.nn.functional.silu(g) * u


def gelu(x: torch.Tensor, tanh_approx: bool) -> torch.Tensor:
    """flash_santacoder_modeling.py:303-307 / LlamaMLP's gelu branch (flash_llama_modeling.py:303-312)."""
    return torch.nn.functional.gelu(x.float(), approximate="tanh" if tanh_approx else "none")


def greedy(logits: torch.Tensor):
    """Greedy + log_softmax + gather (utils/tokens.py:44-46,265-269,394-397): returns (ids, logprobs)."""
    lf = logits.float()
    ids = lf.argmax(dim=-1)
    lp = torch.log_softmax(lf, dim=-1).gather(1, ids[:, None]).squeeze(1)
    return ids, lp


# ---- KV page layout (our own) — used by tests to read the cache back --------------------------------
def kv_page_unpack(k_pool: torch.Tensor, v_pool: torch.Tensor, page: int, Hkv: int, D: int):
    """Return (K[32,Hkv,D], V[32,Hkv,D]) for one page of the pools [num_pages, Hkv, 32*D]."""
    kb = k_pool[page].reshape(Hkv, 2, D // 8, 16, 8)  # [h][tile][chunk][tok][8]
    K = kb.permute(1, 3, 0, 2, 4).reshape(32, Hkv, D)
    vb = v_pool[page].reshape(Hkv, 4, D, 8).permute(0, 2, 1, 3).reshape(Hkv, D, 32)  # [h][column group][d][8] -> [h][d][col]
    tok = torch.arange(32)
    i = tok & 15
    col = (i >> 2) * 8 + (tok >> 4) * 4 + (i & 3)
    V = vb[:, :, col].permute(2, 0, 1)
    return K, V
