"""TEST-ONLY stand-in for tgis_amd.native on machines without a GPU: every op is replaced by the CPU oracle so that
the HOST logic above the kernels (tensor-parallel sharding, collectives, model wiring, batch bookkeeping) can be
exercised under gloo.  Installed only by tests through `install()`; the product never imports this module and
tgis_amd.native itself has no fallback."""
import torch

from oracle import ops_ref


class _Workspace:
    def __init__(self, nbytes, device):
        self.nbytes = nbytes

    def ensure(self, n):
        pass

    ptr = 0


class _GptqWeight:
    def __init__(self, qweight, qzeros, scales, g_idx, bits, groupsize, gate_up=False, rope=None):
        # (the rope image only re-orders columns inside the device image: the stand-in keeps the natural order)
        self.flags = 1 if gate_up else (2 if rope is not None else 0)
        self.K, self.N = qweight.shape[0] * 8, qweight.shape[1]
        self.groups = qzeros.shape[0]
        self.perm = None
        self.in_features = self.K
        explicit = None
        if isinstance(g_idx, tuple):  # act-order row shard: rows in image order + gather index (-1 = zero activation)
            _, explicit, self.in_features = g_idx
            g_idx = None
        gi = None if g_idx is None else g_idx.cpu().numpy()
        self.w = ops_ref.gptq_dequant(qweight.cpu().numpy(), qzeros.cpu().numpy(), scales.float().cpu(), gi,
                                      self.K // self.groups)
        if explicit is not None:  # fold the gather into the matrix: W_x[c] = sum of the image rows that read column c
            wx = torch.zeros((self.in_features, self.N), dtype=self.w.dtype)
            real = explicit >= 0
            wx[explicit[real].long()] = self.w[real]
            self.w = wx

    def workspace_bytes(self, M):
        return 0

    def fused_rows(self, act=0):
        return 256


class _DenseWeight:
    def __init__(self, weight, gate_up=False, rope=None):
        self.N, self.K = weight.shape
        self.dtype = weight.dtype
        self.flags = 1 if gate_up else 0
        self.w = weight.float().t().contiguous()

    def workspace_bytes(self, M):
        return 0


def _act(x, K):
    return ops_ref.silu_mul(x, K).to(x.dtype).float()


def _gptq_gemm(x, w, ws, bias=None, act=0, out=None):
    xf = _act(x, w.in_features) if act == 1 else x.float()
    y = xf @ w.w
    if bias is not None:
        y = y + bias.float()
    y = y.to(torch.float16)
    if act == 2:
        y = ops_ref.silu_mul(y, w.N // 2).to(torch.float16)
    return y


def _dense_gemm(x, w, ws, bias=None, out_f32=False, act=0, out=None):
    xf = _act(x, w.K) if act == 1 else x.float()
    y = xf @ w.w
    if bias is not None:
        y = y + bias.float()
    if act == 2:
        return ops_ref.silu_mul(y.to(w.dtype), w.N // 2).to(w.dtype)
    if act in (4, 5):
        return ops_ref.gelu(y.to(w.dtype), act == 5).to(w.dtype)
    return y if out_f32 else y.to(w.dtype)


def _rmsnorm(x, residual, weight, eps, y=None, res_out=None, frag=False):
    assert not frag  # gptq_fragments_ok is False on this backend
    yy, res = ops_ref.rmsnorm_residual(x, residual, weight, eps)
    return yy.to(x.dtype), (res.to(x.dtype) if residual is not None else x)


def _layernorm(x, residual, weight, bias, eps, y=None, res_out=None):
    yy, res = ops_ref.layernorm_residual(x, residual, weight, bias, eps)
    return yy.to(x.dtype), (res.to(x.dtype) if residual is not None else x)


_POOLS = {}


def _rope_kv_write(qkv, cos, sin, positions, slots, k_pool, v_pool, H, Hkv, D, rot_dim):
    T = qkv.shape[0]
    q = qkv[:, :H * D].view(T, H, D)
    k = qkv[:, H * D:(H + Hkv) * D].view(T, Hkv, D)
    v = qkv[:, (H + Hkv) * D:(H + 2 * Hkv) * D].view(T, Hkv, D)
    if cos is not None:
        c, s = cos[positions.long()], sin[positions.long()]
        q.copy_(ops_ref.apply_rope(q, c, s).to(qkv.dtype))
        k.copy_(ops_ref.apply_rope(k, c, s).to(qkv.dtype))
    store = _POOLS.setdefault(k_pool.data_ptr(), {})
    for t in range(T):
        store[int(slots[t])] = (k[t].clone(), v[t].clone())
    return qkv


def _rope_kv_write_prefill(qkv, cos, sin, positions, cu_seqlens, block_tables, k_pool, v_pool, max_len, H, Hkv, D, rot_dim):
    # by contract the per-token kernel with slot(b, i) = block_tables[b][i // 32] * 32 + i % 32
    cu = [int(v) for v in cu_seqlens]
    slots = torch.tensor([int(block_tables[b, i // 32]) * 32 + i % 32
                          for b in range(len(cu) - 1) for i in range(cu[b + 1] - cu[b])], dtype=torch.int32)
    return _rope_kv_write(qkv, cos, sin, positions, slots, k_pool, v_pool, H, Hkv, D, rot_dim)


def _attn_paged(q, ld_q, k_pool, v_pool, block_tables, ctx_lens, cu_q, out, B, H, Hkv, D, max_q_len, max_ctx, scale,
                num_splits, ws):
    store = _POOLS[k_pool.data_ptr()]
    T = out.shape[0]
    qv = q[:, :H * D].reshape(T, H, D)
    for b in range(B):
        q0, q1 = int(cu_q[b]), int(cu_q[b + 1])
        ctx = int(ctx_lens[b])
        slots = [int(block_tables[b, p // 32]) * 32 + p % 32 for p in range(ctx)]
        kb = torch.stack([store[s][0] for s in slots])
        vb = torch.stack([store[s][1] for s in slots])
        o = ops_ref.attention_varlen(qv[q0:q1], kb, vb, [0, q1 - q0], [0, ctx], scale)
        out[q0:q1] = o.reshape(q1 - q0, H * D).to(out.dtype)
    return out


def _embedding(ids, table, positions=None, pos_table=None, id_offset=0, out=None):
    local = ids - id_offset
    ok = (local >= 0) & (local < table.shape[0])
    e = torch.zeros((ids.numel(), table.shape[1]), dtype=table.dtype)
    e[ok] = table[local[ok]]
    if pos_table is not None:
        e = (e.float() + pos_table[positions.long()].float()).to(table.dtype)
    return e


def _decode_slots(positions, block_tables, slots, ctx_lens):
    for b in range(positions.numel()):
        p = int(positions[b])
        slots[b] = int(block_tables[b, p // 32]) * 32 + p % 32
        ctx_lens[b] = p + 1


def _argmax_logprob(logits, ids_out=None, logprob_out=None, scratch=None):
    return ops_ref.greedy(logits)


def install(monkeypatch):
    from tgis_amd import native
    from tgis_amd.utils import layers

    _POOLS.clear()
    layers._WORKSPACES.clear()
    for name, fn in dict(
        Workspace=_Workspace, GptqWeight=_GptqWeight, DenseWeight=_DenseWeight, gptq_gemm=_gptq_gemm,
        dense_gemm=_dense_gemm, rmsnorm_residual=_rmsnorm, layernorm_residual=_layernorm,
        rope_kv_write=_rope_kv_write, rope_kv_write_prefill=_rope_kv_write_prefill, attn_paged=_attn_paged, embedding=_embedding, decode_slots=_decode_slots,
        argmax_logprob=_argmax_logprob, attn_num_splits=lambda *a: 1, attn_workspace_bytes=lambda *a: 0,
        act_mul=lambda gu, I, out=None: ops_ref.silu_mul(gu, I).to(gu.dtype),
        gptq_gemm_partial=lambda x, w, bias=None, act=0: _gptq_gemm(x, w, None, bias=bias, act=act),
        gptq_rope_ok=lambda M, w, D: 1 <= M <= 64 and w.perm is None,
        gptq_fragments_ok=lambda M, w, act=0: False,  # fragment order is a device layout: the CPU stand-ins stay row-major
        rope_gemm_ok=lambda M, w, D: 1 <= M <= 64 and getattr(w, "perm", None) is None,
        dense_gemm_rope=lambda x, w, bias, cos, sin, positions, slots, k_pool, v_pool, H, Hkv, D, out=None: _rope_kv_write(
            _dense_gemm(x, w, None, bias=bias), cos, sin, positions, slots, k_pool, v_pool, H, Hkv, D, D),
        gptq_gemm_rope=lambda x, w, bias, cos, sin, positions, slots, k_pool, v_pool, H, Hkv, D, out=None: _rope_kv_write(
            _gptq_gemm(x, w, None, bias=bias), cos, sin, positions, slots, k_pool, v_pool, H, Hkv, D, D),
        dense_gemm_partial=lambda x, w, bias=None, act=0: _dense_gemm(x, w, None, bias=bias, act=act),
    ).items():
        monkeypatch.setattr(native, name, fn)
