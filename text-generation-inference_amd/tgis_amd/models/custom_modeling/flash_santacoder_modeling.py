"""GPT-BigCode (Santacoder / Starcoder, multi-query attention) forward on the gfx950 kernels.

Mirrors custom_modeling/flash_santacoder_modeling.py of the reference: `load_multi_mqa` (:19-160: q sharded across
ranks, the single k/v head replicated on every rank), `load_col`/`load_row` (:163-192), `FlashMQAttention` (:194-278),
`MLP` (:281-307), `Block` (:310-353), `FlashSantacoderModel` (:356-459: wte + wpe partial embeddings then ONE
all-reduce, :408-414) and `FlashSantacoderForCausalLM` (:462-498, lm_head tied to wte).
Differences are the same as for Llama (paged KV through `KVArgs`, fp32 logits); there is no RoPE, so
tgis_rope_kv_write runs with cos = NULL and only scatters k/v into their page slots."""
from typing import List, Optional

import os

import torch
import torch.distributed

from tgis_amd import native
from tgis_amd.utils.graph_segments import collective
from tgis_amd.models.custom_modeling.flash_llama_modeling import KVArgs
from tgis_amd.utils.layers import (
    FastLinear,
    TensorParallelColumnLinear,
    TensorParallelEmbedding,
    TensorParallelHead,
    TensorParallelRowLinear,
    get_linear,
    workspace,
)


def _shard_q_keep_kv(t: torch.Tensor, dim: int, head_size: int, rank: int, world: int) -> torch.Tensor:
    """[..., H*D + 2*D] along `dim`: this rank's slice of the q part followed by the full (replicated) kv part."""
    n = t.shape[dim]
    assert (n - 2 * head_size) % world == 0
    block = (n - 2 * head_size) // world
    q = t.narrow(dim, rank * block, block)
    kv = t.narrow(dim, n - 2 * head_size, 2 * head_size)
    return torch.cat([q, kv], dim=dim)


def load_multi_mqa(config, prefix: str, weights, bias: bool, head_size, num_heads, hidden_size):
    world, rank = weights.process_group.size(), weights.process_group.rank()
    if config.quantize == "gptq":
        if not weights.has(f"{prefix}.c_attn.qweight") or getattr(config, "transpose", False):
            raise NotImplementedError("Gptq loading with santacoder is not implemented")
        qweight = _shard_q_keep_kv(weights._full(f"{prefix}.c_attn.qweight"), 1, head_size, rank, world)
        scales = _shard_q_keep_kv(weights._full(f"{prefix}.c_attn.scales"), 1, head_size, rank, world)
        assert 2 * head_size % 8 == 0
        qzeros = _shard_q_keep_kv(weights._full(f"{prefix}.c_attn.qzeros"), 1, head_size // 8, rank, world)
        g_idx = weights.get_tensor(f"{prefix}.c_attn.g_idx")
        bits, groupsize = weights._get_gptq_params()
        weight = (qweight.to(weights.device), qzeros.to(weights.device), scales.to(weights.device), g_idx, bits,
                  groupsize, True)
        b = None
        if bias:
            b = weights._finish(_shard_q_keep_kv(weights._full(f"{prefix}.c_attn.bias"), 0, head_size, rank, world))
        return TensorParallelColumnLinear(get_linear(weight, b, config.quantize))
    if weights.has(f"{prefix}.c_attn.weight"):
        w = weights._full(f"{prefix}.c_attn.weight")
        if getattr(config, "transpose", False):  # GPT2-style Conv1D checkpoints store [in, out]
            weight = _shard_q_keep_kv(w, 1, head_size, rank, world).T
        else:
            weight = _shard_q_keep_kv(w, 0, head_size, rank, world)
        b = _shard_q_keep_kv(weights._full(f"{prefix}.c_attn.bias"), 0, head_size, rank, world) if bias else None
    else:
        if getattr(config, "transpose", False):
            weight = torch.cat([weights.get_sharded(f"{prefix}.q_attn.weight", dim=1).T,
                                weights.get_tensor(f"{prefix}.kv_attn.weight").T], dim=0)
        else:
            weight = torch.cat([weights.get_sharded(f"{prefix}.q_attn.weight", dim=0),
                                weights.get_tensor(f"{prefix}.kv_attn.weight")], dim=0)
        b = torch.cat([weights.get_sharded(f"{prefix}.q_attn.bias", dim=0),
                       weights.get_tensor(f"{prefix}.kv_attn.bias")], dim=0) if bias else None
    weight = weights._finish(weight).contiguous()
    assert list(weight.shape) == [(num_heads + 2) * head_size, hidden_size], \
        f"{list(weight.shape)} != {[(num_heads + 2) * head_size, hidden_size]}"
    if b is not None:
        b = weights._finish(b).contiguous()
    return TensorParallelColumnLinear(get_linear(weight, b, config.quantize))


def load_col(config, prefix: str, weights, bias: bool):
    if getattr(config, "transpose", False):
        weight = weights.get_sharded(f"{prefix}.weight", dim=1).T.contiguous()
    else:
        weight = weights.get_multi_weights_col([prefix], quantize=config.quantize, dim=0)
    b = weights.get_sharded(f"{prefix}.bias", dim=0) if bias else None
    return TensorParallelColumnLinear(get_linear(weight, b, config.quantize))


def load_row(config, prefix: str, weights, bias: bool):
    if getattr(config, "transpose", False):
        weight = weights.get_sharded(f"{prefix}.weight", dim=0).T.contiguous()
    else:
        weight = weights.get_multi_weights_row(prefix, quantize=config.quantize)
    b = weights.get_tensor(f"{prefix}.bias") if bias and weights.process_group.rank() == 0 else None
    return TensorParallelRowLinear(get_linear(weight, b, config.quantize), process_group=weights.process_group)


class FastLayerNorm:
    def __init__(self, prefix, weights, eps):
        self.weight = weights.get_tensor(f"{prefix}.weight").contiguous()
        self.bias = weights.get_tensor(f"{prefix}.bias").contiguous()
        self.eps = eps

    def forward(self, hidden_states, residual=None):
        return native.layernorm_residual(hidden_states, residual, self.weight, self.bias, self.eps)

    __call__ = forward


class FlashMQAttention:
    def __init__(self, prefix, config, weights):
        self.hidden_size = config.hidden_size
        self.head_size = config.hidden_size // config.num_attention_heads
        tp = weights.process_group.size()
        if config.num_attention_heads % tp != 0:
            raise ValueError(f"`num_heads` must be divisible by `num_shards` (got `num_heads`: "
                             f"{config.num_attention_heads} and `num_shards`: {tp}")
        self.num_heads = config.num_attention_heads // tp
        self.softmax_scale = self.head_size ** (-0.5)
        self.c_attn = load_multi_mqa(config, prefix=prefix, weights=weights, bias=True, head_size=self.head_size,
                                     hidden_size=config.hidden_size, num_heads=self.num_heads)
        self.c_proj = load_row(config, prefix=f"{prefix}.c_proj", weights=weights, bias=True)

    def forward(self, hidden_states, position_ids, cu_seqlens_q, layer_id: int, kv: KVArgs):
        H, D = self.num_heads, self.head_size
        # [T, (H + 2) D]: H query heads, then the single k and v heads; at decode sizes the split-K sum is left to
        # the cache-write kernel below (native.Partial)
        qkv = self.c_attn(hidden_states, partial=True)
        k_pool, v_pool = kv.cache.k_pool(layer_id), kv.cache.v_pool(layer_id)
        if kv.fresh_prefill and not isinstance(qkv, native.Partial):
            qkv = native.rope_kv_write_prefill(qkv, None, None, None, cu_seqlens_q, kv.block_tables, k_pool, v_pool,
                                               kv.max_q_len, H, 1, D, D)  # no rotary: page-wise cache write only
        else:
            qkv = native.rope_kv_write(qkv, None, None, None, kv.slots, k_pool, v_pool, H, 1, D, D)  # no rotary
        T = qkv.shape[0]
        attn_output = torch.empty((T, H * D), dtype=qkv.dtype, device=qkv.device)
        ws = None
        if kv.num_splits > 1:
            ws = workspace(qkv.device)
            ws.ensure(native.attn_workspace_bytes(T, H, 1, D, kv.num_splits))
        native.attn_paged(qkv, qkv.stride(0), k_pool, v_pool, kv.block_tables, kv.ctx_lens, cu_seqlens_q,
                          attn_output, kv.block_tables.shape[0], H, 1, D, kv.max_q_len, kv.max_ctx,
                          self.softmax_scale, kv.num_splits, ws)
        return self.c_proj(attn_output, partial=True)  # summed by the following add + LayerNorm

    __call__ = forward


class MLP:
    def __init__(self, prefix, config, weights):
        act = config.activation_function
        if "gelu" not in act:
            raise NotImplementedError(f"activation {act}: only the gelu family is wired in")
        self.tanh = act in ("gelu_fast", "gelu_pytorch_tanh")
        self.c_fc = load_col(config, prefix=f"{prefix}.c_fc", weights=weights, bias=True)
        self.c_proj = load_row(config, prefix=f"{prefix}.c_proj", weights=weights, bias=True)

    def forward(self, hidden_states):
        if isinstance(self.c_fc.linear, FastLinear):
            h = self.c_fc(hidden_states, gelu=self.tanh)  # GELU where the decode GEMM finishes its output
        else:
            h = native.gelu(self.c_fc(hidden_states), self.tanh)
        return self.c_proj(h, partial=True)  # summed by the next block's add + LayerNorm

    __call__ = forward


class Block:
    def __init__(self, layer_id, config, weights):
        prefix = f"transformer.h.{layer_id}"
        self.layer_id = layer_id
        self.ln_1 = FastLayerNorm(f"{prefix}.ln_1", weights, config.layer_norm_epsilon)
        self.ln_2 = FastLayerNorm(f"{prefix}.ln_2", weights, config.layer_norm_epsilon)
        self.attn = FlashMQAttention(prefix=f"{prefix}.attn", config=config, weights=weights)
        self.mlp = MLP(prefix=f"{prefix}.mlp", config=config, weights=weights)

    def forward(self, hidden_states, residual, position_ids, cu_seqlens_q, kv: KVArgs):
        hidden_states, residual = self.ln_1(hidden_states, residual)
        hidden_states = self.attn(hidden_states, position_ids, cu_seqlens_q, self.layer_id, kv)
        hidden_states, residual = self.ln_2(hidden_states, residual)
        return self.mlp(hidden_states), residual

    __call__ = forward


class FlashSantacoderModel:
    def __init__(self, config, weights):
        self.config = config
        self.process_group = weights.process_group
        self.wte = TensorParallelEmbedding(prefix="transformer.wte", weights=weights, reduce=False)
        self.wpe = TensorParallelEmbedding(prefix="transformer.wpe", weights=weights, reduce=False)
        self.h = [Block(i, config, weights) for i in range(config.num_hidden_layers)]
        self.layers = self.h
        self.ln_f = FastLayerNorm("transformer.ln_f", weights, config.layer_norm_epsilon)
        self.head_size = self.h[0].attn.head_size
        self.num_heads = self.h[0].attn.num_heads
        self.num_key_value_heads = 1  # the single kv head is replicated on every rank

    def forward(self, input_ids, position_ids, cu_seqlens_q, max_s, inputs_embeds, kv: KVArgs):
        if input_ids is not None and inputs_embeds is not None:
            raise ValueError("You cannot specify both input_ids and inputs_embeds at the same time")
        pos = self.wpe(position_ids.to(torch.int64))
        tok = inputs_embeds if inputs_embeds is not None else self.wte(input_ids)
        hidden_states = tok + pos  # partial sums of both vocab-sharded tables ...
        if self.process_group.size() > 1:  # ... completed by ONE all-reduce (reference :408-414)
            pg = self.process_group
            # bound now: a replayed seam must keep reducing THIS tensor, not whatever `hidden_states` names later
            collective(lambda t=hidden_states: torch.distributed.all_reduce(t, group=pg))
        residual = None
        for layer in self.h:
            hidden_states, residual = layer(hidden_states, residual, position_ids, cu_seqlens_q, kv)
        hidden_states, _ = self.ln_f(hidden_states, residual)
        return hidden_states

    __call__ = forward


class FlashSantacoderForCausalLM:
    def __init__(self, config, weights):
        self.config = config
        self.transformer = FlashSantacoderModel(config, weights)
        self.lm_head = TensorParallelHead.load(config, prefix="transformer.wte", weights=weights)  # tied
        self.gptq_linears: List = []
        for blk in self.transformer.h:
            for lin in (blk.attn.c_attn.linear, blk.attn.c_proj.linear, blk.mlp.c_fc.linear, blk.mlp.c_proj.linear):
                if hasattr(lin, "post_init"):
                    self.gptq_linears.append(lin)

    @property
    def model(self):
        return self.transformer

    def post_init(self):
        for lin in self.gptq_linears:
            if lin.q_handle is None:
                lin.post_init()

    def get_input_embeddings(self):
        return self.transformer.wte

    def forward(self, input_ids, position_ids, cu_seqlens_q, max_s, inputs_embeds=None, kv: KVArgs = None,
                lm_head_indices: Optional[torch.Tensor] = None):
        hidden_states = self.transformer(input_ids, position_ids, cu_seqlens_q, max_s, inputs_embeds, kv)
        if lm_head_indices is not None:
            hidden_states = hidden_states.index_select(0, lm_head_indices)
        return self.lm_head(hidden_states)

    __call__ = forward
