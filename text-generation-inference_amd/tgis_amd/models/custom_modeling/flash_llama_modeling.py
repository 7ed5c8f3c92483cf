"""Llama (MHA / GQA) forward for prefill and batched decode on the gfx950 kernels of libtgis_hip.so.

Mirrors custom_modeling/flash_llama_modeling.py of the reference — `LlamaConfig` (:37-99), `LlamaRMSNorm`
(:102-152), `FlashLlamaAttention` (:183-297), `LlamaMLP` (:300-335), `FlashLlamaLayer` (:338-395),
`FlashLlamaModel` (:398-497), `FlashLlamaForCausalLM` (:500-540) — with these deliberate differences:
  * the KV cache is the paged pool of utils/kv_cache.py instead of a per-batch contiguous tensor, so
    `forward` takes a `KVArgs` (block tables, context lengths, slots) where the reference takes
    `past_key_values` / `pre_allocate_past_size`;
  * RoPE, the KV write and the q/k/v split are one kernel; SiLU*mul is fused into down_proj's operand
    staging; every residual-add is fused into the following RMSNorm (as in the reference);
  * logits are produced in fp32.
Tensor-parallel sharding follows the reference exactly (heads split across ranks, qkv / gate_up
column-parallel, o_proj / down_proj row-parallel + all-reduce, vocab-parallel embedding and head)."""
import os
from dataclasses import dataclass
from typing import List, Optional

import torch

from tgis_amd import native
from tgis_amd.utils.layers import (
    PositionRotaryEmbedding,
    TensorParallelColumnLinear,
    TensorParallelEmbedding,
    TensorParallelHead,
    TensorParallelRowLinear,
)


class LlamaConfig:
    """The subset of the HF Llama config the forward needs (reference :37-99; eps default 1e-6 there)."""

    def __init__(self, vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                 num_attention_heads=32, num_key_value_heads=None, hidden_act="silu", max_position_embeddings=2048,
                 rms_norm_eps=1e-6, rope_scaling=None, rope_theta=10000.0, attention_bias=False, mlp_bias=False,
                 pad_token_id=None, bos_token_id=1, eos_token_id=2, tie_word_embeddings=False, quantize=None,
                 **kwargs):
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.num_key_value_heads = num_key_value_heads if num_key_value_heads is not None else num_attention_heads
        self.hidden_act = hidden_act
        self.max_position_embeddings = max_position_embeddings
        self.rms_norm_eps = rms_norm_eps
        self.rope_scaling = rope_scaling
        self.rope_theta = rope_theta
        self.attention_bias = attention_bias
        self.mlp_bias = mlp_bias
        self.pad_token_id = pad_token_id
        self.bos_token_id = bos_token_id
        self.eos_token_id = eos_token_id
        self.tie_word_embeddings = tie_word_embeddings
        self.quantize = quantize
        self.model_type = "llama"
        for k, v in kwargs.items():
            setattr(self, k, v)

    def to_dict(self):
        return dict(vars(self))


@dataclass
class KVArgs:
    """Where this forward's keys/values live in the paged cache."""
    cache: "object"                    # utils.kv_cache.PagedKVCache
    block_tables: torch.Tensor         # [B, max_pages] int32 (device)
    ctx_lens: Optional[torch.Tensor]   # [B] int32, tokens per sequence incl. this forward's (prefill); decode: filled in
    slots: Optional[torch.Tensor]      # [T] int32 physical slot per token (prefill); decode: filled in
    max_q_len: int                     # longest q run in this forward (1 for decode)
    max_ctx: int                       # upper bound of ctx_lens (launch shaping only)
    num_splits: int = 1                # attention key splits (decode)
    fresh_prefill: bool = False        # every sequence starts at cache position 0: page-wise cache writes


# (The persistent decode tail, the rope-in-attention launch and the norm-as-a-GEMM-phase launches of rounds 2 and 3 were all
# measured slower than the launches they replace: experiments/README.md is their ledger.)


def frag_rows(linear, rows: int, kv: "KVArgs", act: int = 0) -> bool:
    """Decode step of <= 64 rows in front of an int4 linear that takes its operand in fragment order (native.FragAct)."""
    return (kv.max_q_len == 1 and not kv.fresh_prefill and rows <= 64 and hasattr(linear, "wants_fragments")
            and linear.wants_fragments(rows, act))


class LlamaRMSNorm:
    def __init__(self, prefix, weights, eps=1e-6):
        self.weight = weights.get_tensor(f"{prefix}.weight").contiguous()
        self.variance_epsilon = eps

    def forward(self, hidden_states, residual=None, frag: bool = False):
        # returns (normed, res) like the reference; res is hidden_states itself when residual is None
        # frag: the normed activation leaves in fragment order (native.FragAct) for the int4 GEMM behind it
        return native.rmsnorm_residual(hidden_states, residual, self.weight, self.variance_epsilon, frag=frag)

    __call__ = forward


def rope_settings(config):
    """(theta, linear scaling factor) from either config generation: `rope_theta` + `rope_scaling{type, factor}` (the
    reference's transformers 4.40, flash_llama_modeling.py:181-191) or `rope_parameters{rope_theta, rope_type, factor}`
    (transformers >= 5, which dropped the flat attributes)."""
    params = getattr(config, "rope_parameters", None) or {}
    theta = getattr(config, "rope_theta", None)
    if theta is None:
        theta = params.get("rope_theta", 10000.0)
    rs = getattr(config, "rope_scaling", None) or {}
    kind = rs.get("type") or rs.get("rope_type") or params.get("rope_type") or "default"
    if kind in ("default", None):
        return float(theta), 1.0
    if kind == "linear":
        return float(theta), float(rs.get("factor", params.get("factor", 1.0)))
    raise ValueError(f"rope_scaling of type {kind} is not supported")


class FlashLlamaAttention:
    def __init__(self, prefix: str, config, weights):
        self.num_heads = config.num_attention_heads
        self.hidden_size = config.hidden_size
        self.head_size = self.hidden_size // self.num_heads
        theta, scaling = rope_settings(config)
        self.rotary_emb = PositionRotaryEmbedding.static(dim=self.head_size, base=theta,
                                                         device=weights.device, scaling_factor=scaling)
        self.softmax_scale = self.head_size ** -0.5
        tp = weights.process_group.size()
        if self.num_heads % tp != 0:
            raise ValueError(f"`num_heads` must be divisible by `num_shards` (got `num_heads`: {self.num_heads} "
                             f"and `num_shards`: {tp}")
        assert config.num_key_value_heads % tp == 0, "num_key_value_heads must be divisible by the shard count"
        self.num_heads = self.num_heads // tp
        self.num_key_value_heads = config.num_key_value_heads // tp
        self.query_key_value = TensorParallelColumnLinear.load_multi(
            config, prefixes=[f"{prefix}.q_proj", f"{prefix}.k_proj", f"{prefix}.v_proj"], dim=0, weights=weights,
            bias=config.attention_bias)
        self.o_proj = TensorParallelRowLinear.load(config, prefix=f"{prefix}.o_proj", weights=weights,
                                                   bias=config.attention_bias)
        # int4 qkv: decode steps of up to 32 rows rotate q / k and write the cache in the GEMM epilogue
        if hasattr(self.query_key_value.linear, "rope_heads"):
            self.query_key_value.linear.rope_heads = (self.num_heads, self.num_key_value_heads, self.head_size)

    def project_qkv(self, hidden_states, cos, sin, position_ids, cu_seqlens_q, layer_id: int, kv: KVArgs):
        """qkv GEMM, rotation of q and k in place, k and v scattered to their page slots (reference :251-268,282)."""
        H, Hkv, D = self.num_heads, self.num_key_value_heads, self.head_size
        k_pool, v_pool = kv.cache.k_pool(layer_id), kv.cache.v_pool(layer_id)
        lin = self.query_key_value.linear
        rope_w = getattr(lin, "rope_handle", None)
        if isinstance(hidden_states, native.FragAct) or (
                rope_w is not None and kv.slots is not None and not kv.fresh_prefill and kv.max_q_len == 1
                and hidden_states.shape[0] <= 64 and cos.shape[1] * 2 == D
                and native.rope_gemm_ok(hidden_states.shape[0], rope_w, D)):
            # one launch: GEMM + rotary embedding + cache write (native.gptq_gemm_rope / native.dense_gemm_rope)
            fused = native.gptq_gemm_rope if isinstance(rope_w, native.GptqWeight) else native.dense_gemm_rope
            return fused(hidden_states, rope_w, lin.bias, cos, sin, position_ids, kv.slots, k_pool, v_pool, H, Hkv, D)
        # [T, (H + 2 Hkv) D]; at decode sizes the split-K sum of the GPTQ GEMM is finished inside the rope kernel
        qkv = self.query_key_value(hidden_states, partial=True)
        if kv.fresh_prefill and not isinstance(qkv, native.Partial):
            return native.rope_kv_write_prefill(qkv, cos, sin, position_ids, cu_seqlens_q, kv.block_tables, k_pool, v_pool,
                                                kv.max_q_len, H, Hkv, D, D)
        return native.rope_kv_write(qkv, cos, sin, position_ids, kv.slots, k_pool, v_pool, H, Hkv, D, D)

    def attend(self, qkv, cu_seqlens_q, layer_id: int, kv: KVArgs):
        """Attention of the rotated q over the layer's cache pages (reference :271-295): [T, H D]."""
        H, Hkv, D = self.num_heads, self.num_key_value_heads, self.head_size
        k_pool, v_pool = kv.cache.k_pool(layer_id), kv.cache.v_pool(layer_id)
        T = qkv.shape[0]
        B = kv.block_tables.shape[0]
        if frag_rows(self.o_proj.linear, T, kv):  # decode: the o_proj GEMM reads its operand in fragment order
            attn_output = native.FragAct.empty(T, H * D, qkv.device)
        else:
            attn_output = torch.empty((T, H * D), dtype=qkv.dtype, device=qkv.device)
        ws = None
        if kv.num_splits > 1:
            from tgis_amd.utils.layers import workspace
            ws = workspace(qkv.device)
            ws.ensure(native.attn_workspace_bytes(T, H, Hkv, D, kv.num_splits))
        native.attn_paged(qkv, qkv.stride(0), k_pool, v_pool, kv.block_tables, kv.ctx_lens, cu_seqlens_q,
                          attn_output, B, H, Hkv, D, kv.max_q_len, kv.max_ctx, self.softmax_scale, kv.num_splits, ws)
        return attn_output

    def forward(self, hidden_states, cos, sin, position_ids, cu_seqlens_q, layer_id: int, kv: KVArgs):
        qkv = self.project_qkv(hidden_states, cos, sin, position_ids, cu_seqlens_q, layer_id, kv)
        attn_output = self.attend(qkv, cu_seqlens_q, layer_id, kv)
        # may be a native.Partial: the following fused add+RMSNorm finishes the split-K sum
        return self.o_proj(attn_output, partial=True)

    __call__ = forward


class LlamaMLP:
    def __init__(self, prefix, config, weights):
        if config.hidden_act != "silu":
            raise NotImplementedError(f"hidden_act {config.hidden_act}: only silu is wired into the fused kernel")
        self.gate_up_proj = TensorParallelColumnLinear.load_multi(
            config, prefixes=[f"{prefix}.gate_proj", f"{prefix}.up_proj"], weights=weights, dim=0,
            bias=config.mlp_bias)
        self.down_proj = TensorParallelRowLinear.load(config, prefix=f"{prefix}.down_proj", weights=weights,
                                                      bias=config.mlp_bias)
        self.intermediate_size = config.intermediate_size // weights.process_group.size()
        # GPTQ: SiLU(gate)*up runs in the gate_up GEMM epilogue (columns interleaved at prepare time);
        # dense: it runs while down_proj stages its operand.  Reference: eager ops at :332-335.
        self.fused_epilogue = hasattr(self.gate_up_proj.linear, "gate_up") and self.intermediate_size % 16 == 0
        if self.fused_epilogue:
            self.gate_up_proj.linear.gate_up = True

    def forward(self, hidden_states):
        if self.fused_epilogue:
            if isinstance(hidden_states, native.FragAct):  # decode: SiLU * up leaves in the layout down_proj reads
                rows = hidden_states.shape[0]
                act = self.gate_up_proj(hidden_states, out_frag=self.down_proj.linear.wants_fragments(rows))
                return self.down_proj(act, partial=True)
            act = self.gate_up_proj(hidden_states)  # [T, I], already silu(gate) * up
            return self.down_proj(act, partial=True)
        gate_up_states = self.gate_up_proj(hidden_states)  # [T, 2, I]
        return self.down_proj(gate_up_states, act=1, partial=True)

    __call__ = forward


class FlashLlamaLayer:
    def __init__(self, layer_id, config, weights):
        prefix = f"model.layers.{layer_id}"
        self.layer_id = layer_id
        self.self_attn = FlashLlamaAttention(prefix=f"{prefix}.self_attn", config=config, weights=weights)
        self.mlp = LlamaMLP(prefix=f"{prefix}.mlp", config=config, weights=weights)
        self.input_layernorm = LlamaRMSNorm(prefix=f"{prefix}.input_layernorm", weights=weights,
                                            eps=config.rms_norm_eps)
        self.post_attention_layernorm = LlamaRMSNorm(prefix=f"{prefix}.post_attention_layernorm", weights=weights,
                                                     eps=config.rms_norm_eps)

    def forward(self, hidden_states, residual, cos, sin, position_ids, cu_seqlens_q, kv: KVArgs):
        rows = hidden_states.shape[0]
        att = self.self_attn
        # decode steps of <= 32 rows hand the int4 GEMMs their operands in fragment order (native.FragAct): the qkv + rotary
        # launch when it serves the step, the MLP when the SiLU * up epilogue does
        qkv_frag = (kv.slots is not None and cos.shape[1] * 2 == att.head_size
                    and frag_rows(att.query_key_value.linear, rows, kv, act=3))
        normed_hidden_states, res = self.input_layernorm(hidden_states, residual, frag=qkv_frag)
        attn_output = att(normed_hidden_states, cos, sin, position_ids, cu_seqlens_q, self.layer_id, kv)
        mlp_frag = self.mlp.fused_epilogue and frag_rows(self.mlp.gate_up_proj.linear, rows, kv, act=2)
        normed_attn_res_output, attn_res = self.post_attention_layernorm(attn_output, res, frag=mlp_frag)
        mlp_output = self.mlp(normed_attn_res_output)
        return mlp_output, attn_res

    __call__ = forward


class FlashLlamaModel:
    def __init__(self, config, weights):
        self.config = config
        process_group = weights.process_group
        self.tp_rank = process_group.rank()
        self.tp_world_size = process_group.size()
        self.embed_tokens = TensorParallelEmbedding(prefix="model.embed_tokens", weights=weights)
        self.layers = [FlashLlamaLayer(i, config, weights) for i in range(config.num_hidden_layers)]
        self.norm = LlamaRMSNorm(prefix="model.norm", weights=weights, eps=config.rms_norm_eps)
        self.head_size = self.layers[0].self_attn.head_size
        self.num_heads = self.layers[0].self_attn.num_heads
        self.num_key_value_heads = config.num_key_value_heads // process_group.size()
        self.max_positions = 0
    def rope_tables(self, dtype, device, max_s: int):
        # Sized once for the model's whole position range, so the tables normally never move.  A longer request
        # still works: PositionRotaryEmbedding keeps the replaced tables allocated, because decode graphs captured
        # earlier hold their raw pointers (and only ever index positions inside the table they captured).
        if max_s > self.max_positions:
            declared = min(int(getattr(self.config, "max_position_embeddings", 0) or 0), 1 << 17)
            self.max_positions = max(max_s, 2 * self.max_positions, declared, 2048)
        return self.layers[0].self_attn.rotary_emb.tables(dtype, device, self.max_positions)

    def forward(self, input_ids, position_ids, cu_seqlens_q, max_s, inputs_embeds, kv: KVArgs):
        if input_ids is not None and inputs_embeds is not None:
            raise ValueError("You cannot specify both input_ids and inputs_embeds at the same time")
        hidden_states = inputs_embeds if inputs_embeds is not None else self.embed_tokens(input_ids)
        cos, sin = self.rope_tables(hidden_states.dtype, hidden_states.device, max_s)
        residual = None
        for layer in self.layers:
            hidden_states, residual = layer(hidden_states, residual, cos, sin, position_ids, cu_seqlens_q, kv)
        hidden_states, _ = self.norm(hidden_states, residual)
        return hidden_states

    __call__ = forward


class FlashLlamaForCausalLM:
    def __init__(self, config, weights):
        self.config = config
        self.model = FlashLlamaModel(config, weights)
        self.lm_head = TensorParallelHead.load(config, prefix="lm_head", weights=weights)
        self.gptq_linears: List = []
        for layer in self.model.layers:
            for lin in (layer.self_attn.query_key_value.linear, layer.self_attn.o_proj.linear,
                        layer.mlp.gate_up_proj.linear, layer.mlp.down_proj.linear):
                if hasattr(lin, "post_init"):
                    self.gptq_linears.append(lin)

    def post_init(self):
        """Repack every GPTQ linear for the kernels (the reference does this in serve(), server.py:334-358)."""
        for lin in self.gptq_linears:
            if lin.q_handle is None:
                lin.post_init()

    def get_input_embeddings(self):
        return self.model.embed_tokens

    @property
    def num_layers(self):
        return len(self.model.layers)

    def forward(self, input_ids, position_ids, cu_seqlens_q, max_s, inputs_embeds=None, kv: KVArgs = None,
                lm_head_indices: Optional[torch.Tensor] = None):
        """position_ids int32 [T]; returns fp32 logits [T or len(lm_head_indices), vocab]."""
        hidden_states = self.model(input_ids, position_ids, cu_seqlens_q, max_s, inputs_embeds, kv)
        if lm_head_indices is not None:
            hidden_states = hidden_states.index_select(0, lm_head_indices)
        return self.lm_head(hidden_states)

    __call__ = forward
