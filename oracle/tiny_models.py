"""Seeded tiny Llama checkpoints for parity tests.  TEST INFRASTRUCTURE ONLY.

The weights are a pure function of (config, seed): the fixture generator (tests/golden/make_fixtures.py, run in
the survey container against the imported reference) and the tests on the GPU box rebuild the identical tensors,
so only outputs need to be committed as fixtures.  Design notes (SURVEY.md §8c "Fixture design note"):
random-init models give near-uniform logits, so the lm_head is scaled up until the greedy margin is far above
fp16 noise, and GPTQ tensors come from real min/max quantisation of zero-mean float weights (random nibbles
would give every matrix a large input-independent mean and make the argmax insensitive to the prompt)."""
from typing import Dict, Optional

import numpy as np
import torch

from oracle import ops_ref


class TinyLlamaConfig:
    model_type = "llama"

    def __init__(self, vocab_size=256, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                 num_attention_heads=4, num_key_value_heads=2, rms_norm_eps=1e-5, rope_theta=10000.0,
                 max_position_embeddings=512):
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.num_key_value_heads = num_key_value_heads
        self.rms_norm_eps = rms_norm_eps
        self.rope_theta = rope_theta
        self.rope_scaling = None
        self.max_position_embeddings = max_position_embeddings
        self.hidden_act = "silu"
        self.attention_bias = False
        self.mlp_bias = False
        self.tie_word_embeddings = False
        self.pad_token_id = 0
        self.bos_token_id = 1
        self.eos_token_id = 2

    def to_dict(self):
        return {k: v for k, v in vars(self).items()}


def _quantize_int(w_kn: torch.Tensor, groupsize: int):
    """Asymmetric 4-bit min/max quantisation of W[K,N] per (group of consecutive rows, column).
    Returns (intw uint8 [K,N], zero uint8 [G,N] in 1..16, scale fp16-rounded float [G,N])."""
    K, N = w_kn.shape
    G = K // groupsize
    w = w_kn.float().view(G, groupsize, N)
    wmin = w.min(dim=1).values.clamp(max=0)
    wmax = w.max(dim=1).values.clamp(min=0)
    scale = ((wmax - wmin) / 15.0).clamp(min=1e-8).half().float()  # scales are stored in fp16
    zero = torch.round(-wmin / scale).clamp(1, 16)  # the stored (zero - 1) must fit a nibble
    q = torch.clamp(torch.round(w / scale[:, None, :]) + zero[:, None, :], 0, 15)
    return q.view(K, N).to(torch.uint8).numpy(), zero.to(torch.uint8).numpy(), scale


def quantize_gptq(w_kn: torch.Tensor, groupsize: int, perm: Optional[np.ndarray] = None):
    """GPTQ tensors whose dequantisation (ops_ref.gptq_dequant) is the fake-quantised W.  With `perm` the groups
    are formed in that row order (act-order): g_idx[perm[j]] = j // groupsize."""
    K, N = w_kn.shape
    if perm is None:
        intw, zero, scale = _quantize_int(w_kn, groupsize)
        g_idx = (np.arange(K) // groupsize).astype(np.int32)
    else:
        intw_p, zero, scale = _quantize_int(w_kn[torch.from_numpy(perm)], groupsize)
        intw = np.empty_like(intw_p)
        intw[perm] = intw_p
        g_idx = np.empty(K, dtype=np.int32)
        g_idx[perm] = (np.arange(K) // groupsize).astype(np.int32)
    qweight, qzeros = ops_ref.gptq_pack(intw, zero)
    return qweight, qzeros, scale.half().numpy(), g_idx


def tiny_llama_tensors(cfg, seed: int, quantize: Optional[str] = None, groupsize: int = 64,
                       dtype=torch.float16, head_scale: float = 40.0, act_order: bool = False) -> Dict[str, torch.Tensor]:
    """name -> tensor in HF Llama naming.  Dense weights are rounded to `dtype` (so that the fp32 oracle and the
    fp16 kernels see the same values); GPTQ tensors replace the seven projection matrices of every layer."""
    g = torch.Generator().manual_seed(seed)
    E, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    D = E // cfg.num_attention_heads
    Hkv = cfg.num_key_value_heads
    t: Dict[str, torch.Tensor] = {}

    perms = {}

    def lin(name, n, k, std):
        w = torch.randn(n, k, generator=g) * std  # torch Linear layout [N, K]
        if quantize == "gptq":
            # act-order: one row order per (layer, input width) so that fused q/k/v and gate/up share g_idx,
            # as the reference requires (utils/weights.py:124-127)
            perm = perms.get(k) if act_order else None
            qw, qz, sc, gi = quantize_gptq(w.t().contiguous(), groupsize, perm)
            t[f"{name}.qweight"] = torch.from_numpy(qw)
            t[f"{name}.qzeros"] = torch.from_numpy(qz)
            t[f"{name}.scales"] = torch.from_numpy(sc)
            t[f"{name}.g_idx"] = torch.from_numpy(gi)
        else:
            t[f"{name}.weight"] = w.to(dtype)

    t["model.embed_tokens.weight"] = torch.randn(V, E, generator=g).to(dtype)
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}"
        perms = {E: torch.randperm(E, generator=g).numpy(), I: torch.randperm(I, generator=g).numpy()}
        lin(f"{p}.self_attn.q_proj", E, E, E ** -0.5)
        lin(f"{p}.self_attn.k_proj", Hkv * D, E, E ** -0.5)
        lin(f"{p}.self_attn.v_proj", Hkv * D, E, E ** -0.5)
        lin(f"{p}.self_attn.o_proj", E, E, E ** -0.5)
        lin(f"{p}.mlp.gate_proj", I, E, E ** -0.5)
        lin(f"{p}.mlp.up_proj", I, E, E ** -0.5)
        lin(f"{p}.mlp.down_proj", E, I, I ** -0.5)
        t[f"{p}.input_layernorm.weight"] = (1.0 + 0.1 * torch.randn(E, generator=g)).to(dtype)
        t[f"{p}.post_attention_layernorm.weight"] = (1.0 + 0.1 * torch.randn(E, generator=g)).to(dtype)
    t["model.norm.weight"] = (1.0 + 0.1 * torch.randn(E, generator=g)).to(dtype)
    t["lm_head.weight"] = (torch.randn(V, E, generator=g) * (E ** -0.5) * head_scale).to(dtype)
    return t


def dense_state_dict(cfg, tensors: Dict[str, torch.Tensor], groupsize: int = 64) -> Dict[str, torch.Tensor]:
    """HF LlamaForCausalLM state dict in fp32 (GPTQ matrices dequantised with the reference formula): what the
    reference's CPU path runs, since it refuses quantisation on CPU (server.py:290-291)."""
    sd = {}
    for name, v in tensors.items():
        if name.endswith(".qweight"):
            base = name[:-len(".qweight")]
            w = ops_ref.gptq_dequant(v.numpy(), tensors[f"{base}.qzeros"].numpy(), tensors[f"{base}.scales"],
                                     tensors[f"{base}.g_idx"].numpy(), groupsize)
            sd[f"{base}.weight"] = w.t().contiguous()  # [N, K]
        elif name.endswith((".qzeros", ".scales", ".g_idx")):
            continue
        else:
            sd[name] = v.float()
    return sd


class TinyBigCodeConfig:
    model_type = "gpt_bigcode"

    def __init__(self, vocab_size=256, hidden_size=256, n_inner=1024, num_hidden_layers=2, num_attention_heads=4,
                 layer_norm_epsilon=1e-5, n_positions=512, activation_function="gelu_pytorch_tanh"):
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.n_inner = n_inner
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.layer_norm_epsilon = layer_norm_epsilon
        self.n_positions = n_positions
        self.activation_function = activation_function
        self.multi_query = True
        self.architectures = ["GPTBigCodeForCausalLM"]
        self.transpose = False
        self.pad_token_id = 0
        self.bos_token_id = 1
        self.eos_token_id = 2
        self.tie_word_embeddings = True

    def to_dict(self):
        return {k: v for k, v in vars(self).items()}


def tiny_bigcode_tensors(cfg, seed: int, dtype=torch.float16, embed_scale: float = 6.0) -> Dict[str, torch.Tensor]:
    """HF GPTBigCode naming.  The head is tied to wte, so the embedding table is scaled up instead of the head to get
    decisive greedy margins."""
    g = torch.Generator().manual_seed(seed)
    E, I, V = cfg.hidden_size, cfg.n_inner, cfg.vocab_size
    D = E // cfg.num_attention_heads
    t: Dict[str, torch.Tensor] = {}

    def lin(name, n, k):
        t[f"{name}.weight"] = (torch.randn(n, k, generator=g) * k ** -0.5).to(dtype)
        t[f"{name}.bias"] = (torch.randn(n, generator=g) * 0.1).to(dtype)

    def ln(name):
        t[f"{name}.weight"] = (1.0 + 0.1 * torch.randn(E, generator=g)).to(dtype)
        t[f"{name}.bias"] = (0.1 * torch.randn(E, generator=g)).to(dtype)

    t["transformer.wte.weight"] = (torch.randn(V, E, generator=g) * embed_scale * E ** -0.5).to(dtype)
    t["transformer.wpe.weight"] = (torch.randn(cfg.n_positions, E, generator=g) * 0.3).to(dtype)
    for i in range(cfg.num_hidden_layers):
        p = f"transformer.h.{i}"
        ln(f"{p}.ln_1")
        lin(f"{p}.attn.c_attn", E + 2 * D, E)
        lin(f"{p}.attn.c_proj", E, E)
        ln(f"{p}.ln_2")
        lin(f"{p}.mlp.c_fc", I, E)
        lin(f"{p}.mlp.c_proj", E, I)
    ln("transformer.ln_f")
    return t


class TinyGPT2Config:
    """GPT-2 (multi-head attention, learned positions, LayerNorm, gelu_new, tied head): BASELINE config 1's family."""
    model_type = "gpt2"

    def __init__(self, vocab_size=256, n_embd=64, n_layer=2, n_head=4, n_positions=128, layer_norm_epsilon=1e-5,
                 activation_function="gelu_new"):
        self.vocab_size = vocab_size
        self.n_embd = self.hidden_size = n_embd
        self.n_layer = self.num_hidden_layers = n_layer
        self.n_head = self.num_attention_heads = n_head
        self.n_inner = 4 * n_embd
        self.n_positions = n_positions
        self.layer_norm_epsilon = layer_norm_epsilon
        self.activation_function = activation_function
        self.pad_token_id = 0
        self.bos_token_id = 1
        self.eos_token_id = 2

    def to_dict(self):
        return {k: v for k, v in vars(self).items()}


def tiny_gpt2_tensors(cfg, seed: int, embed_scale: float = 6.0) -> Dict[str, torch.Tensor]:
    """fp32 state dict in HF GPT2LMHeadModel naming.  Linear weights are stored [in, out] (HF's Conv1D).  The head is
    tied to wte, so the embedding table is scaled up to get decisive greedy margins."""
    g = torch.Generator().manual_seed(seed)
    E, I, V = cfg.n_embd, cfg.n_inner, cfg.vocab_size
    t: Dict[str, torch.Tensor] = {}

    def conv1d(name, k, n):
        t[f"{name}.weight"] = torch.randn(k, n, generator=g) * k ** -0.5
        t[f"{name}.bias"] = torch.randn(n, generator=g) * 0.1

    def ln(name):
        t[f"{name}.weight"] = 1.0 + 0.1 * torch.randn(E, generator=g)
        t[f"{name}.bias"] = 0.1 * torch.randn(E, generator=g)

    t["transformer.wte.weight"] = torch.randn(V, E, generator=g) * embed_scale * E ** -0.5
    t["transformer.wpe.weight"] = torch.randn(cfg.n_positions, E, generator=g) * 0.3
    for i in range(cfg.n_layer):
        p = f"transformer.h.{i}"
        ln(f"{p}.ln_1")
        conv1d(f"{p}.attn.c_attn", E, 3 * E)
        conv1d(f"{p}.attn.c_proj", E, E)
        ln(f"{p}.ln_2")
        conv1d(f"{p}.mlp.c_fc", E, I)
        conv1d(f"{p}.mlp.c_proj", I, E)
    ln("transformer.ln_f")
    return t
