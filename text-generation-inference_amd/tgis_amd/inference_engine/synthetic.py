"""Engine for seeded synthetic checkpoints held in memory (benchmarks, smoke and parity tests): same model
classes, same `Weights` interface and TP slicing as tgis_native, no files and no tokenizer download.
The weight recipe is the one SURVEY.md §8(d) defines (N(0, 0.02) dense; GPTQ: random nibbles, random zero nibbles in 0..13,
scales U(0.5,1.5)*2/15*0.02, g_idx = k // group) — there is no network for real checkpoints."""
import os
from typing import Any, Dict, Optional

import torch

from tgis_amd.inference_engine.engine import BaseInferenceEngine
from tgis_amd.inference_engine.tgis_native import _barrier, model_class_for
from tgis_amd.utils.dist import initialize_torch_distributed
from tgis_amd.utils.weights import DictWeights


class InferenceEngine(BaseInferenceEngine):
    def __init__(self, tensors: Dict[str, torch.Tensor], model_config: Any, dtype: torch.dtype,
                 quantize: Optional[str], tokenizer=None, gptq_bits: int = 4, gptq_groupsize: int = 128):
        super().__init__(None, model_config, tokenizer=tokenizer)
        model_class, _aliases = model_class_for(self._config)
        self._config.quantize = quantize
        self.process_group = initialize_torch_distributed(self.world_size, self.rank)
        _barrier(self.process_group)
        weights = DictWeights(tensors, device=self.device, dtype=dtype, process_group=self.process_group)
        if quantize == "gptq":
            weights.gptq_bits, weights.gptq_groupsize = gptq_bits, gptq_groupsize
        model = model_class(self._config, weights)
        _barrier(self.process_group)
        if not hasattr(model, "config"):
            model.config = self._config
        self.model = model


def llama_tensors(config, quantize: Optional[str], seed: int, groupsize: int = 128, device="cpu",
                  dtype=torch.float16, head_scale: float = 1.0) -> Dict[str, torch.Tensor]:
    """Seeded full (unsharded) Llama checkpoint as a name -> tensor dict in HF naming."""
    g = torch.Generator(device=device).manual_seed(seed)
    E, I, V = config.hidden_size, config.intermediate_size, config.vocab_size
    D = E // config.num_attention_heads
    Hkv = config.num_key_value_heads
    t: Dict[str, torch.Tensor] = {}

    def dense(name, n, k, std=0.02):
        t[f"{name}.weight"] = (torch.randn(n, k, generator=g, device=device) * std).to(dtype)

    def gptq(name, n, k):
        G = k // groupsize
        t[f"{name}.qweight"] = torch.randint(-2**31, 2**31 - 1, (k // 8, n), generator=g, device=device,
                                             dtype=torch.int32)
        # stored zero nibbles 0..13 (true zero points 1..14, mean 7.5 = the mean nibble): with 0..15 the dequantised
        # weights carry a common -1 * scale offset, a rank-one term that overflows fp16 activations within a few layers
        zn = torch.randint(0, 14, (G, n // 8, 8), generator=g, device=device, dtype=torch.int64)
        zw = (zn << (4 * torch.arange(8, device=device, dtype=torch.int64))).sum(-1)
        t[f"{name}.qzeros"] = torch.where(zw >= 2**31, zw - 2**32, zw).to(torch.int32)
        t[f"{name}.scales"] = ((torch.rand(G, n, generator=g, device=device) + 0.5) * (2.0 / 15.0) * 0.02
                               ).to(torch.float16)
        t[f"{name}.g_idx"] = (torch.arange(k, device=device, dtype=torch.int32) // groupsize)

    lin = gptq if quantize == "gptq" else dense
    t["model.embed_tokens.weight"] = (torch.randn(V, E, generator=g, device=device) * 0.02).to(dtype)
    for i in range(config.num_hidden_layers):
        p = f"model.layers.{i}"
        lin(f"{p}.self_attn.q_proj", E, E)
        lin(f"{p}.self_attn.k_proj", Hkv * D, E)
        lin(f"{p}.self_attn.v_proj", Hkv * D, E)
        lin(f"{p}.self_attn.o_proj", E, E)
        lin(f"{p}.mlp.gate_proj", I, E)
        lin(f"{p}.mlp.up_proj", I, E)
        lin(f"{p}.mlp.down_proj", E, I)
        t[f"{p}.input_layernorm.weight"] = torch.ones(E, device=device, dtype=dtype)
        t[f"{p}.post_attention_layernorm.weight"] = torch.ones(E, device=device, dtype=dtype)
    t["model.norm.weight"] = torch.ones(E, device=device, dtype=dtype)
    t["lm_head.weight"] = (torch.randn(V, E, generator=g, device=device) * 0.02 * head_scale).to(dtype)
    return t


class BigCodeConfig:
    """The fields of HF's GPTBigCodeConfig that FlashSantacoderForCausalLM reads (multi-query attention)."""
    model_type = "gpt_bigcode"

    def __init__(self, vocab_size=49152, hidden_size=6144, n_inner=24576, num_hidden_layers=40, num_attention_heads=48,
                 layer_norm_epsilon=1e-5, n_positions=8192, activation_function="gelu_pytorch_tanh"):
        self.vocab_size, self.hidden_size, self.n_inner = vocab_size, hidden_size, n_inner
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.layer_norm_epsilon, self.n_positions = layer_norm_epsilon, n_positions
        self.activation_function = activation_function
        self.multi_query = True
        self.architectures = ["GPTBigCodeForCausalLM"]
        self.transpose = False
        self.pad_token_id, self.bos_token_id, self.eos_token_id = 0, 0, 0
        self.tie_word_embeddings = True
        # the names the benchmark's byte accounting shares with Llama
        self.intermediate_size = n_inner
        self.num_key_value_heads = 1


def bigcode_tensors(config, seed: int, device="cpu", dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    """Seeded full GPT-BigCode checkpoint in HF naming (tied head): N(0, 1/sqrt(fan_in)) weights, small biases."""
    g = torch.Generator(device=device).manual_seed(seed)
    E, I, V = config.hidden_size, config.n_inner, config.vocab_size
    D = E // config.num_attention_heads
    t: Dict[str, torch.Tensor] = {}

    def lin(name, n, k):
        t[f"{name}.weight"] = (torch.randn(n, k, generator=g, device=device) * k ** -0.5).to(dtype)
        t[f"{name}.bias"] = (torch.randn(n, generator=g, device=device) * 0.02).to(dtype)

    def ln(name):
        t[f"{name}.weight"] = torch.ones(E, device=device, dtype=dtype)
        t[f"{name}.bias"] = torch.zeros(E, device=device, dtype=dtype)

    t["transformer.wte.weight"] = (torch.randn(V, E, generator=g, device=device) * 0.02).to(dtype)
    t["transformer.wpe.weight"] = (torch.randn(config.n_positions, E, generator=g, device=device) * 0.02).to(dtype)
    for i in range(config.num_hidden_layers):
        p = f"transformer.h.{i}"
        ln(f"{p}.ln_1")
        lin(f"{p}.attn.c_attn", E + 2 * D, E)
        lin(f"{p}.attn.c_proj", E, E)
        ln(f"{p}.ln_2")
        lin(f"{p}.mlp.c_fc", I, E)
        lin(f"{p}.mlp.c_proj", E, I)
    ln("transformer.ln_f")
    return t
