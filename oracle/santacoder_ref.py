"""CPU fp32 restatement of the GPT-BigCode (multi-query) forward.  TEST INFRASTRUCTURE ONLY.

Follows custom_modeling/flash_santacoder_modeling.py of the reference: FlashSantacoderModel.forward (:392-459)
wte(ids) + wpe(pos) -> per block [fused add+LayerNorm ln_1 (:342), c_attn -> q [T,H,D] and one shared k,v head
(:232-241), attention over all cached slots (:243-276), c_proj (:278), fused add+LayerNorm ln_2 (:352), c_fc -> gelu
-> c_proj (:303-307)] -> ln_f -> lm_head tied to wte (:466-468).  The reference's CPU path runs the same arithmetic
through HF GPTBigCodeForCausalLM; tests/golden/bigcode_*.npz pin this file against it."""
from typing import Dict, List

import torch

from oracle import ops_ref
from oracle.llama_ref import LlamaRef


class SantacoderRef(LlamaRef):
    def __init__(self, cfg, tensors: Dict[str, torch.Tensor]):
        self.cfg = cfg
        self.E = cfg.hidden_size
        self.H = cfg.num_attention_heads
        self.Hkv = 1
        self.D = self.E // self.H
        self.L = cfg.num_hidden_layers
        self.eps = cfg.layer_norm_epsilon
        self.tanh = cfg.activation_function in ("gelu_fast", "gelu_pytorch_tanh")
        self.quantize = None
        self.t = tensors
        self._w = {}

    def forward(self, input_ids, position_ids, seq_of_token: List[int], state, hidden_in=None, return_hidden=False,
                last_only=False):
        """last_only: see LlamaRef.forward."""
        T = len(seq_of_token)
        x = self._vec("transformer.wte.weight")[input_ids.long()] + self._vec("transformer.wpe.weight")[position_ids.long()]
        seqs = sorted(set(seq_of_token), key=seq_of_token.index)
        tok_idx = {b: [i for i, s in enumerate(seq_of_token) if s == b] for b in seqs}
        residual = None
        H, D = self.H, self.D
        keep = None
        for l in range(self.L):
            p = f"transformer.h.{l}"
            h, residual = ops_ref.layernorm_residual(x, residual, self._vec(f"{p}.ln_1.weight"),
                                                     self._vec(f"{p}.ln_1.bias"), self.eps)
            w_attn, b_attn = self._lin(f"{p}.attn.c_attn"), self._vec(f"{p}.attn.c_attn.bias")
            kv = h @ w_attn[:, H * D:] + b_attn[H * D:]
            k = kv[:, :D].reshape(T, 1, D)
            v = kv[:, D:].reshape(T, 1, D)
            if last_only and l == self.L - 1:
                keep = self._last_rows(seq_of_token)
            rows = keep if keep is not None else list(range(T))
            hq = h[rows] if keep is not None else h
            q = (hq @ w_attn[:, :H * D] + b_attn[:H * D]).view(len(rows), H, D)
            attn = torch.empty((len(rows), H, D), dtype=torch.float32)
            row_of = {r: j for j, r in enumerate(rows)}
            for b in seqs:
                idx = tok_idx[b]
                past = state[b][l]
                kb = k[idx] if past is None else torch.cat([past[0], k[idx]])
                vb = v[idx] if past is None else torch.cat([past[1], v[idx]])
                state[b][l] = (kb, vb)
                qi = [row_of[i] for i in idx if i in row_of]
                attn[qi] = ops_ref.attention_varlen(q[qi], kb, vb, [0, len(qi)], [0, kb.shape[0]], D ** -0.5)
            if keep is not None:
                residual = residual[rows]
            o = attn.reshape(len(rows), H * D) @ self._lin(f"{p}.attn.c_proj") + self._vec(f"{p}.attn.c_proj.bias")
            h2, residual = ops_ref.layernorm_residual(o, residual, self._vec(f"{p}.ln_2.weight"),
                                                      self._vec(f"{p}.ln_2.bias"), self.eps)
            f = ops_ref.gelu(h2 @ self._lin(f"{p}.mlp.c_fc") + self._vec(f"{p}.mlp.c_fc.bias"), self.tanh)
            x = f @ self._lin(f"{p}.mlp.c_proj") + self._vec(f"{p}.mlp.c_proj.bias")
        if last_only and keep is None:
            keep = self._last_rows(seq_of_token)
            x = x[keep]
        hf, _ = ops_ref.layernorm_residual(x, residual, self._vec("transformer.ln_f.weight"),
                                           self._vec("transformer.ln_f.bias"), self.eps)
        return hf @ self._vec("transformer.wte.weight").t()
