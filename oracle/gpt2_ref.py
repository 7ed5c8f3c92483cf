"""CPU fp32 restatement of the GPT-2 forward behind the reference's padded `causal_lm` path (BASELINE config 1).
TEST INFRASTRUCTURE ONLY.

The reference computes this model through HF `GPT2LMHeadModel` (models/causal_lm.py:604-634 ->
transformers/models/gpt2/modeling_gpt2.py): wte(ids) + wpe(position_ids) -> per block [ln_1, c_attn -> q, k, v
(all n_head heads), causal attention over the cached keys, c_proj, residual add, ln_2, c_fc -> gelu_new -> c_proj,
residual add] -> ln_f -> lm_head tied to wte.  Position ids of a left-padded row count only attended tokens
(`cumsum(attention_mask) - 1`, causal_lm.py:193-199), which is exactly "token i of the sequence sits at position i" in
the var-len formulation used here.  tests/golden/gpt2_*.npz hold logits, token ids and batch tensors captured from
the reference in this container; tests/test_oracle_golden.py pins this file against them."""
from typing import Dict, List

import torch

from oracle import ops_ref
from oracle.llama_ref import LlamaRef


class GPT2Ref(LlamaRef):
    def __init__(self, cfg, tensors: Dict[str, torch.Tensor]):
        self.cfg = cfg
        self.E = cfg.n_embd
        self.H = self.Hkv = cfg.n_head
        self.D = self.E // self.H
        self.L = cfg.n_layer
        self.eps = cfg.layer_norm_epsilon
        assert cfg.activation_function in ("gelu_new", "gelu_pytorch_tanh", "gelu_fast")  # all the tanh form
        self.quantize = None
        self.t = tensors
        self._w = {}

    def _mat(self, name: str) -> torch.Tensor:  # Conv1D weights are already [in, out]
        return self.t[f"{name}.weight"].float().cpu()

    def forward(self, input_ids, position_ids, seq_of_token: List[int], state, hidden_in=None, return_hidden=False,
                last_only=False):
        T = len(seq_of_token)
        x = self._vec("transformer.wte.weight")[input_ids.long()] + self._vec("transformer.wpe.weight")[position_ids.long()]
        seqs = sorted(set(seq_of_token), key=seq_of_token.index)
        tok_idx = {b: [i for i, s in enumerate(seq_of_token) if s == b] for b in seqs}
        H, D, E = self.H, self.D, self.E
        for l in range(self.L):
            p = f"transformer.h.{l}"
            h, _ = ops_ref.layernorm_residual(x, None, self._vec(f"{p}.ln_1.weight"), self._vec(f"{p}.ln_1.bias"), self.eps)
            qkv = h @ self._mat(f"{p}.attn.c_attn") + self._vec(f"{p}.attn.c_attn.bias")
            q, k, v = (qkv[:, j * E:(j + 1) * E].reshape(T, H, D) for j in range(3))
            attn = torch.empty((T, H, D), dtype=torch.float32)
            for b in seqs:
                idx = tok_idx[b]
                past = state[b][l]
                kb = k[idx] if past is None else torch.cat([past[0], k[idx]])
                vb = v[idx] if past is None else torch.cat([past[1], v[idx]])
                state[b][l] = (kb, vb)
                attn[idx] = ops_ref.attention_varlen(q[idx], kb, vb, [0, len(idx)], [0, kb.shape[0]], D ** -0.5)
            x = x + attn.reshape(T, E) @ self._mat(f"{p}.attn.c_proj") + self._vec(f"{p}.attn.c_proj.bias")
            h2, _ = ops_ref.layernorm_residual(x, None, self._vec(f"{p}.ln_2.weight"), self._vec(f"{p}.ln_2.bias"), self.eps)
            f = ops_ref.gelu(h2 @ self._mat(f"{p}.mlp.c_fc") + self._vec(f"{p}.mlp.c_fc.bias"), True)
            x = x + f @ self._mat(f"{p}.mlp.c_proj") + self._vec(f"{p}.mlp.c_proj.bias")
        if last_only:
            x = x[self._last_rows(seq_of_token)]
        hf, _ = ops_ref.layernorm_residual(x, None, self._vec("transformer.ln_f.weight"), self._vec("transformer.ln_f.bias"),
                                           self.eps)
        return hf @ self._vec("transformer.wte.weight").t()
