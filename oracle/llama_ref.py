"""CPU fp32 restatement of the Llama forward that FlashCausalLM.generate_token drives, plus the greedy
generate loop around it.  TEST INFRASTRUCTURE ONLY (see oracle/ops_ref.py header).

Follows custom_modeling/flash_llama_modeling.py of the reference line by line:
  FlashLlamaModel.forward (:425-497): embed -> per layer [fused add+RMSNorm (:368), fused qkv matmul and split
  (:251-259), RoPE on q and k (:262-263), KV append (:268,282), attention over all cached slots incl. the new
  one (:271-295), o_proj (:297), fused add+RMSNorm (:383-385), gate_up -> silu*mul -> down (:332-335)] ->
  final norm (:495) -> lm_head (:539).
The same arithmetic is what the reference's CPU `causal_lm` path computes through HF transformers'
LlamaForCausalLM (models/causal_lm.py:604-634); tests/golden/*.npz hold logits and token ids captured from
that path in this container (tests/golden/make_fixtures.py) and tests/test_oracle_golden.py pins this file
against them.  Everything is fp32: cos/sin are NOT rounded to the model dtype here (the CPU path is fp32).
"""
from typing import Dict, List, Optional

import numpy as np
import torch

from oracle import ops_ref


class LlamaRef:
    def __init__(self, cfg, tensors: Dict[str, torch.Tensor], quantize: Optional[str] = None, groupsize: int = 128,
                 layers: Optional[int] = None):
        self.cfg = cfg
        self.E = cfg.hidden_size
        self.H = cfg.num_attention_heads
        self.Hkv = getattr(cfg, "num_key_value_heads", None) or self.H
        self.D = self.E // self.H
        self.I = cfg.intermediate_size
        self.L = layers if layers is not None else cfg.num_hidden_layers
        self.eps = cfg.rms_norm_eps
        self.theta = getattr(cfg, "rope_theta", 10000.0)
        rs = getattr(cfg, "rope_scaling", None)
        self.rope_factor = float(rs.get("factor", 1.0)) if rs and rs.get("type") == "linear" else 1.0
        self.quantize = quantize
        self.groupsize = groupsize
        self.t = tensors
        self._w = {}

    # weight matrices as fp32 [K, N] (x @ W)
    def _lin(self, name: str) -> torch.Tensor:
        w = self._w.get(name)
        if w is None:
            if self.quantize == "gptq" and f"{name}.qweight" in self.t:
                g = self.t.get(f"{name}.g_idx")
                w = ops_ref.gptq_dequant(self.t[f"{name}.qweight"].cpu().numpy(), self.t[f"{name}.qzeros"].cpu().numpy(),
                                         self.t[f"{name}.scales"].float().cpu(), None if g is None else g.cpu().numpy(),
                                         self.groupsize)
            else:
                w = self.t[f"{name}.weight"].float().cpu().t().contiguous()
            self._w[name] = w
        return w

    def _vec(self, name: str) -> torch.Tensor:
        return self.t[name].float().cpu()

    def drop_cache(self):
        self._w = {}

    def new_state(self, batch: int):
        """Per sequence, per layer: K [t,Hkv,D] and V [t,Hkv,D] (rotated keys), grown by forward()."""
        return [[None] * self.L for _ in range(batch)]

    @staticmethod
    def _last_rows(seq_of_token: List[int]) -> List[int]:
        """Index of the last token of every sequence (sequences are contiguous runs)."""
        return [i for i in range(len(seq_of_token)) if i + 1 == len(seq_of_token) or seq_of_token[i + 1] != seq_of_token[i]]

    def forward(self, input_ids: torch.Tensor, position_ids: torch.Tensor, seq_of_token: List[int], state,
                hidden_in: Optional[torch.Tensor] = None, return_hidden: bool = False,
                last_only: bool = False) -> torch.Tensor:
        """One forward over T tokens (any mix of prefill runs and decode tokens): token i belongs to sequence
        seq_of_token[i] at position position_ids[i]; tokens of one sequence are contiguous and ascending.
        Returns fp32 logits [T, V] and appends this forward's K/V to `state`.
        last_only=True returns the logits of each sequence's last token only ([B, V], what generate_token consumes:
        flash_causal_lm.py:515-521 `out[cu_seqlens[1:] - 1]`) and skips work nothing depends on: in the LAST layer the
        queries, attention output, o_proj and MLP of the other tokens feed nothing (their K/V are still computed
        and cached).  Same arithmetic for every value that is returned or cached; it only makes prompts of
        workload length affordable on CPU."""
        T = len(seq_of_token)
        x = hidden_in if hidden_in is not None else self._vec("model.embed_tokens.weight")[input_ids.long()]
        max_pos = int(position_ids.max()) + 1
        cos, sin = ops_ref.rope_tables(self.D, self.theta, max_pos, torch.float32, self.rope_factor)
        cos, sin = cos[position_ids.long()], sin[position_ids.long()]
        seqs = sorted(set(seq_of_token), key=seq_of_token.index)
        tok_idx = {b: [i for i, s in enumerate(seq_of_token) if s == b] for b in seqs}
        residual = None
        keep = None
        for l in range(self.L):
            if last_only and l == self.L - 1:
                keep = self._last_rows(seq_of_token)
            x, residual = self._layer(l, x, residual, cos, sin, seq_of_token, state, keep)
        if last_only and keep is None:  # no layers at all
            keep = self._last_rows(seq_of_token)
            x = x[keep]
        if return_hidden:
            return x if residual is None else x + residual
        hfin, _ = ops_ref.rmsnorm_residual(x, residual, self._vec("model.norm.weight"), self.eps)
        return hfin @ self._lin("lm_head")

    def _layer(self, l: int, x, residual, cos, sin, seq_of_token: List[int], state, keep):
        """Decoder layer l over the tokens of one forward (flash_llama_modeling.py:360-393); `keep` (last layer of a
        last_only forward) = the rows whose queries / attention / MLP are computed at all.  Appends K/V to state[b][l]."""
        T = len(seq_of_token)
        seqs = sorted(set(seq_of_token), key=seq_of_token.index)
        tok_idx = {b: [i for i, s in enumerate(seq_of_token) if s == b] for b in seqs}
        p = f"model.layers.{l}"
        h, residual = ops_ref.rmsnorm_residual(x, residual, self._vec(f"{p}.input_layernorm.weight"), self.eps)
        k = (h @ self._lin(f"{p}.self_attn.k_proj")).view(T, self.Hkv, self.D)
        v = (h @ self._lin(f"{p}.self_attn.v_proj")).view(T, self.Hkv, self.D)
        k = ops_ref.apply_rope(k, cos, sin)
        rows = keep if keep is not None else list(range(T))
        hq = h[rows] if keep is not None else h
        q = (hq @ self._lin(f"{p}.self_attn.q_proj")).view(len(rows), self.H, self.D)
        q = ops_ref.apply_rope(q, cos[rows], sin[rows])
        attn = torch.empty((len(rows), self.H, self.D), dtype=torch.float32)
        row_of = {r: j for j, r in enumerate(rows)}
        for b in seqs:
            idx = tok_idx[b]
            past = state[b][l]
            kb = k[idx] if past is None else torch.cat([past[0], k[idx]])
            vb = v[idx] if past is None else torch.cat([past[1], v[idx]])
            state[b][l] = (kb, vb)
            qi = [row_of[i] for i in idx if i in row_of]  # the LAST len(qi) tokens of this run
            attn[qi] = ops_ref.attention_varlen(q[qi], kb, vb, [0, len(qi)], [0, kb.shape[0]], self.D ** -0.5)
        if keep is not None:
            residual = residual[rows]
        o = attn.reshape(len(rows), self.H * self.D) @ self._lin(f"{p}.self_attn.o_proj")
        h2, residual = ops_ref.rmsnorm_residual(o, residual, self._vec(f"{p}.post_attention_layernorm.weight"),
                                                self.eps)
        gate = h2 @ self._lin(f"{p}.mlp.gate_proj")
        up = h2 @ self._lin(f"{p}.mlp.up_proj")
        return (torch.nn.functional.silu(gate) * up) @ self._lin(f"{p}.mlp.down_proj"), residual

    def generate_forced_layer_major(self, prompts: List[List[int]], forced: List[List[int]], keep_cache_layers=(0,)):
        """generate_greedy(prompts, len(forced), forced) evaluated LAYER by layer instead of step by step: with every fed
        token known in advance (teacher forcing) the prefill and all decode forwards go through layer l before anyone
        touches layer l + 1, so each layer's weights are dequantised once and dropped — a 32-layer int4 model at width
        4096 then needs one layer (0.8 GB fp32) on the host at a time.  Same functions, same order of arithmetic per
        value; K/V are kept only for `keep_cache_layers`.  Returns the same per-step dicts."""
        B = len(prompts)
        n_steps = len(forced)
        state = self.new_state(B)
        self.last_state = state
        emb = self._vec("model.embed_tokens.weight")
        lengths = [len(p) for p in prompts]
        passes = [(torch.tensor([t for p in prompts for t in p], dtype=torch.int64),
                   torch.tensor([i for p in prompts for i in range(len(p))], dtype=torch.int64),
                   [b for b, p in enumerate(prompts) for _ in p])]
        for step in range(n_steps - 1):
            passes.append((torch.tensor(forced[step], dtype=torch.int64),
                           torch.tensor([n + step for n in lengths], dtype=torch.int64), list(range(B))))
        xs = [emb[ids.long()] for ids, _, _ in passes]
        rs = [None] * len(passes)
        ropes = []
        for _, pos, _ in passes:
            cos, sin = ops_ref.rope_tables(self.D, self.theta, int(pos.max()) + 1, torch.float32, self.rope_factor)
            ropes.append((cos[pos.long()], sin[pos.long()]))
        for l in range(self.L):
            for i, (_, _, seq) in enumerate(passes):
                keep = self._last_rows(seq) if (i == 0 and l == self.L - 1) else None  # prefill: last_only
                xs[i], rs[i] = self._layer(l, xs[i], rs[i], ropes[i][0], ropes[i][1], seq, state, keep)
            self._w = {k: v for k, v in self._w.items() if not k.startswith(f"model.layers.{l}.")}
            if l not in keep_cache_layers:
                for b in range(B):
                    state[b][l] = None
        if self.L == 0:
            xs[0] = xs[0][self._last_rows(passes[0][2])]
        cu = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
        cu_q = np.arange(B + 1, dtype=np.int64)
        steps = []
        for i in range(n_steps):
            hfin, _ = ops_ref.rmsnorm_residual(xs[i], rs[i], self._vec("model.norm.weight"), self.eps)
            logits = hfin @ self._lin("lm_head")
            tok, lp = ops_ref.greedy(logits)
            cu = cu + cu_q
            steps.append({"logits": logits, "token_ids": tok.clone(), "logprobs": lp,
                          "slot_indices": torch.from_numpy(cu[1:] - 1)})
        return steps

    def generate_greedy(self, prompts: List[List[int]], new_tokens: int, forced: Optional[List[List[int]]] = None):
        """Prefill + greedy decode, the loop of FlashCausalLM.generate_token / CausalLM.generate_token
        (models/flash_causal_lm.py:405-460).  Returns per step: logits [B,V], token ids [B], logprobs [B], and
        the logical new-token slot indices `cu_seqlens[1:] - 1` the reference would use
        (flash_llama_modeling.py:465; bookkeeping flash_causal_lm.py:439-458).
        `forced[step][b]` teacher-forces the token fed back (the oracle's own argmax otherwise)."""
        B = len(prompts)
        state = self.new_state(B)
        self.last_state = state  # per sequence, per layer (K, V): tests read the cache contents back against it
        ids = torch.tensor([t for p in prompts for t in p], dtype=torch.int64)
        pos = torch.tensor([i for p in prompts for i in range(len(p))], dtype=torch.int64)
        seq = [b for b, p in enumerate(prompts) for _ in p]
        logits = self.forward(ids, pos, seq, state, last_only=True)
        lengths = [len(p) for p in prompts]
        cu = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
        cu_q = np.arange(B + 1, dtype=np.int64)
        steps = []
        for step in range(new_tokens):
            tok, lp = ops_ref.greedy(logits)
            cu = cu + cu_q  # one more (free) slot per sequence after every forward
            steps.append({"logits": logits, "token_ids": tok.clone(), "logprobs": lp,
                          "slot_indices": torch.from_numpy(cu[1:] - 1)})
            if step + 1 == new_tokens:
                break
            feed = tok if forced is None else torch.tensor(forced[step], dtype=torch.int64)
            pos = torch.tensor(lengths, dtype=torch.int64)
            logits = self.forward(feed, pos, list(range(B)), state)
            lengths = [l + 1 for l in lengths]
        return steps
