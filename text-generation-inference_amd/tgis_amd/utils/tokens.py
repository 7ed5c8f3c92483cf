"""Next-token selection for a heterogeneous batch, and extraction of the per-token details the router asks for.

Mirrors utils/tokens.py of the reference: `Sampling` / `Greedy` (:32-46), `HeterogeneousNextTokenChooser`
(:161-333: per-row min-new-tokens EOS mask and length penalty :242-256, repetition penalty, warpers,
choice, log-softmax only when some request wants logprobs), `HeterogeneousSampling` (:336-385),
`get_token_info` (:388-425) and `get_input_tokens_info` (:429-506).

Differences: when the whole batch is plain greedy with no processors, `choose_greedy_fused` runs one HIP
kernel (argmax + logprob) and the caller does a single device->host copy instead of one `.item()` per request;
for any other batch whose logits are on the GPU, `choose_fused` runs the whole chain (EOS mask / length penalty,
repetition penalty, warpers, argmax or draw, log-softmax at the chosen id) as one launch of `tgis_warp_sample`.
The torch processors below remain the definition of the semantics (pinned to HF's per-row processors on CPU) and
the path for logits that live on the host."""
import os
from itertools import chain, repeat
from typing import List, Optional, Tuple, Union

import torch

from tgis_amd import native
from tgis_amd.pb import generate_pb2
from tgis_amd.utils.logits_process import (
    HeterogeneousRepetitionPenaltyLogitsProcessor,
    HeterogeneousTemperatureLogitsWarper,
    HeterogeneousTopKLogitsWarper,
    HeterogeneousTopPLogitsWarper,
    HeterogeneousTypicalLogitsWarper,
)
from tgis_amd.utils.token_types import InputTokens, TokenInfo, TopToken

NONES = repeat(None)

_MASK64 = 0xFFFFFFFFFFFFFFFF
_seed_base = int.from_bytes(os.urandom(8), "little")
_seed_count = 0


def set_seed_base(base: int) -> None:
    """Seeds of requests that bring none are `mix(base, n)` for the n-th such request of this process.  Tensor-parallel
    ranks call this with rank 0's base so that they draw identical tokens for the same request sequence."""
    global _seed_base, _seed_count
    _seed_base, _seed_count = base & _MASK64, 0


def seed_base() -> int:
    return _seed_base


def _fresh_seed() -> int:
    global _seed_count
    _seed_count += 1
    z = (_seed_base + 0x9E3779B97F4A7C15 * _seed_count) & _MASK64  # splitmix64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK64
    return z ^ (z >> 31)


class Sampling:
    """One request's random stream: (seed, number of draws so far).  It belongs to the request, so concatenate /
    prune keep every request's stream (tokens.py:32-41).  On the GPU the draw is made inside `tgis_warp_sample`
    (Philox keyed by `seed`, counter = `offset`); on host tensors, softmax(logits) / Exp(1) noise -> argmax with a
    torch generator seeded the same way — a multinomial draw without a sync either way."""

    def __init__(self, seed: Optional[int] = None, device="cpu"):
        self.seed = int(seed) if seed is not None else _fresh_seed()
        self.offset = 0
        self.device = device
        self._generator = None

    @property
    def generator(self) -> torch.Generator:
        if self._generator is None:
            self._generator = torch.Generator(self.device).manual_seed(self.seed)
        return self._generator

    @property
    def state(self) -> Tuple[int, int]:
        """(seed, offset) as two's-complement int64 values for the device-side [B, 2] state tensor."""
        seed = self.seed & 0xFFFFFFFFFFFFFFFF
        return (seed - (1 << 64) if seed >= (1 << 63) else seed), self.offset

    def __call__(self, logits: torch.Tensor) -> torch.Tensor:
        probs = torch.nn.functional.softmax(logits, -1)
        q = torch.empty_like(probs).exponential_(1, generator=self.generator)
        self.offset += 1
        return probs.div_(q).argmax()


class Greedy:
    def __call__(self, logits: torch.Tensor) -> torch.Tensor:
        return logits.argmax(dim=-1)


class HeterogeneousSampling:
    """Greedy rows and sampled rows in one batch: argmax for all, then overwrite the sampled rows."""

    def __init__(self, do_sample: List[bool], seeds: List[Optional[Union[int, Sampling]]], device):
        self.greedy_indices: List[int] = []
        self.sampling_mapping = {}
        self.samplings: List[Optional[Sampling]] = []
        for i, (sample, seed) in enumerate(zip(do_sample, seeds)):
            if not sample:
                self.greedy_indices.append(i)
                self.samplings.append(None)
                continue
            s = seed if isinstance(seed, Sampling) else Sampling(seed, device)
            self.sampling_mapping[i] = s
            self.samplings.append(s)

    def __call__(self, logits: torch.Tensor) -> torch.Tensor:
        out = torch.empty(logits.shape[0], dtype=torch.int64, device=logits.device)
        if self.greedy_indices:
            torch.argmax(logits, -1, out=out)
        for i, s in self.sampling_mapping.items():
            out[i] = s(logits[i])
        return out

    def filter(self, indices):
        mapping, greedy = {}, []
        for new_i, old_i in enumerate(indices):
            if old_i in self.sampling_mapping:
                mapping[new_i] = self.sampling_mapping[old_i]
            else:
                greedy.append(new_i)
        self.sampling_mapping, self.greedy_indices = mapping, greedy
        self.samplings = [self.samplings[i] for i in indices]
        return self


class HeterogeneousNextTokenChooser:
    def __init__(self, temperature: List[float], top_k: List[int], top_p: List[float], typical_p: List[float],
                 seeds: List[Optional[Union[int, Sampling]]], repetition_penalty: List[float],
                 length_penalty: List[Optional[Tuple[int, float]]], min_new_tokens: List[int],
                 return_logprobs: List[bool], eos_token_id: Optional[int] = None,
                 pad_token_id: Optional[int] = None, device=None, dtype=None,
                 current_tokens: Optional[List[int]] = None):
        self.repetition_processor = None
        if any(x != 1.0 for x in repetition_penalty):
            # the eos id is not penalised when it doubles as the pad id (it fills all_input_ids padding)
            self.repetition_processor = HeterogeneousRepetitionPenaltyLogitsProcessor(
                repetition_penalty, dtype, device,
                id_to_exclude=eos_token_id if eos_token_id == pad_token_id else None)
        do_sample = [t != 0.0 for t in temperature]
        warpers = []
        if any(do_sample):
            if any(t != 1.0 for t in temperature):
                warpers.append(HeterogeneousTemperatureLogitsWarper(
                    [t if t != 0 else 1 for t in temperature], dtype, device))
            if any(k != 0 for k in top_k):
                warpers.append(HeterogeneousTopKLogitsWarper(top_k, device))
            if any(p < 1.0 for p in top_p):
                warpers.append(HeterogeneousTopPLogitsWarper(top_p, dtype, device))
            if any(p < 1.0 for p in typical_p):
                warpers.append(HeterogeneousTypicalLogitsWarper(typical_p, dtype, device))
            self.choice = HeterogeneousSampling(do_sample, seeds, device)
        else:
            self.choice = Greedy()
        self.warpers = warpers
        self.eos_token_id = eos_token_id
        self.pad_token_id = pad_token_id
        self.length_penalty = length_penalty
        self.min_new_tokens = min_new_tokens
        self.current_tokens = current_tokens if current_tokens is not None else [0] * len(do_sample)
        self.do_sample = do_sample
        self.dtype = dtype
        self.device = device
        self.return_logprobs = return_logprobs
        self._fused = None  # device-side parameter arrays of choose_fused, rebuilt after filter()

    @property
    def samplings(self):
        if isinstance(self.choice, Greedy):
            return [None] * len(self.do_sample)
        return self.choice.samplings

    @property
    def is_plain_greedy(self) -> bool:
        """True when __call__ would be argmax (+ log-softmax) of the unmodified logits for every row."""
        return (isinstance(self.choice, Greedy) and self.repetition_processor is None and not self.warpers
                and all(lp is None for lp in self.length_penalty)
                and all(ct >= mn for ct, mn in zip(self.current_tokens, self.min_new_tokens)))

    def __call__(self, input_ids: torch.Tensor, scores: torch.Tensor):
        if scores.is_cuda:
            next_ids, _, lse, warped = self.choose_fused(input_ids, scores)
            logprobs = warped - lse[:, None] if any(self.return_logprobs) else NONES
            return next_ids, warped, logprobs
        for idx, adj in self._eos_adjustments().items():
            if adj[0] == 1.0:
                scores[idx, self.eos_token_id] = -float("inf")
            else:
                eos = scores[idx, self.eos_token_id]
                # penalise through |logit| so negative logits are handled too
                scores[idx, self.eos_token_id] = eos + torch.abs(eos) * adj[1]
        if self.repetition_processor is not None:
            scores = self.repetition_processor(input_ids, scores)
        for warper in self.warpers:
            scores = warper(input_ids, scores)
        next_ids = self.choice(scores)
        logprobs = torch.log_softmax(scores, -1) if any(self.return_logprobs) else NONES
        return next_ids, scores, logprobs

    def _eos_adjustments(self):
        """{row: (mode, factor)} for this step — mode 1: EOS masked while the request is below min_new_tokens,
        mode 2: EOS score += |score| * factor (length penalty) — and the step bookkeeping of tokens.py:242-256."""
        rows = {}
        for idx in range(len(self.current_tokens)):
            cur, lp = self.current_tokens[idx], self.length_penalty[idx]
            if cur < self.min_new_tokens[idx]:
                rows[idx] = (1.0, 0.0)
                self.current_tokens[idx] += 1
            elif lp is not None:
                tokens_past = cur - lp[0]
                if tokens_past > 0:
                    rows[idx] = (2.0, pow(lp[1], tokens_past) - 1)
                self.current_tokens[idx] += 1
        return rows

    def _fused_params(self, device):
        if self._fused is not None:
            return self._fused
        B = len(self.do_sample)

        def row_f32(t):
            return t.detach().to(device=device, dtype=torch.float32).reshape(-1).contiguous()

        p = {"temperature": None, "top_k": None, "top_p_cut": None, "typical_p": None, "rep_penalty": None,
             "exclude_id": -1, "do_sample": None, "rng": None}
        for w in self.warpers:  # exactly the processors the torch chain would run, with the values it would use
            if isinstance(w, HeterogeneousTemperatureLogitsWarper):
                p["temperature"] = row_f32(w.tensor)
            elif isinstance(w, HeterogeneousTopKLogitsWarper):
                keep = w.min_tokens_to_keep
                p["top_k"] = torch.tensor([max(k, keep) if k else 0 for k in w.top_k], dtype=torch.int32,
                                          device=device)
            elif isinstance(w, HeterogeneousTopPLogitsWarper):
                p["top_p_cut"] = row_f32(w.tensor)  # already 1 - top_p
            elif isinstance(w, HeterogeneousTypicalLogitsWarper):
                p["typical_p"] = row_f32(w.tensor)
        if self.repetition_processor is not None:
            p["rep_penalty"] = row_f32(self.repetition_processor.tensor)
            ex = self.repetition_processor.id_to_exclude
            p["exclude_id"] = ex if ex is not None and B != 1 else -1
        if isinstance(self.choice, HeterogeneousSampling):
            p["do_sample"] = torch.tensor([int(x) for x in self.do_sample], dtype=torch.int32, device=device)
            p["rng"] = torch.tensor([s.state if s is not None else (0, 0) for s in self.choice.samplings],
                                    dtype=torch.int64, device=device)
        self._fused = p
        return p

    def choose_fused(self, input_ids: torch.Tensor, scores: torch.Tensor):
        """(ids int64 [B], logprob f32 [B] of the chosen id under the warped scores, lse f32 [B], warped scores
        f32 [B, V]) by one launch of tgis_warp_sample; log_softmax(warped)[b] = warped[b] - lse[b]."""
        p = self._fused_params(scores.device)
        adj = self._eos_adjustments()
        eos_adjust = None
        if adj:
            host = torch.zeros((len(self.do_sample), 2), dtype=torch.float32)
            for idx, a in adj.items():
                host[idx, 0], host[idx, 1] = a
            eos_adjust = host.to(scores.device, non_blocking=True)
        logits = scores if scores.dtype == torch.float32 else scores.float()
        if logits.stride(-1) != 1:
            logits = logits.contiguous()
        out = native.warp_sample(
            logits, temperature=p["temperature"], top_k=p["top_k"], top_p_cut=p["top_p_cut"],
            typical_p=p["typical_p"], rep_penalty=p["rep_penalty"],
            input_ids=input_ids if p["rep_penalty"] is not None else None, exclude_id=p["exclude_id"],
            eos_adjust=eos_adjust, eos_id=self.eos_token_id if eos_adjust is not None else -1,
            do_sample=p["do_sample"], rng=p["rng"])
        if p["rng"] is not None:  # host mirror of the offsets the kernel advanced
            for s in self.choice.sampling_mapping.values():
                s.offset += 1
        return out

    def choose_greedy_fused(self, scores: torch.Tensor):
        """(ids int64 [B], logprob f32 [B]) by one kernel; valid only when `is_plain_greedy`."""
        return native.argmax_logprob(scores)

    @classmethod
    def from_pb(cls, pb: List[generate_pb2.NextTokenChooserParameters], model_eos_token_id: Optional[int],
                model_pad_token_id: Optional[int], return_logprobs: List[bool], dtype, device,
                samplings: Optional[List[Sampling]] = None, current_tokens: Optional[List[int]] = None):
        seeds = samplings if samplings else [p.seed if p.HasField("seed") else None for p in pb]
        return cls(
            temperature=[p.temperature for p in pb],
            repetition_penalty=[p.repetition_penalty if p.HasField("repetition_penalty") else 1.0 for p in pb],
            top_k=[p.top_k for p in pb],
            top_p=[p.top_p if p.top_p > 0 else 1.0 for p in pb],          # 0 means "disabled"
            typical_p=[p.typical_p if p.typical_p > 0 else 1.0 for p in pb],
            length_penalty=[(p.length_penalty.start_index, p.length_penalty.decay_factor)
                            if p.HasField("length_penalty") else None for p in pb],
            seeds=seeds,
            min_new_tokens=[p.min_new_tokens for p in pb],
            eos_token_id=model_eos_token_id, pad_token_id=model_pad_token_id,
            return_logprobs=return_logprobs, device=device, dtype=dtype, current_tokens=current_tokens)

    def filter(self, indices):
        if self.repetition_processor is not None:
            self.repetition_processor = self.repetition_processor.filter(indices)
        self.warpers = [w2 for w2 in (w.filter(indices) for w in self.warpers) if w2 is not None]
        self.do_sample = [self.do_sample[i] for i in indices]
        self.current_tokens = [self.current_tokens[i] for i in indices]
        self.min_new_tokens = [self.min_new_tokens[i] for i in indices]
        self.length_penalty = [self.length_penalty[i] for i in indices]
        self.return_logprobs = [self.return_logprobs[i] for i in indices]
        if any(self.do_sample):
            self.choice.filter(indices)
        else:
            self.choice = Greedy()
        self._fused = None
        return self


def _sorted_desc(tts: List[TopToken]) -> List[TopToken]:
    tts.sort(reverse=True)
    return tts


def get_token_info(request: generate_pb2.Request, scores: torch.Tensor, next_token: torch.Tensor,
                   logprobs: Optional[torch.Tensor]) -> TokenInfo:
    """scores / logprobs: [1, vocab]; next_token: 1-element tensor."""
    tok = next_token.item()
    info = TokenInfo(request_id=request.id, token_id=tok)
    if logprobs is not None:
        info.logprob = logprobs[-1, tok].item()
    n = request.details.top_n_toks
    if n:
        flat = scores[-1]
        n = min(n, flat.size(-1))
        nth = flat.topk(n).values[-1]
        torch.nan_to_num_(nth, neginf=torch.finfo(flat.dtype).min)  # e.g. top_n > top_k
        # every id whose score ties or beats the n-th best, capped at 4n
        ids = (flat >= nth).nonzero().squeeze(-1)[:n * 4]
        if logprobs is None:
            info.top_tokens = [TopToken(token_id=i.item()) for i in ids]
        else:
            info.top_tokens = _sorted_desc([TopToken(token_id=i.item(), logprob=logprobs[-1, i].item()) for i in ids])
    if request.details.ranks:
        info.rank = int((scores > scores[0, tok]).sum() + 1)
    return info


def get_input_tokens_info(request, input_token_ids: torch.Tensor, all_input_logits: torch.Tensor) -> InputTokens:
    """Details for the prompt tokens: logits row i predicts input token i+1; the first token has no score."""
    want_lp = request.details.logprobs
    targets = input_token_ids[1:].unsqueeze(-1)
    if want_lp:
        all_lp = torch.log_softmax(all_input_logits, -1)
        tok_lp = all_lp.gather(1, targets)
        lp_iter = chain([float("nan")], tok_lp.squeeze(-1))
    else:
        all_lp = tok_lp = None
        lp_iter = repeat(0.0)
    if request.details.ranks:
        if want_lp:
            ranks = (all_lp > tok_lp).sum(dim=1) + 1
        else:
            ranks = (all_input_logits > all_input_logits.gather(1, targets)).sum(dim=1) + 1
        rank_iter = chain([0], ranks)
    else:
        rank_iter = repeat(0)
    n = request.details.top_n_toks
    if n:
        n = min(n, all_input_logits.size(-1))
        nth = torch.topk(all_input_logits, n).values[..., -1, None]
        marked = all_input_logits >= nth
        per_tok = [marked[i].nonzero().squeeze(-1)[:n * 4] for i in range(marked.shape[0])]
        if want_lp and per_tok:
            padded = torch.nn.utils.rnn.pad_sequence(per_tok, batch_first=True)
            lps = all_lp.gather(1, padded)
            top_iter = chain([None], ((ids, lps[i][:len(ids)]) for i, ids in enumerate(per_tok)))
        elif want_lp:
            top_iter = iter([None])
        else:
            top_iter = chain([None], per_tok)
    else:
        top_iter = NONES

    def tops(t):
        if t is None:
            return None
        if not want_lp:
            return [TopToken(int(i)) for i in t]
        return _sorted_desc([TopToken(int(i), float(lp)) for i, lp in zip(*t)])

    return InputTokens(
        request_id=request.id,
        tokens=[TokenInfo(token_id=int(tid), logprob=float(lp), rank=int(rk), top_tokens=tops(tt))
                for tid, lp, rk, tt in zip(input_token_ids, lp_iter, rank_iter, top_iter)])
