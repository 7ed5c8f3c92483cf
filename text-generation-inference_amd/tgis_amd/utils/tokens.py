"""Next-token selection for a heterogeneous batch, and extraction of the per-token details the router asks for.

Mirrors utils/tokens.py of the reference: `Sampling` / `Greedy` (:32-46), `HeterogeneousNextTokenChooser`
(:161-333: per-row min-new-tokens EOS mask and length penalty :242-256, repetition penalty, warpers,
choice, log-softmax only when some request wants logprobs), `HeterogeneousSampling` (:336-385),
`get_token_info` (:388-425) and `get_input_tokens_info` (:429-506).

Difference: when the whole batch is plain greedy with no processors, `choose_greedy_fused` runs one HIP
kernel (argmax + logprob) and the caller does a single device->host copy instead of one `.item()` per request."""
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


class Sampling:
    """softmax(logits) / Exp(1) noise -> argmax == a multinomial draw without a host sync; one generator per
    request so concatenate / prune keep every request's RNG stream (tokens.py:32-41)."""

    def __init__(self, seed: Optional[int] = None, device="cpu"):
        self.generator = None if seed is None else torch.Generator(device).manual_seed(seed)

    def __call__(self, logits: torch.Tensor) -> torch.Tensor:
        probs = torch.nn.functional.softmax(logits, -1)
        q = torch.empty_like(probs).exponential_(1, generator=self.generator)
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
        for idx in range(len(self.current_tokens)):
            cur, lp = self.current_tokens[idx], self.length_penalty[idx]
            if cur < self.min_new_tokens[idx]:
                scores[idx, self.eos_token_id] = -float("inf")
                self.current_tokens[idx] += 1
            elif lp is not None:
                tokens_past = cur - lp[0]
                if tokens_past > 0:
                    eos = scores[idx, self.eos_token_id]
                    # penalise through |logit| so negative logits are handled too
                    scores[idx, self.eos_token_id] = eos + torch.abs(eos) * (pow(lp[1], tokens_past) - 1)
                self.current_tokens[idx] += 1
        if self.repetition_processor is not None:
            scores = self.repetition_processor(input_ids, scores)
        for warper in self.warpers:
            scores = warper(input_ids, scores)
        next_ids = self.choice(scores)
        logprobs = torch.log_softmax(scores, -1) if any(self.return_logprobs) else NONES
        return next_ids, scores, logprobs

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
