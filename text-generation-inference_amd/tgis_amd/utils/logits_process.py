"""Batched logits processors with one parameter value per row, on device.

Same behaviour as the reference's utils/logits_process.py:93-402 (each Heterogeneous* class there must
equal the per-row HF processor, server/tests/test_logit_processors.py is the spec): repetition penalty
(:93-143), temperature (:146-175), top-p (:178-236), top-k (:239-317), typical-p (:320-402).
All run in place on `scores[B, V]`; `filter(indices)` keeps the given rows and returns None when the
processor became a no-op for every remaining row."""
import math
from typing import List, Optional

import torch


class _RowParam:
    """Holds one python value per row plus its [B,1] device tensor."""

    noop_value = None

    def __init__(self, values: List, dtype, device):
        self.values = list(values)
        self.tensor = torch.tensor(self.values, dtype=dtype, device=device).unsqueeze(1)

    def _keep(self, indices) -> bool:
        self.values = [self.values[i] for i in indices]
        if all(v == self.noop_value for v in self.values):
            return False
        self.tensor = self.tensor[indices]
        return True


class HeterogeneousRepetitionPenaltyLogitsProcessor(_RowParam):
    noop_value = 1.0

    def __init__(self, penalty: List[float], dtype, device, id_to_exclude: Optional[int] = None):
        super().__init__(penalty, dtype, device)
        self.id_to_exclude = id_to_exclude

    @property
    def penalty(self):
        return self.values

    def __call__(self, input_ids: torch.Tensor, scores: torch.Tensor) -> torch.Tensor:
        # the excluded id (eos == pad) keeps its score; skipped for a single request (no padding there)
        exclude = self.id_to_exclude is not None and input_ids.shape[0] != 1
        saved = scores[:, self.id_to_exclude].clone() if exclude else None
        seen = scores.gather(1, input_ids)
        seen = torch.where(seen < 0, seen * self.tensor, seen / self.tensor)
        scores.scatter_(1, input_ids, seen)
        if exclude:
            scores[:, self.id_to_exclude] = saved
        return scores

    def filter(self, indices):
        return self if self._keep(indices) else None


class HeterogeneousTemperatureLogitsWarper(_RowParam):
    noop_value = 1.0

    @property
    def temperature(self):
        return self.values

    def __call__(self, input_ids, scores):
        return scores.div_(self.tensor)

    def filter(self, indices):
        return self if self._keep(indices) else None


class HeterogeneousTopPLogitsWarper(_RowParam):
    noop_value = 1.0

    def __init__(self, top_p: List[float], dtype, device, filter_value: float = -math.inf,
                 min_tokens_to_keep: int = 1):
        super().__init__(top_p, dtype, device)
        self.tensor = 1 - self.tensor  # compare ascending cumulative mass against (1 - top_p)
        self.filter_value = filter_value
        self.min_tokens_to_keep = min_tokens_to_keep

    @property
    def top_p(self):
        return self.values

    def __call__(self, input_ids, scores):
        sorted_logits, sorted_idx = torch.sort(scores, descending=False)
        cum = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
        remove_sorted = cum <= self.tensor
        remove_sorted[..., -self.min_tokens_to_keep:] = False
        remove = remove_sorted.scatter(1, sorted_idx, remove_sorted)
        return scores.masked_fill_(remove, self.filter_value)

    def filter(self, indices):
        return self if self._keep(indices) else None


class HeterogeneousTopKLogitsWarper:
    def __init__(self, top_k: List[int], device, filter_value: float = -math.inf, min_tokens_to_keep: int = 1):
        self.top_k = list(top_k)
        self.device = device
        self.filter_value = filter_value
        self.min_tokens_to_keep = min_tokens_to_keep
        self._rebuild()

    def _rebuild(self):
        self.max_top_k = max(self.top_k)
        # 0-based index of the k-th best score; 0 disables top-k for that row
        self.top_k_tensor = torch.tensor([max(k - 1, self.min_tokens_to_keep - 1) for k in self.top_k],
                                         dtype=torch.int64, device=self.device).unsqueeze(1)
        off = [k == 0 for k in self.top_k]
        self.top_k_disabled_mask = (torch.tensor(off, dtype=torch.bool, device=self.device).view(-1, 1)
                                    if any(off) else None)

    def __call__(self, input_ids, scores):
        vocab = scores.size(-1)
        kmax = min(self.max_top_k, vocab)
        idx = self.top_k_tensor.clamp_max(kmax - 1) if self.max_top_k > vocab else self.top_k_tensor
        kth = torch.topk(scores, kmax).values.gather(1, idx)
        if self.top_k_disabled_mask is not None:
            kth.masked_fill_(self.top_k_disabled_mask, self.filter_value)
        return scores.masked_fill_(scores < kth, self.filter_value)

    def filter(self, indices):
        self.top_k = [self.top_k[i] for i in indices]
        if all(k == 0 for k in self.top_k):
            return None
        self._rebuild()
        return self


class HeterogeneousTypicalLogitsWarper(_RowParam):
    noop_value = 1.0

    def __init__(self, mass: List[float], dtype, device, filter_value: float = -math.inf,
                 min_tokens_to_keep: int = 1):
        super().__init__(mass, dtype, device)
        self.device = device
        self.filter_value = filter_value
        self.min_tokens_to_keep = min_tokens_to_keep
        self._mask()

    def _mask(self):
        off = [m == 1.0 for m in self.values]
        self.disabled_mask = torch.tensor(off, dtype=torch.bool, device=self.device) if any(off) else None

    @property
    def mass(self):
        return self.values

    def __call__(self, input_ids, scores):
        logp = torch.nn.functional.log_softmax(scores, dim=-1)
        p = logp.exp()
        entropy = -(logp * p).nansum(-1, keepdim=True)
        dist = ((-logp) - entropy).abs()
        sorted_dist, sorted_idx = torch.sort(dist, descending=False)
        cum = scores.gather(-1, sorted_idx).softmax(dim=-1).cumsum(dim=-1)
        last = (cum < self.tensor).sum(dim=1).clamp_(max=sorted_dist.shape[-1] - 1)
        if self.disabled_mask is not None:
            last.masked_fill_(self.disabled_mask, scores.shape[-1] - 1)
        remove_sorted = sorted_dist > sorted_dist.gather(1, last.view(-1, 1))
        if self.min_tokens_to_keep > 1:
            remove_sorted[..., :self.min_tokens_to_keep] = False
        remove = remove_sorted.scatter(1, sorted_idx, remove_sorted)
        return scores.masked_fill_(remove, self.filter_value)

    def filter(self, indices):
        if not self._keep(indices):
            return None
        self._mask()
        return self
