"""CPU restatement of the next-token chooser for one request row, and of the counter-based generator the GPU draw uses.

TEST INFRASTRUCTURE ONLY (see oracle/ops_ref.py): the checker for `tgis_warp_sample`, never on the product path.

`warp_row` restates, for ONE row in float32 numpy and in the reference's order, what
HeterogeneousNextTokenChooser.__call__ (utils/tokens.py:242-270) does to the scores through the Heterogeneous*
processors (utils/logits_process.py: repetition penalty :93-143, temperature :146-175, top-p :178-236, top-k :239-317,
typical-p :320-402 — each of which the reference's own test, server/tests/test_logit_processors.py, defines as "equal to
the HF per-row processor").  It sorts where the reference sorts; tests/test_sampling_cpu.py pins it to the HF
processors.

`philox4x32_10` is the Random123 generator (Salmon et al., SC'11), pinned to its published known-answer vectors in
tests/test_sampling_cpu.py; `race_choice` restates the draw: argmax_i p_i / E_i with E_i = -log(u_i) ~ Exp(1), the form
the reference's `Sampling` (utils/tokens.py:32-41, `torch.multinomial`) samples from — a categorical draw from
softmax(scores).

Pinned: `warp_row` against tests/golden/chooser_reference.npz — outputs of the reference's own
HeterogeneousNextTokenChooser run on CPU in the build container (tests/golden/make_chooser_fixture.py) — and against
the HF per-row processors; `philox4x32_10` against Random123's known answers (tests/test_sampling_cpu.py)."""
from typing import Optional, Sequence, Tuple

import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_U32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter: np.ndarray, key: Tuple[int, int]) -> np.ndarray:
    """counter uint32 [..., 4] -> uint32 [..., 4]; key = (k0, k1)."""
    c = counter.astype(np.uint64)
    c0, c1, c2, c3 = c[..., 0], c[..., 1], c[..., 2], c[..., 3]
    k0, k1 = key[0] & 0xFFFFFFFF, key[1] & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & _U32, p1 >> np.uint64(32), p1 & _U32
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def race_uniforms(seed: int, offset: int, n: int) -> np.ndarray:
    """u_i in (0, 1) for i < n of draw number `offset` of the stream `seed`: first word of
    Philox(counter = (i, 0, offset_lo, offset_hi), key = (seed_lo, seed_hi)), mapped as the kernel maps it."""
    ctr = np.zeros((n, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(n, dtype=np.uint32)
    ctr[:, 2] = offset & 0xFFFFFFFF
    ctr[:, 3] = (offset >> 32) & 0xFFFFFFFF
    r = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))[:, 0]
    u = (r.astype(np.float32) + np.float32(0.5)) * np.float32(2.3283064365386963e-10)
    return np.minimum(u, np.float32(0.99999994))


def race_choice(scores: np.ndarray, seed: int, offset: int) -> Tuple[int, float]:
    """(chosen index, margin to the runner-up in the race's log domain)."""
    u = race_uniforms(seed, offset, scores.shape[0]).astype(np.float64)
    with np.errstate(divide="ignore"):
        g = scores.astype(np.float64) - scores.max() - np.log(-np.log(u))
    g[np.isneginf(scores)] = -np.inf
    order = np.argsort(-g, kind="stable")
    return int(order[0]), float(g[order[0]] - g[order[1]])


def _softmax(x: np.ndarray) -> np.ndarray:
    e = np.exp(x - x.max(), dtype=np.float32)
    return e / e.sum(dtype=np.float32)


def warp_row(logits: np.ndarray, *, temperature: float = 1.0, top_k: int = 0, top_p_cut: float = 0.0,
             typical_p: float = 1.0, rep_penalty: float = 1.0, input_ids: Optional[Sequence[int]] = None,
             exclude_id: int = -1, eos_id: int = -1, eos_mode: int = 0, eos_factor: float = 0.0) -> np.ndarray:
    """Warped scores of one row (float32, -inf where filtered)."""
    s = logits.astype(np.float32).copy()
    f32 = np.float32
    if eos_mode == 1:  # tokens.py:244-246
        s[eos_id] = -np.inf
    elif eos_mode == 2:  # tokens.py:247-254
        s[eos_id] = s[eos_id] + np.abs(s[eos_id]) * f32(eos_factor)
    if rep_penalty != 1.0 and input_ids is not None:  # logits_process.py:118-136
        ids = np.unique(np.asarray(input_ids, dtype=np.int64))
        ids = ids[ids != exclude_id]
        seen = s[ids]
        s[ids] = np.where(seen < 0, seen * f32(rep_penalty), seen / f32(rep_penalty))
    s = s / f32(temperature)  # :164-166
    V = s.shape[0]
    if top_k:  # :283-303
        kth = np.sort(s)[::-1][min(top_k, V) - 1]
        s[s < kth] = -np.inf
    if top_p_cut > 0.0:  # :209-222
        order = np.argsort(s, kind="stable")
        cum = np.cumsum(_softmax(s[order]), dtype=np.float32)
        remove = cum <= f32(top_p_cut)
        remove[-1] = False
        s[order[remove]] = -np.inf
    if typical_p < 1.0:  # :360-389
        m = s.max()
        logp = (s - m) - np.log(np.exp(s - m, dtype=np.float32).sum(dtype=np.float32))
        with np.errstate(invalid="ignore"):
            ent = -np.nansum(np.exp(logp) * logp, dtype=np.float32)
            dist = np.abs((-logp) - ent)
        order = np.argsort(dist, kind="stable")
        cum = np.cumsum(_softmax(s[order]), dtype=np.float32)
        last = min(int((cum < f32(typical_p)).sum()), V - 1)
        s[dist > dist[order[last]]] = -np.inf
    return s
