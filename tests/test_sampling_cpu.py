"""CPU: the batched per-row logits processors must equal HF's single-row processors row by row (the spec the
reference uses in server/tests/test_logit_processors.py), and the chooser / token-info helpers follow
utils/tokens.py semantics.  The processors are device-agnostic torch code, so CPU is a faithful check."""
import math

import pytest
import torch
from transformers.generation.logits_process import (
    RepetitionPenaltyLogitsProcessor,
    TemperatureLogitsWarper,
    TopKLogitsWarper,
    TopPLogitsWarper,
    TypicalLogitsWarper,
)

from tgis_amd.pb import generate_pb2 as pb
from tgis_amd.utils import logits_process as lp
from tgis_amd.utils.tokens import HeterogeneousNextTokenChooser, get_input_tokens_info, get_token_info

VOCAB = 41
IDS = torch.tensor([[1, 2, 1, 3, 4, 6, 7, 1, 1, 1], [1, 7, 0, 3, 4, 6, 7, 1, 1, 1], [5, 5, 5, 9, 9, 0, 2, 2, 8, 3]])


def _scores(seed=0):
    return torch.randn(3, VOCAB, generator=torch.Generator().manual_seed(seed)) * 3


def _rows(fn_per_row, scores):
    return torch.stack([fn_per_row(i, scores[i:i + 1].clone()).squeeze(0) for i in range(scores.shape[0])])


def test_repetition_penalty_rows():
    pen = [1.0, 2.5, 1.3]
    got = lp.HeterogeneousRepetitionPenaltyLogitsProcessor(pen, torch.float32, None)(IDS, _scores())
    want = _rows(lambda i, s: RepetitionPenaltyLogitsProcessor(penalty=pen[i])(IDS[i:i + 1], s), _scores())
    assert torch.allclose(got, want)


def test_repetition_penalty_excludes_pad_eos():
    s = _scores()
    got = lp.HeterogeneousRepetitionPenaltyLogitsProcessor([2.0] * 3, torch.float32, None, id_to_exclude=1)(IDS, s.clone())
    assert torch.equal(got[:, 1], s[:, 1])  # id 1 occurs in every row but keeps its score
    single = lp.HeterogeneousRepetitionPenaltyLogitsProcessor([2.0], torch.float32, None, id_to_exclude=1)(IDS[:1], s[:1].clone())
    assert not torch.equal(single[:, 1], s[:1, 1])  # no exclusion for a single request (no padding there)


def test_temperature_topk_topp_typical_rows():
    t = [0.25, 1.0, 1.7]
    got = lp.HeterogeneousTemperatureLogitsWarper(t, torch.float32, None)(IDS, _scores(1))
    want = _rows(lambda i, s: TemperatureLogitsWarper(t[i])(IDS[i:i + 1], s), _scores(1))
    assert torch.allclose(got, want)

    k = [5, 0, 100]  # 0 disables, 100 > vocab clamps
    got = lp.HeterogeneousTopKLogitsWarper(k, None)(IDS, _scores(2))
    want = _rows(lambda i, s: s if k[i] == 0 else TopKLogitsWarper(min(k[i], VOCAB))(IDS[i:i + 1], s), _scores(2))
    assert torch.equal(got, want)

    p = [0.3, 1.0, 0.9]
    got = lp.HeterogeneousTopPLogitsWarper(p, torch.float32, None)(IDS, _scores(3))
    want = _rows(lambda i, s: s if p[i] == 1.0 else TopPLogitsWarper(p[i])(IDS[i:i + 1], s), _scores(3))
    assert torch.equal(torch.isinf(got), torch.isinf(want)) and torch.allclose(got[~torch.isinf(got)], want[~torch.isinf(want)])

    m = [0.5, 1.0, 0.2]
    got = lp.HeterogeneousTypicalLogitsWarper(m, torch.float32, None)(IDS, _scores(4))
    want = _rows(lambda i, s: s if m[i] == 1.0 else TypicalLogitsWarper(mass=m[i])(IDS[i:i + 1], s), _scores(4))
    assert torch.equal(torch.isinf(got), torch.isinf(want))


def test_filter_drops_noop_processors():
    w = lp.HeterogeneousTemperatureLogitsWarper([0.5, 1.0, 1.0], torch.float32, None)
    assert w.filter([1, 2]) is None
    w = lp.HeterogeneousTopKLogitsWarper([3, 0, 7], None)
    assert w.filter([1]) is None
    w = lp.HeterogeneousTopKLogitsWarper([3, 0, 7], None).filter([0, 2])
    assert w.top_k == [3, 7] and w.top_k_disabled_mask is None
    w = lp.HeterogeneousTypicalLogitsWarper([0.5, 1.0, 0.2], torch.float32, None).filter([1, 2])
    assert w.mass == [1.0, 0.2] and w.disabled_mask.tolist() == [True, False]


def _params(**kw):
    p = pb.NextTokenChooserParameters()
    for k, v in kw.items():
        if k == "length_penalty":
            p.length_penalty.start_index, p.length_penalty.decay_factor = v
        else:
            setattr(p, k, v)
    return p


def test_chooser_greedy_min_new_tokens_and_length_penalty():
    ps = [_params(), _params(min_new_tokens=2), _params(length_penalty=(1, 1.5))]
    ch = HeterogeneousNextTokenChooser.from_pb(ps, 2, 0, [True, False, True], torch.float32, "cpu")
    assert not ch.is_plain_greedy  # row 1 still owes EOS masking, row 2 has a length penalty
    base = torch.zeros(3, VOCAB)
    base[:, 2] = 5.0  # EOS is the argmax everywhere
    base[:, 7] = 4.0
    ids, scores, logprobs = ch(IDS, base.clone())
    assert ids.tolist() == [2, 7, 2] and math.isinf(float(scores[1, 2]))
    assert torch.allclose(logprobs.exp().sum(-1), torch.ones(3))
    assert ch.current_tokens == [0, 1, 1]
    ids, scores, _ = ch(IDS, base.clone())
    ids, scores, _ = ch(IDS, base.clone())  # row 1 now past min_new_tokens; row 2: tokens_past = 2-1 = 1
    assert ids[1] == 2
    assert float(scores[2, 2]) == pytest.approx(5.0 + 5.0 * (1.5 ** 1 - 1))
    plain = HeterogeneousNextTokenChooser.from_pb([_params(), _params()], 2, 0, [False, True], torch.float32, "cpu")
    assert plain.is_plain_greedy


def test_chooser_sampling_is_seeded_and_survives_filter():
    ps = [_params(temperature=0.8, seed=5), _params(), _params(temperature=1.2, seed=9, top_k=8)]
    ch = HeterogeneousNextTokenChooser.from_pb(ps, 2, 0, [False] * 3, torch.float32, "cpu")
    s = _scores(7)
    a, _, _ = ch(IDS, s.clone())
    ch2 = HeterogeneousNextTokenChooser.from_pb(ps, 2, 0, [False] * 3, torch.float32, "cpu")
    b, _, _ = ch2(IDS, s.clone())
    assert a.tolist() == b.tolist() and a[1] == s[1].argmax()  # greedy row unaffected by its neighbours
    # filter keeps the RNG streams of the kept rows: next draw of row 2 equals an unfiltered chooser's next draw
    nxt_full, _, _ = ch(IDS, s.clone())
    ch2.filter([2])
    nxt_filtered, _, _ = ch2(IDS[2:], s[2:].clone())
    assert int(nxt_filtered[0]) == int(nxt_full[2])
    # concatenate path: samplings handed over through from_pb keep their state as well
    ch3 = HeterogeneousNextTokenChooser.from_pb(ps, 2, 0, [False] * 3, torch.float32, "cpu", samplings=ch.samplings,
                                                current_tokens=ch.current_tokens)
    assert ch3.samplings[0] is ch.samplings[0] and ch3.samplings[1] is None


def test_token_info_top_n_rank_and_input_tokens():
    r = pb.Request(id=4)
    r.details.logprobs = True
    r.details.top_n_toks = 2
    r.details.ranks = True
    scores = torch.tensor([[0.1, 3.0, 2.0, 2.0, -1.0]])
    logprobs = torch.log_softmax(scores, -1)
    info = get_token_info(r, scores, torch.tensor([2]), logprobs)
    # n-th best value is 2.0 and ties with id 3: all ids >= 2.0 are returned, sorted by logprob then LOWER id
    assert [t.token_id for t in info.top_tokens] == [1, 2, 3] and info.rank == 2
    assert info.logprob == pytest.approx(float(logprobs[0, 2]))
    ids = torch.tensor([3, 1, 4])
    logits = torch.tensor([[0.0, 2.0, 1.0, 0.0, 0.5], [0.0, 0.0, 0.0, 0.0, 9.0]])
    it = get_input_tokens_info(r, ids, logits)
    assert [t.token_id for t in it.tokens] == [3, 1, 4]
    assert math.isnan(it.tokens[0].logprob) and it.tokens[0].rank == 0 and it.tokens[0].top_tokens is None
    assert it.tokens[1].rank == 1 and it.tokens[2].rank == 1
    assert it.tokens[1].logprob == pytest.approx(float(torch.log_softmax(logits[0], -1)[1]))
