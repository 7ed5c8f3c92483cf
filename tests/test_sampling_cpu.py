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


# ---- the oracle of the fused GPU chooser (oracle/sampler_ref.py) --------------------------------------------------
def test_philox_known_answers():
    """Random123's published known-answer vectors for philox4x32-10 (kat_vectors: zero, all-ones, pi digits)."""
    import numpy as np

    from oracle.sampler_ref import philox4x32_10

    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        assert tuple(int(x) for x in philox4x32_10(np.array(ctr, dtype=np.uint32), key)) == want


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_oracle_warp_row_equals_hf_processors(seed):
    """oracle.sampler_ref.warp_row, the checker of tgis_warp_sample, against the HF per-row processors in the
    reference's order (repetition penalty, temperature, top-k, top-p, typical-p)."""
    from oracle.sampler_ref import warp_row

    V = 500
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(1, V, generator=g) * 4
    ids = torch.randint(0, V, (1, 30), generator=g)
    cases = [dict(temperature=0.7), dict(top_k=12), dict(top_p=0.85), dict(typical_p=0.6), dict(rep_penalty=1.3),
             dict(temperature=1.4, top_k=40, top_p=0.9, typical_p=0.8, rep_penalty=1.15)]
    for c in cases:
        want = logits.clone()
        if "rep_penalty" in c:
            want = RepetitionPenaltyLogitsProcessor(penalty=c["rep_penalty"])(ids, want)
        if "temperature" in c:
            want = TemperatureLogitsWarper(c["temperature"])(ids, want)
        if "top_k" in c:
            want = TopKLogitsWarper(c["top_k"])(ids, want)
        if "top_p" in c:
            want = TopPLogitsWarper(c["top_p"])(ids, want)
        if "typical_p" in c:
            want = TypicalLogitsWarper(mass=c["typical_p"])(ids, want)
        got = warp_row(logits[0].numpy(), temperature=c.get("temperature", 1.0), top_k=c.get("top_k", 0),
                       top_p_cut=float(1 - torch.tensor(c["top_p"])) if "top_p" in c else 0.0,
                       typical_p=c.get("typical_p", 1.0), rep_penalty=c.get("rep_penalty", 1.0),
                       input_ids=ids[0].tolist())
        got = torch.from_numpy(got)
        assert torch.equal(torch.isinf(got), torch.isinf(want[0])), c
        keep = ~torch.isinf(got)
        assert torch.allclose(got[keep], want[0][keep], rtol=1e-6, atol=1e-6), c


def test_race_choice_is_a_categorical_draw():
    """The exponential race over Philox uniforms samples softmax(scores): chi-square over 20000 draws."""
    import numpy as np

    from oracle.sampler_ref import race_choice

    scores = np.array([2.0, 1.0, 0.0, -1.0, -np.inf, 0.5], dtype=np.float32)
    p = np.exp(scores - scores.max())
    p /= p.sum()
    n = 20000
    counts = np.bincount([race_choice(scores, seed=77, offset=o)[0] for o in range(n)], minlength=6)
    assert counts[4] == 0
    live = p > 0
    chi2 = (((counts - n * p) ** 2)[live] / (n * p[live])).sum()
    assert chi2 < 20.5  # 4 degrees of freedom: P(chi2 > 20.5) < 4e-4


def test_sampling_state_and_seedless_seeds():
    from tgis_amd.utils import tokens

    s = tokens.Sampling(seed=(1 << 63) + 5)
    assert s.state == ((1 << 63) + 5 - (1 << 64), 0)
    s(torch.randn(10))
    assert s.state[1] == 1
    # seedless requests: a deterministic function of (base, arrival number), identical on every TP rank
    tokens.set_seed_base(1234)
    a = [tokens.Sampling().seed for _ in range(3)]
    tokens.set_seed_base(1234)
    b = [tokens.Sampling().seed for _ in range(3)]
    assert a == b and len(set(a)) == 3


def test_eos_adjustments_follow_the_reference_bookkeeping():
    ps = [_params(min_new_tokens=2), _params(), _params()]
    ps[1].length_penalty.start_index = 1
    ps[1].length_penalty.decay_factor = 1.5
    ch = HeterogeneousNextTokenChooser.from_pb(ps, 2, 0, [False] * 3, torch.float32, "cpu")
    assert ch._eos_adjustments() == {0: (1.0, 0.0)} and ch.current_tokens == [1, 1, 0]
    assert ch._eos_adjustments() == {0: (1.0, 0.0)} and ch.current_tokens == [2, 2, 0]
    assert ch._eos_adjustments() == {1: (2.0, 0.5)} and ch.current_tokens == [2, 3, 0]
    assert ch._eos_adjustments() == {1: (2.0, 1.25)}


# ---- golden vectors from the reference's own chooser (tests/golden/make_chooser_fixture.py) -----------------------
def _chooser_golden():
    import os

    import numpy as np

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chooser_reference.npz"))
    cases = sorted({k.split(".")[0] for k in z.files})
    for name in cases:
        n = len([k for k in z.files if k.startswith(f"{name}.params.")])
        params = [pb.NextTokenChooserParameters.FromString(z[f"{name}.params.{i}"].tobytes()) for i in range(n)]
        yield name, params, z


def test_host_chooser_equals_reference_golden():
    """The torch chain (the semantics the GPU kernel is held to) against the reference's chooser, two consecutive calls:
    filtered sets and greedy ids exact, surviving scores and log-probabilities to fp32 rounding."""
    for name, params, z in _chooser_golden():
        ch = HeterogeneousNextTokenChooser.from_pb(params, 2, 2, [True] * len(params), torch.float32, "cpu")
        ids = torch.from_numpy(z[f"{name}.ids"])
        for step in range(2):
            next_ids, scores, logprobs = ch(ids, torch.from_numpy(z[f"{name}.{step}.logits"]).clone())
            want = torch.from_numpy(z[f"{name}.{step}.scores"])
            assert torch.equal(torch.isinf(scores), torch.isinf(want)), (name, step)
            keep = ~torch.isinf(want)
            assert torch.allclose(scores[keep], want[keep], rtol=1e-6, atol=1e-6)
            assert torch.allclose(logprobs[keep], torch.from_numpy(z[f"{name}.{step}.logprobs"])[keep], rtol=1e-5, atol=1e-5)
            greedy = torch.from_numpy(z[f"{name}.{step}.greedy_rows"])
            assert torch.equal(next_ids[greedy], torch.from_numpy(z[f"{name}.{step}.next_ids"])[greedy])


def test_oracle_warp_row_equals_reference_golden():
    """oracle/sampler_ref.py (the checker of tgis_warp_sample) against the same vectors, row by row."""
    import numpy as np

    from oracle.sampler_ref import warp_row

    for name, params, z in _chooser_golden():
        B = len(params)
        ch = HeterogeneousNextTokenChooser.from_pb(params, 2, 2, [True] * B, torch.float32, "cpu")  # host bookkeeping only
        sampled = any(p.temperature != 0 for p in params)
        ids = z[f"{name}.ids"]
        for step in range(2):
            adj = ch._eos_adjustments()
            for b, p in enumerate(params):
                mode, factor = adj.get(b, (0.0, 0.0))
                rep = p.repetition_penalty if p.HasField("repetition_penalty") else 1.0
                got = warp_row(
                    z[f"{name}.{step}.logits"][b],
                    temperature=(p.temperature or 1.0) if sampled else 1.0, top_k=p.top_k if sampled else 0,
                    top_p_cut=float(np.float32(1) - np.float32(p.top_p)) if sampled and 0 < p.top_p < 1 else 0.0,
                    typical_p=p.typical_p if sampled and 0 < p.typical_p < 1 else 1.0, rep_penalty=rep,
                    input_ids=ids[b], exclude_id=2 if B != 1 else -1, eos_id=2, eos_mode=int(mode), eos_factor=factor)
                want = z[f"{name}.{step}.scores"][b]
                assert np.array_equal(np.isneginf(got), np.isneginf(want)), (name, step, b)
                keep = ~np.isneginf(want)
                np.testing.assert_allclose(got[keep], want[keep], rtol=1e-6, atol=1e-6)
