"""-m gpu: the fused next-token chooser (`tgis_warp_sample`, through the C ABI) against oracle/sampler_ref.py — the
row-by-row restatement of the reference's chooser that tests/test_sampling_cpu.py pins to the HF processors — and the
chooser classes end to end (GPU launch vs the torch chain on the host).

Bars: filtered sets (the -inf pattern of the warped scores) and greedy token ids bit-exact; surviving scores, lse and
logprobs within 1e-5 relative (fp32 arithmetic in another association order); sampled ids equal to the oracle's race
over the same Philox stream whenever the race is not a photo finish (margin > 1e-3); draws distributed as
softmax(scores) by chi-square."""
import numpy as np
import pytest
import torch

from oracle import sampler_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat(gpu_device):
    from tgis_amd import native

    native.load_library()
    return native


def _dev(x, dtype, dev):
    return None if x is None else torch.tensor(x, dtype=dtype, device=dev)


def _run(nat, dev, logits, rows, input_ids=None, exclude_id=-1, eos_id=-1, rng=None):
    """rows: list of per-row dicts (temperature, top_k, top_p_cut, typical_p, rep_penalty, eos, sample)."""
    B = len(rows)

    def col(name, default, dtype):
        if all(name not in r for r in rows):
            return None
        return torch.tensor([r.get(name, default) for r in rows], dtype=dtype, device=dev)

    eos = None
    if any("eos" in r for r in rows):
        eos = torch.tensor([r.get("eos", (0.0, 0.0)) for r in rows], dtype=torch.float32, device=dev)
    do_sample = col("sample", 0, torch.int32)
    return nat.warp_sample(
        torch.from_numpy(logits).to(dev), temperature=col("temperature", 1.0, torch.float32),
        top_k=col("top_k", 0, torch.int32), top_p_cut=col("top_p_cut", 0.0, torch.float32),
        typical_p=col("typical_p", 1.0, torch.float32), rep_penalty=col("rep_penalty", 1.0, torch.float32),
        input_ids=None if input_ids is None else torch.from_numpy(input_ids).to(dev), exclude_id=exclude_id,
        eos_adjust=eos, eos_id=eos_id, do_sample=do_sample, rng=rng)


def _oracle_row(logits_row, r, input_ids_row, exclude_id, eos_id):
    mode, factor = r.get("eos", (0.0, 0.0))
    return sampler_ref.warp_row(
        logits_row, temperature=r.get("temperature", 1.0), top_k=r.get("top_k", 0), top_p_cut=r.get("top_p_cut", 0.0),
        typical_p=r.get("typical_p", 1.0), rep_penalty=r.get("rep_penalty", 1.0), input_ids=input_ids_row,
        exclude_id=exclude_id, eos_id=eos_id, eos_mode=int(mode), eos_factor=factor)


F = np.float32
ROWSETS = {
    "temperature": [dict(temperature=0.25), dict(temperature=1.0), dict(temperature=1.7)],
    "top_k": [dict(top_k=1), dict(top_k=50), dict(top_k=0), dict(top_k=10 ** 6)],
    "top_p": [dict(top_p_cut=float(F(1) - F(0.3))), dict(top_p_cut=0.0), dict(top_p_cut=float(F(1) - F(0.9))),
              dict(top_p_cut=float(F(1) - F(0.999)))],
    "typical": [dict(typical_p=0.5), dict(typical_p=1.0), dict(typical_p=0.2), dict(typical_p=0.95)],
    "repetition": [dict(rep_penalty=1.0), dict(rep_penalty=2.5), dict(rep_penalty=1.3)],
    "eos": [dict(eos=(1.0, 0.0)), dict(eos=(2.0, 0.5)), dict(), dict(eos=(2.0, 3.0), rep_penalty=1.2)],
    "all": [dict(temperature=0.8, top_k=50, top_p_cut=float(F(1) - F(0.9)), rep_penalty=1.2),
            dict(temperature=1.3, top_k=200, top_p_cut=float(F(1) - F(0.95)), typical_p=0.9, rep_penalty=1.05,
                 eos=(2.0, 0.25)),
            dict(),
            dict(typical_p=0.7, top_k=1000)],
}


@pytest.mark.parametrize("V", [41, 32000, 50257])
@pytest.mark.parametrize("name", sorted(ROWSETS))
def test_warped_scores_and_greedy_choice_match_oracle(nat, gpu_device, name, V):
    rows = ROWSETS[name]
    B, L = len(rows), 37
    rs = np.random.RandomState(len(name) * 1000 + V)
    logits = (rs.randn(B, V) * 3).astype(np.float32)
    ids = rs.randint(0, V, size=(B, L)).astype(np.int64)
    ids[:, -5:] = 3  # padding-like repeats; id 3 is the excluded one
    eos_id = 7
    use_ids = any("rep_penalty" in r for r in rows)
    got_ids, got_lp, got_lse, got = _run(nat, gpu_device, logits, rows, ids if use_ids else None, exclude_id=3,
                                         eos_id=eos_id)
    got, got_ids, got_lp, got_lse = got.cpu().numpy(), got_ids.cpu().numpy(), got_lp.cpu().numpy(), got_lse.cpu().numpy()
    for b, r in enumerate(rows):
        want = _oracle_row(logits[b], r, ids[b] if use_ids else None, 3, eos_id)
        assert np.array_equal(np.isneginf(got[b]), np.isneginf(want)), \
            f"row {b} {r}: kept {int((~np.isneginf(got[b])).sum())} vs oracle {int((~np.isneginf(want)).sum())}"
        keep = ~np.isneginf(want)
        np.testing.assert_allclose(got[b][keep], want[keep], rtol=1e-6, atol=1e-6)
        assert got_ids[b] == int(np.argmax(want))  # greedy rows: lowest index of the maximum
        m = want.max()
        lse = m + np.log(np.exp((want - m).astype(np.float64)).sum())
        assert abs(got_lse[b] - lse) < 1e-5 * max(1.0, abs(lse))
        assert abs(got_lp[b] - (want[got_ids[b]] - lse)) < 1e-5 * max(1.0, abs(lse))


def test_logits_are_read_only_and_rows_are_independent(nat, gpu_device):
    """A row's result depends on its own scores, parameters and stream only — not on its neighbours or position."""
    V = 32000
    rs = np.random.RandomState(5)
    logits = (rs.randn(6, V) * 2).astype(np.float32)
    rows = [dict(temperature=0.9, top_k=40, top_p_cut=float(F(1) - F(0.9)), sample=1)] * 6
    dev_logits = torch.from_numpy(logits).to(gpu_device)
    rng = torch.tensor([[100 + i, 3] for i in range(6)], dtype=torch.int64, device=gpu_device)
    ids, lps, _, scores = _run(nat, gpu_device, logits, rows, rng=rng.clone())
    assert torch.equal(dev_logits.cpu(), torch.from_numpy(logits))
    perm = [4, 2, 5]
    ids2, lps2, _, scores2 = _run(nat, gpu_device, logits[perm], [rows[i] for i in perm], rng=rng[perm].clone())
    assert ids2.tolist() == [ids.tolist()[i] for i in perm]
    assert torch.equal(scores2, scores[perm]) and torch.equal(lps2, lps[perm])


def test_draws_follow_the_philox_race_and_advance_the_stream(nat, gpu_device):
    V, B = 32000, 16
    rs = np.random.RandomState(11)
    logits = (rs.randn(B, V) * 2.5).astype(np.float32)
    rows = [dict(temperature=0.8, top_k=(0 if b % 2 else 64), sample=1) for b in range(B)]
    rows[3] = dict(temperature=1.0)  # a greedy row in the middle: no draw, stream untouched
    seeds = [(1 << 40) * (b + 1) + 17 for b in range(B)]
    seeds[5] = -3  # a seed above 2^63 in two's complement
    rng = torch.tensor([[s, 9] for s in seeds], dtype=torch.int64, device=gpu_device)
    exact = 0
    for draw in range(3):
        ids, _, _, scores = _run(nat, gpu_device, logits, rows, rng=rng)
        scores = scores.cpu().numpy()
        for b in range(B):
            if b == 3:
                assert int(ids[b]) == int(np.argmax(scores[b]))
                continue
            want, margin = sampler_ref.race_choice(scores[b], seeds[b] & 0xFFFFFFFFFFFFFFFF, 9 + draw)
            if margin > 1e-3:
                assert int(ids[b]) == want, f"row {b} draw {draw}"
                exact += 1
            assert not np.isneginf(scores[b][int(ids[b])])
    assert exact >= 40
    off = rng[:, 1].tolist()
    assert off[3] == 9 and all(o == 12 for i, o in enumerate(off) if i != 3)


def test_nan_rows_do_not_fault(nat, gpu_device):
    """Overflowed logits (NaN / inf rows) must yield some in-range id, never an out-of-bounds access."""
    V = 32000
    logits = np.full((3, V), np.nan, dtype=np.float32)
    logits[1] = np.inf
    logits[2, :100] = 1.0
    rows = [dict(temperature=0.8, top_k=5, top_p_cut=0.1, typical_p=0.5, sample=1)] * 3
    rng = torch.tensor([[1, 0], [2, 0], [3, 0]], dtype=torch.int64, device=gpu_device)
    ids, _, _, _ = _run(nat, gpu_device, logits, rows, rng=rng)
    torch.cuda.synchronize()
    assert all(0 <= i < V for i in ids.tolist())


def test_draws_are_distributed_as_softmax(nat, gpu_device):
    """4096 independent streams over the same 24-token row (one launch): chi-square against softmax(scores / T)
    restricted to the top-k survivors."""
    V, B = 24, 4096
    rs = np.random.RandomState(3)
    row = (rs.randn(V) * 1.5).astype(np.float32)
    logits = np.tile(row, (B, 1))
    rows = [dict(temperature=0.7, top_k=12, sample=1)] * B
    rng = torch.tensor([[b * 7919 + 1, 0] for b in range(B)], dtype=torch.int64, device=gpu_device)
    ids, _, _, _ = _run(nat, gpu_device, logits, rows, rng=rng)
    want = sampler_ref.warp_row(row, temperature=0.7, top_k=12)
    p = np.exp(want - want.max())
    p /= p.sum()
    counts = np.bincount(ids.cpu().numpy(), minlength=V)
    assert counts[p == 0].sum() == 0
    live = p > 0
    chi2 = (((counts - B * p) ** 2)[live] / (B * p[live])).sum()
    assert chi2 < 33.0  # 11 degrees of freedom: P(chi2 > 33) < 6e-4


def test_chooser_gpu_equals_host_chain(nat, gpu_device):
    """HeterogeneousNextTokenChooser end to end: the same requests, logits on the GPU (one tgis_warp_sample launch) vs
    logits on the host (the torch chain that mirrors the reference): same warped scores, same greedy ids, same
    logprobs; sampled rows stay inside their filtered set; concatenate / prune keep the streams."""
    from tgis_amd.pb import generate_pb2 as pb
    from tgis_amd.utils.tokens import HeterogeneousNextTokenChooser

    def params(**kw):
        p = pb.NextTokenChooserParameters()
        for k, v in kw.items():
            if k == "length_penalty":
                p.length_penalty.start_index, p.length_penalty.decay_factor = v
            else:
                setattr(p, k, v)
        return p

    V = 32000
    ps = [params(temperature=0.8, top_k=50, top_p=0.9, seed=5, repetition_penalty=1.2),
          params(repetition_penalty=1.3, min_new_tokens=2),
          params(temperature=1.2, typical_p=0.8, seed=9),
          params(length_penalty=(1, 1.5)),
          params(temperature=0.5, top_p=0.5)]
    g = torch.Generator().manual_seed(0)
    all_ids = torch.randint(0, V, (5, 21), generator=g)

    def make(device):
        return HeterogeneousNextTokenChooser.from_pb(ps, 2, 0, [True] * 5, torch.float32, device)

    gpu, host = make(gpu_device), make("cpu")
    for step in range(4):
        logits = torch.randn(5, V, generator=g) * 3
        ids_g, scores_g, lps_g = gpu(all_ids.to(gpu_device), logits.to(gpu_device))
        ids_h, scores_h, lps_h = host(all_ids, logits.clone())
        scores_g, lps_g = scores_g.cpu(), lps_g.cpu()
        assert torch.equal(torch.isinf(scores_g), torch.isinf(scores_h)), f"step {step}"
        keep = ~torch.isinf(scores_h)
        assert torch.allclose(scores_g[keep], scores_h[keep], rtol=1e-6, atol=1e-6)
        assert torch.allclose(lps_g[keep], lps_h[keep], rtol=1e-5, atol=1e-5)
        for b in (1, 3):  # greedy rows
            assert int(ids_g[b]) == int(ids_h[b])
        for b in (0, 2, 4):
            assert not torch.isinf(scores_g[b, int(ids_g[b])])
        assert gpu.current_tokens == host.current_tokens
    assert [s.offset if s else None for s in gpu.samplings] == [4, None, 4, None, 4]
    # prune to rows (2, 4) and rebuild a chooser from the carried samplings (concatenate path): streams continue
    full = make(gpu_device)
    for s_new, s_old in zip(full.samplings, gpu.samplings):
        if s_new is not None:
            s_new.seed, s_new.offset = s_old.seed, s_old.offset  # row 4 brought no seed: copy the one it was given
    logits = (torch.randn(5, V, generator=g) * 3).to(gpu_device)
    want, _, _ = full(all_ids.to(gpu_device), logits)
    gpu.filter([2, 4])
    got, _, _ = gpu(all_ids[[2, 4]].to(gpu_device), logits[[2, 4]])
    assert got.tolist() == [int(want[2]), int(want[4])]
    carried = HeterogeneousNextTokenChooser.from_pb([ps[2], ps[4]], 2, 0, [True] * 2, torch.float32, gpu_device,
                                                    samplings=gpu.samplings, current_tokens=gpu.current_tokens)
    logits2 = (torch.randn(5, V, generator=g) * 3).to(gpu_device)
    want2, _, _ = full(all_ids.to(gpu_device), logits2)
    got2, _, _ = carried(all_ids[[2, 4]].to(gpu_device), logits2[[2, 4]])
    assert got2.tolist() == [int(want2[2]), int(want2[4])]


def test_fused_chooser_equals_reference_golden(nat, gpu_device):
    """tests/golden/chooser_reference.npz: outputs of the reference's own HeterogeneousNextTokenChooser (CPU, fp32),
    generated by tests/golden/make_chooser_fixture.py.  The same requests through the GPU launch, two consecutive
    calls: filtered sets and greedy ids exact; surviving scores and log-probabilities to fp32 rounding."""
    import os

    from tgis_amd.pb import generate_pb2 as pb
    from tgis_amd.utils.tokens import HeterogeneousNextTokenChooser

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chooser_reference.npz"))
    for name in sorted({k.split(".")[0] for k in z.files}):
        n = len([k for k in z.files if k.startswith(f"{name}.params.")])
        params = [pb.NextTokenChooserParameters.FromString(z[f"{name}.params.{i}"].tobytes()) for i in range(n)]
        ch = HeterogeneousNextTokenChooser.from_pb(params, 2, 2, [True] * n, torch.float32, gpu_device)
        ids = torch.from_numpy(z[f"{name}.ids"]).to(gpu_device)
        for step in range(2):
            next_ids, scores, logprobs = ch(ids, torch.from_numpy(z[f"{name}.{step}.logits"]).to(gpu_device))
            scores, logprobs, next_ids = scores.cpu(), logprobs.cpu(), next_ids.cpu()
            want = torch.from_numpy(z[f"{name}.{step}.scores"])
            assert torch.equal(torch.isinf(scores), torch.isinf(want)), (name, step)
            keep = ~torch.isinf(want)
            assert torch.allclose(scores[keep], want[keep], rtol=1e-6, atol=1e-6)
            assert torch.allclose(logprobs[keep], torch.from_numpy(z[f"{name}.{step}.logprobs"])[keep], rtol=1e-5, atol=1e-5)
            greedy = torch.from_numpy(z[f"{name}.{step}.greedy_rows"])
            assert torch.equal(next_ids[greedy], torch.from_numpy(z[f"{name}.{step}.next_ids"])[greedy])
            for b in range(n):  # sampled rows: drawn from inside the reference's filtered set
                assert not torch.isinf(want[b, int(next_ids[b])])
