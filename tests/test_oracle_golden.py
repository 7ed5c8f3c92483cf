"""CPU tests (-m "not gpu"): the oracle (oracle/*.py) is pinned against outputs of the REFERENCE's own CPU causal_lm
path captured in tests/golden/ — token ids bit-exact, logits/logprobs to fp32 round-off — and against the
reference's own GPTQ packer.  These are what make the oracle a trustworthy checker for the -m gpu parity tests."""
import numpy as np
import pytest
import torch

from oracle import ops_ref
from oracle.llama_ref import LlamaRef
from oracle.tiny_models import TinyLlamaConfig, tiny_llama_tensors
from tests.fixture_utils import load_fixture

# both sides are fp32 on CPU; they differ by summation order only (HF attention/MLP vs ours)
LOGIT_ATOL = 2e-3


def _ref_model(meta):
    cfg = TinyLlamaConfig(**{k: v for k, v in meta["config"].items() if k in (
        "vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
        "num_key_value_heads", "rms_norm_eps", "rope_theta", "max_position_embeddings")})
    tensors = tiny_llama_tensors(cfg, seed=meta["seed"], quantize=meta["quantize"], groupsize=meta["groupsize"])
    return cfg, LlamaRef(cfg, tensors, quantize=meta["quantize"], groupsize=meta["groupsize"])


@pytest.mark.parametrize("variant", ["dense", "gptq"])
@pytest.mark.parametrize("scenario", ["equal", "ragged"])
def test_oracle_matches_reference_generate(variant, scenario):
    meta, steps = load_fixture(f"llama_{variant}_{scenario}")
    cfg, ref = _ref_model(meta)
    forced = [s["ids"].tolist() for s in steps]
    got = ref.generate_greedy(meta["prompts"], len(steps), forced=forced)
    for i, (g, w) in enumerate(zip(got, steps)):
        assert g["token_ids"].tolist() == w["ids"].tolist(), f"step {i}: token ids differ from the reference"
        np.testing.assert_allclose(g["logits"].numpy(), w["logits"], atol=LOGIT_ATOL, rtol=1e-4,
                                   err_msg=f"step {i} logits")
        np.testing.assert_allclose(g["logprobs"].numpy(), w["logprobs"], atol=1e-4, err_msg=f"step {i} logprobs")
        if meta.get("ranks"):
            assert w["ranks"].tolist() == [1] * len(w["ids"])  # greedy token is rank 1 in the reference output
    # logical slot indices follow the reference bookkeeping: after prefill cu = cumsum(len) + arange, then +arange
    lens = np.array([len(p) for p in meta["prompts"]])
    for i, g in enumerate(got):
        want = np.cumsum(lens + i + 1) - 1
        assert g["slot_indices"].tolist() == want.tolist()


@pytest.mark.parametrize("variant", ["dense", "gptq"])
def test_oracle_matches_reference_continuous_batching(variant):
    """prefill A, 2 decodes, prefill B, concatenate, 2 decodes, prune id 0, 2 decodes: per-request token streams of
    the reference (which re-pads / concatenates KV) equal the oracle's independent per-sequence streams."""
    meta, steps = load_fixture(f"llama_{variant}_continuous")
    cfg, ref = _ref_model(meta)
    prompts = {0: meta["prompts_a"][0], 1: meta["prompts_a"][1], 2: meta["prompts_b"][0]}
    want_stream = {0: [], 1: [], 2: []}
    want_logits = {0: [], 1: [], 2: []}
    for s in steps:
        for rid, tok, lg in zip(s["request_ids"], s["ids"], s["logits"]):
            want_stream[int(rid)].append(int(tok))
            want_logits[int(rid)].append(lg)
    for rid, p in prompts.items():
        n = len(want_stream[rid])
        got = ref.generate_greedy([p], n, forced=[[t] for t in want_stream[rid]])
        assert [int(g["token_ids"][0]) for g in got] == want_stream[rid], f"request {rid}"
        for i, g in enumerate(got):
            np.testing.assert_allclose(g["logits"][0].numpy(), want_logits[rid][i], atol=LOGIT_ATOL, rtol=1e-4)


def test_gptq_pack_and_dequant_match_reference_packer():
    """oracle.ops_ref.gptq_pack == QuantLinear.pack (utils/gptq/quant_linear.py:290-345) bit for bit, and
    gptq_dequant inverts it with the matmul_248_kernel formula (:184-192)."""
    z = np.load("tests/golden/gptq_pack_reference.npz")
    qw, qz = ops_ref.gptq_pack(z["intw"], z["zeros"])
    assert np.array_equal(qw, z["qweight"]) and np.array_equal(qz, z["qzeros"])
    w = ops_ref.gptq_dequant(z["qweight"], z["qzeros"], z["ref_scales"], None, 64)
    np.testing.assert_allclose(w.numpy(), z["dequant"], atol=1e-7)
    x = torch.randn(5, 128, generator=torch.Generator().manual_seed(0))
    y = ops_ref.gptq_linear(x, z["qweight"], z["qzeros"], z["ref_scales"], np.arange(128) // 64, 64)
    np.testing.assert_allclose(y.numpy(), x.numpy() @ z["dequant"], atol=1e-5)


def test_kv_page_layout_roundtrip():
    """kv_page_unpack is the inverse of the documented page layout on a synthetic page."""
    Hkv, D = 2, 64
    K = torch.arange(32 * Hkv * D, dtype=torch.float32).view(32, Hkv, D)
    V = -K
    kp = torch.zeros(1, Hkv, 32 * D)
    vp = torch.zeros(1, Hkv, 32 * D)
    for tok in range(32):
        for h in range(Hkv):
            for d in range(D):
                off = ((((tok >> 4) * (D >> 3) + (d >> 3)) * 16 + (tok & 15)) << 3) + (d & 7)
                kp[0, h, off] = K[tok, h, d]
                i = tok & 15
                col = (i >> 2) * 8 + (tok >> 4) * 4 + (i & 3)
                vp[0, h, ((col >> 3) * D + d) * 8 + (col & 7)] = V[tok, h, d]
    K2, V2 = ops_ref.kv_page_unpack(kp, vp, 0, Hkv, D)
    assert torch.equal(K2, K) and torch.equal(V2, V)


@pytest.mark.parametrize("scenario", ["equal", "ragged"])
def test_santacoder_oracle_matches_reference_generate(scenario):
    """GPT-BigCode (MQA, learned positions, LayerNorm, tanh-GELU, tied head) through the reference's CPU path."""
    from oracle.santacoder_ref import SantacoderRef
    from oracle.tiny_models import TinyBigCodeConfig, tiny_bigcode_tensors

    meta, steps = load_fixture(f"bigcode_{scenario}")
    cfg = TinyBigCodeConfig()
    ref = SantacoderRef(cfg, tiny_bigcode_tensors(cfg, seed=meta["seed"], embed_scale=meta["embed_scale"]))
    got = ref.generate_greedy(meta["prompts"], len(steps), forced=[s["ids"].tolist() for s in steps])
    for i, (g, w) in enumerate(zip(got, steps)):
        np.testing.assert_allclose(g["logits"].numpy(), w["logits"], atol=LOGIT_ATOL, rtol=1e-4, err_msg=f"step {i}")
        assert g["token_ids"].tolist() == w["ids"].tolist(), f"step {i}"
        np.testing.assert_allclose(g["logprobs"].numpy(), w["logprobs"], atol=1e-4)


@pytest.mark.parametrize("name", ["gpt2_equal", "gpt2_padded"])
def test_gpt2_oracle_reproduces_reference_fixture(name):
    """oracle/gpt2_ref.py (BASELINE config 1's family) against the reference's padded causal_lm path: the var-len
    restatement must give the logits the left-padded batch gave (a truncated request contributes its last
    `truncate_to` tokens)."""
    from oracle.gpt2_ref import GPT2Ref
    from oracle.tiny_models import TinyGPT2Config, tiny_gpt2_tensors

    meta, steps = load_fixture(name)
    cfg = TinyGPT2Config()
    ref = GPT2Ref(cfg, tiny_gpt2_tensors(cfg, seed=meta["seed"], embed_scale=meta["embed_scale"]))
    prompts = meta["prompts"]
    if meta.get("truncate_to"):
        prompts = [p if k is None else p[-k:] for p, k in zip(prompts, meta["truncate_to"])]
    n = min(len(steps), 8)
    want = ref.generate_greedy(prompts, n, forced=[s["ids"].tolist() for s in steps[:n]])
    for i in range(n):
        np.testing.assert_allclose(want[i]["logits"].numpy(), steps[i]["logits"], atol=2e-3, rtol=0)
        assert want[i]["token_ids"].tolist() == steps[i]["ids"].tolist()
        np.testing.assert_allclose(want[i]["logprobs"].numpy(), steps[i]["logprobs"], atol=2e-3, rtol=0)
