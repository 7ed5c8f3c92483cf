"""Prompt-prefix cache: file conventions, validation rules and LRU accounting of the
reference's prompt_cache.py, exercised on CPU tensors; plus the prefix path of FlashCausalLMBatch.from_pb."""
import os
import threading

import pytest
import torch

from tgis_amd.prompt_cache import PrefixCache, PrefixNotFound, max_prompt_prefix_length


@pytest.fixture()
def store(tmp_path):
    def save(prefix_id, tensor, name="decoder.pt"):
        d = tmp_path / prefix_id
        d.mkdir(parents=True, exist_ok=True)
        torch.save(tensor, d / name)
    return tmp_path, save


def _cache(path, **kw):
    args = dict(device=torch.device("cpu"), dtype=torch.float16, max_length=8, hidden_size=16, store=path, budget_mb=1)
    args.update(kw)
    return PrefixCache(**args)


def test_decoder_pt_is_loaded_cast_and_cached(store):
    path, save = store
    t = torch.randn(3, 16)
    save("team/alpha", t)
    c = _cache(path)
    got = c.get("team/alpha")
    assert got.dtype == torch.float16 and got.shape == (3, 16) and not got.requires_grad
    assert torch.equal(got, t.half())
    os.remove(path / "team/alpha" / "decoder.pt")
    assert c.get("team/alpha") is got, "second lookup must come from the cache, not the disk"
    assert len(c) == 1 and c.size_mb == pytest.approx(512 / 2 ** 20)  # 96 bytes round up to one 512-byte unit


def test_peft_adapter_layouts(store):
    path, save = store
    t = torch.randn(2, 16)
    save("peft_bin", {"prompt_embeddings": t}, name="adapter_model.bin")
    from safetensors.torch import save_file
    (path / "peft_st").mkdir()
    save_file({"prompt_embeddings": t}, str(path / "peft_st" / "adapter_model.safetensors"))
    c = _cache(path)
    assert torch.equal(c.get("peft_bin"), t.half())
    assert torch.equal(c.get("peft_st"), t.half())


@pytest.mark.parametrize("bad_id", ["../escape", "a b", "x;y", "", "ok/../../up"])
def test_bad_ids_are_rejected(store, bad_id):
    path, _ = store
    with pytest.raises(Exception) as e:
        _cache(path).get(bad_id)
    assert not isinstance(e.value, PrefixNotFound) or bad_id == ""


def test_validation_rules(store):
    path, save = store
    c = _cache(path)
    with pytest.raises(PrefixNotFound):
        c.get("missing")
    save("too_long", torch.randn(9, 16))
    save("empty", torch.randn(0, 16))
    save("wrong_dim", torch.randn(2, 8))
    save("not2d", torch.randn(2, 4, 4))
    save("inf", torch.full((2, 16), float("inf")))
    save("overflow", torch.full((2, 16), 1e6))  # finite in fp32, inf in fp16
    save("enc", torch.randn(2, 16), name="encoder.pt")
    for pid, frag in [("too_long", "length"), ("empty", "length"), ("wrong_dim", "does not match"), ("not2d", "Invalid"),
                      ("inf", "non-finite"), ("overflow", "after conversion"), ("enc", "encoder")]:
        with pytest.raises(ValueError, match=frag):
            c.get(pid)
    assert len(c) == 0


def test_lru_eviction_by_size(store):
    path, save = store
    rows = 8
    c = _cache(path, max_length=rows, hidden_size=16384, budget_mb=1)  # one entry = 8*16384*2 B = 0.25 MiB
    for i in range(5):
        save(f"p{i}", torch.randn(rows, 16384))
    for i in range(4):
        c.get(f"p{i}")
    assert c.keys() == ["p0", "p1", "p2", "p3"] and c.size_mb == pytest.approx(1.0)
    c.get("p0")                      # refresh p0: p1 is now the oldest
    c.get("p4")                      # evicts p1
    assert c.keys() == ["p2", "p3", "p0", "p4"] and c.size_mb == pytest.approx(1.0)
    big = _cache(path, max_length=rows, hidden_size=16384, budget_mb=0)
    with pytest.raises(ValueError, match="exceeds"):
        big.get("p0")


def test_return_zero_for_non_reducing_ranks(store):
    path, save = store
    save("z", torch.randn(4, 16))
    got = _cache(path, return_zero=True).get("z")
    assert got.shape == (4, 16) and not got.any()


def test_concurrent_first_loads_share_one_entry(store):
    path, save = store
    save("shared", torch.randn(4, 16))
    c = _cache(path)
    out = []
    ts = [threading.Thread(target=lambda: out.append(c.get("shared"))) for _ in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert len(c) == 1 and all(o is c.get("shared") for o in out)


def test_max_prompt_prefix_length(monkeypatch):
    monkeypatch.delenv("MAX_PROMPT_PREFIX_LENGTH", raising=False)
    assert max_prompt_prefix_length(2049) == 1025
    monkeypatch.setenv("MAX_PROMPT_PREFIX_LENGTH", "100")
    assert max_prompt_prefix_length(2048) == 100
    monkeypatch.setenv("MAX_PROMPT_PREFIX_LENGTH", "2048")
    with pytest.raises(ValueError):
        max_prompt_prefix_length(2048)
    monkeypatch.setenv("MAX_PROMPT_PREFIX_LENGTH", "abc")
    with pytest.raises(ValueError):
        max_prompt_prefix_length(2048)


def test_from_pb_injects_prefix_rows_and_reports_missing_ids(store):
    """flash_causal_lm.py:97-107,157-168: input_length grows by the prefix, the first rows of the request's input
    embeddings are the prefix, a failed lookup drops the request with an error."""
    from tgis_amd.models.flash_causal_lm import FlashCausalLMBatch
    from tgis_amd.testing import SyntheticTokenizer, make_batch_pb

    path, save = store
    pre = torch.randn(3, 16)
    save("soft", pre)
    cache = _cache(path)
    tok = SyntheticTokenizer(64)
    pb = make_batch_pb([5, 4, 6], max_new=4)
    pb.requests[0].prefix_id = "soft"
    pb.requests[1].prefix_id = "nope"
    table = torch.randn(64, 16).half()
    batch, errors = FlashCausalLMBatch.from_pb(pb, tok, torch.float16, torch.device("cpu"), lambda ids: table[ids].clone(),
                                               cache, True)
    assert [e.request_id for e in errors] == [pb.requests[1].id]
    assert [r.id for r in batch.requests] == [pb.requests[0].id, pb.requests[2].id]
    assert batch.input_lengths == [5 + 3, 6] and batch.input_ids is None
    assert torch.equal(batch.inputs_embeds[:3], pre.half())
    ids0 = batch.all_input_ids_tensor[0, 3:8]
    assert torch.equal(batch.inputs_embeds[3:8], table[ids0])
    assert (batch.all_input_ids_tensor[0, :3] == tok.pad_token_id).all(), "prefix positions hold pad ids"
    batch.release()
