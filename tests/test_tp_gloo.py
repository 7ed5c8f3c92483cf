"""CPU, world_size 2 over gloo: tensor-parallel host logic (Weights slicing, column/row parallel linears incl. the
GPTQ sharding rules, vocab-parallel embedding and head, all-reduce / all-gather placement) gives the same logits as
the unsharded oracle.  The kernels are replaced by the oracle through tests/cpu_backend.py — this test covers what
runs ABOVE the kernels when bench.py is launched with --gpus N."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from oracle.llama_ref import LlamaRef
from oracle.tiny_models import TinyLlamaConfig, tiny_llama_tensors


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, quantize, ret):
    import pytest as _pytest

    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "text-generation-inference_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from tests import cpu_backend

    mpatch = _pytest.MonkeyPatch()
    cpu_backend.install(mpatch)
    from tgis_amd.models.custom_modeling.flash_llama_modeling import FlashLlamaForCausalLM, KVArgs, LlamaConfig
    from tgis_amd.utils.dist import initialize_torch_distributed
    from tgis_amd.utils.kv_cache import PagedKVCache
    from tgis_amd.utils.weights import DictWeights

    torch.set_num_threads(2)
    cfg = TinyLlamaConfig()
    tensors = tiny_llama_tensors(cfg, seed=21, quantize=quantize, groupsize=64)
    pg = initialize_torch_distributed(world, rank)
    pcfg = LlamaConfig(**cfg.to_dict())
    pcfg.quantize = quantize
    weights = DictWeights(tensors, torch.device("cpu"), torch.float16, pg)
    weights.gptq_bits, weights.gptq_groupsize = 4, 64
    model = FlashLlamaForCausalLM(pcfg, weights)
    # sharding facts
    attn = model.model.layers[0].self_attn
    assert attn.num_heads == cfg.num_attention_heads // world
    assert attn.num_key_value_heads == cfg.num_key_value_heads // world
    assert model.lm_head.should_gather == (world > 1)
    # forward: 2 sequences prefill, then one decode token each
    lens = [9, 4]
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(3, cfg.vocab_size, (sum(lens),), generator=g)
    cache = PagedKVCache(cfg.num_hidden_layers, attn.num_key_value_heads, attn.head_size, 8, torch.float16, "cpu")
    bt = torch.tensor([[0, 1], [2, 3]], dtype=torch.int32)
    pos = torch.cat([torch.arange(l) for l in lens]).int()
    slots = torch.cat([bt[b, torch.arange(l) // 32].long() * 32 + torch.arange(l) % 32 for b, l in enumerate(lens)]).int()
    cu = torch.tensor([0, 9, 13], dtype=torch.int32)
    kv = KVArgs(cache, bt, torch.tensor(lens, dtype=torch.int32), slots, max(lens), max(lens), 1)
    logits = model.forward(ids, pos, cu, 64, None, kv, lm_head_indices=(cu[1:] - 1).long())
    nxt = logits.argmax(-1)
    pos1 = torch.tensor(lens, dtype=torch.int32)
    slots1 = torch.empty(2, dtype=torch.int32)
    ctx1 = torch.empty(2, dtype=torch.int32)
    from tgis_amd import native

    native.decode_slots(pos1, bt, slots1, ctx1)
    kv1 = KVArgs(cache, bt, ctx1, slots1, 1, 64, 1)
    logits1 = model.forward(nxt, pos1, torch.arange(3, dtype=torch.int32), 64, None, kv1)
    ret[rank] = (logits.float().clone(), logits1.float().clone(), nxt.clone())
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


@pytest.mark.parametrize("quantize", [None, "gptq"])
def test_tp2_equals_unsharded_oracle(quantize):
    mgr = mp.get_context("spawn").Manager()  # never fork a process that has run gRPC (or CUDA) threads
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, quantize, ret), nprocs=2, join=True)
    l0, d0, n0 = ret[0]
    l1, d1, n1 = ret[1]
    # every rank must hold identical logits (the router takes any shard's reply, sharded_client.rs:38-48)
    assert torch.equal(l0, l1) and torch.equal(d0, d1) and torch.equal(n0, n1)
    cfg = TinyLlamaConfig()
    tensors = tiny_llama_tensors(cfg, seed=21, quantize=quantize, groupsize=64)
    ref = LlamaRef(cfg, tensors, quantize=quantize, groupsize=64)
    lens = [9, 4]
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(3, cfg.vocab_size, (sum(lens),), generator=g)
    prompts = [ids[:9].tolist(), ids[9:].tolist()]
    want = ref.generate_greedy(prompts, 2, forced=[n0.tolist()])
    # activations are rounded to fp16 between ops in the model under test; logits scale ~60
    assert (l0 - want[0]["logits"]).abs().max() < 0.5
    assert (d0 - want[1]["logits"]).abs().max() < 0.5
    assert n0.tolist() == want[0]["token_ids"].tolist()


def test_weights_row_col_sharding_rules():
    """GPTQ bundles per rank follow utils/weights.py:115-201: column = dim-1 shards concatenated over the fused
    prefixes with the full g_idx; row = dim-0 shards of qweight/qzeros/scales and no g_idx for tp > 1."""
    from tgis_amd.utils.dist import FakeGroup
    from tgis_amd.utils.weights import DictWeights

    cfg = TinyLlamaConfig()
    t = tiny_llama_tensors(cfg, seed=1, quantize="gptq", groupsize=64)
    E, I = cfg.hidden_size, cfg.intermediate_size
    for rank in (0, 1):
        w = DictWeights(t, torch.device("cpu"), torch.float16, FakeGroup(rank, 2))
        w.gptq_bits, w.gptq_groupsize = 4, 64
        qw, qz, sc, gi, bits, gs, _ = w.get_multi_weights_col(
            ["model.layers.0.mlp.gate_proj", "model.layers.0.mlp.up_proj"], "gptq", 0)
        assert qw.shape == (E // 8, I) and qz.shape == (E // 64, I // 8) and sc.shape == (E // 64, I)
        assert gi.shape == (E,) and (bits, gs) == (4, 64)
        half = I // 2
        assert torch.equal(qw[:, :half], t["model.layers.0.mlp.gate_proj.qweight"][:, rank * half:(rank + 1) * half])
        assert torch.equal(qw[:, half:], t["model.layers.0.mlp.up_proj.qweight"][:, rank * half:(rank + 1) * half])
        qw, qz, sc, gi, *_ = w.get_multi_weights_row("model.layers.0.mlp.down_proj", "gptq")
        assert qw.shape == (I // 16, E) and qz.shape == (I // 128, E // 8) and gi is None
        assert torch.equal(qw, t["model.layers.0.mlp.down_proj.qweight"][rank * I // 16:(rank + 1) * I // 16])
    with pytest.raises(AssertionError):
        DictWeights({"a.weight": torch.zeros(7, 4)}, "cpu", torch.float16, FakeGroup(0, 2)).get_sharded("a.weight", 0)


@pytest.mark.parametrize("K,gs,world", [(11008, 128, 4), (11008, 128, 8), (448, 64, 2), (512, 64, 2)])
def test_row_parallel_gptq_regroups_misaligned_shards(K, gs, world):
    """Row-parallel GPTQ shards whose rows do not end on group boundaries (llama-7B down_proj at tp=4/8) are served by
    splitting groups into sub-groups with copied scale/zero rows: every rank's dequantised slice must equal the slice
    of the unsharded dequantised matrix, bit for bit."""
    from oracle import ops_ref
    from tgis_amd.utils.dist import FakeGroup
    from tgis_amd.utils.weights import DictWeights

    N = 64
    qw, qz, sc, gi = ops_ref.make_gptq_tensors(K, N, gs, seed=K + world)
    full = ops_ref.gptq_dequant(qw, qz, sc, None, gs)
    t = {"p.qweight": torch.from_numpy(qw), "p.qzeros": torch.from_numpy(qz), "p.scales": torch.from_numpy(sc),
         "p.g_idx": torch.arange(K, dtype=torch.int32) // gs}
    rows = K // world
    for rank in range(world):
        w = DictWeights(t, torch.device("cpu"), torch.float16, FakeGroup(rank, world))
        w.gptq_bits, w.gptq_groupsize = 4, gs
        lqw, lqz, lsc, lgi, bits, lgs, _ = w.get_multi_weights_row("p", "gptq")
        assert lgi is None and lqw.shape[0] * 8 == rows and rows % lgs == 0 and lgs % 8 == 0
        assert lqz.shape[0] == lsc.shape[0] == rows // lgs
        if rows % gs == 0:
            assert lgs == gs, "aligned shards keep the checkpoint's groups"
        local = ops_ref.gptq_dequant(lqw.numpy(), lqz.numpy(), lsc.numpy(), None, lgs)
        assert torch.equal(local, full[rank * rows:(rank + 1) * rows])


def _pages_worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    import types

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "text-generation-inference_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from tgis_amd.models.flash_causal_lm import FlashCausalLM
    from tgis_amd.utils.dist import initialize_torch_distributed

    pg = initialize_torch_distributed(world, rank)
    engine = types.SimpleNamespace(process_group=pg, world_size=world)
    me = types.SimpleNamespace(device=torch.device("cpu"))
    ret[rank] = FlashCausalLM._agree_on_min(me, 1000 - 137 * rank, engine)


def test_tp_ranks_agree_on_the_smallest_page_pool():
    """Every tensor-parallel rank must build the same page pool (the rank with the least free memory decides): with
    lazily grown pages a rank that runs out first would leave a decode step its peers have already entered."""
    world = 2
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_pages_worker, args=(r, world, port, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert dict(ret) == {0: 863, 1: 863}


def _act_order_tensors(K, N, gs, seed):
    from oracle import ops_ref

    qw, qz, sc, gi = ops_ref.make_gptq_tensors(K, N, gs, seed=seed, act_order=True)
    t = {"p.qweight": torch.from_numpy(qw), "p.qzeros": torch.from_numpy(qz), "p.scales": torch.from_numpy(sc),
         "p.g_idx": torch.from_numpy(gi).to(torch.int32)}
    return (qw, qz, sc, gi), t


@pytest.mark.parametrize("K,gs,world", [(1024, 128, 2), (2048, 128, 4), (448, 64, 2)])
def test_act_order_row_shards_carry_their_own_permutation(K, gs, world):
    """Act-order GPTQ under row tensor parallelism (the reference's g_idx fallback, utils/weights.py:150-156,190-196): each
    rank gets its rows sorted by group, runs padded to 32 rows, scales / zeros looked up in the FULL tables.  Every real
    image row must dequantise to exactly the row of the unsharded matrix it gathers, pad rows read a zero activation."""
    from oracle import ops_ref
    from tgis_amd.utils.dist import FakeGroup
    from tgis_amd.utils.weights import DictWeights

    N = 64
    (qw, qz, sc, gi), t = _act_order_tensors(K, N, gs, seed=K + world)
    full = ops_ref.gptq_dequant(qw, qz, sc, gi, gs)  # [K, N], natural row order
    rows = K // world
    for rank in range(world):
        w = DictWeights(t, torch.device("cpu"), torch.float16, FakeGroup(rank, world))
        w.gptq_bits, w.gptq_groupsize = 4, gs
        lqw, lqz, lsc, lgi, bits, lgs, _ = w.get_multi_weights_row("p", "gptq")
        tag, perm, xcols = lgi
        Kp = lqw.shape[0] * 8
        assert tag == "perm" and xcols == rows and perm.shape == (Kp,) and lgs == 32 and Kp % 32 == 0
        assert lqz.shape[0] == lsc.shape[0] == Kp // 32
        real = perm >= 0
        assert int(real.sum()) == rows and sorted(perm[real].tolist()) == list(range(rows)), "a permutation of the shard"
        local = ops_ref.gptq_dequant(lqw.numpy(), lqz.numpy(), lsc.numpy(), None, lgs)
        assert torch.equal(local[real], full[rank * rows + perm[real].long()])
        # pad rows hold zero nibbles: whatever they dequantise to is multiplied by a zero activation
        nib = ((lqw.unsqueeze(1) >> (torch.arange(8, dtype=torch.int32) * 4).view(1, 8, 1)) & 15).reshape(Kp, N)
        assert int(nib[~real].abs().sum()) == 0


def _act_order_worker(rank, world, port, ret):
    import pytest as _pytest

    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "text-generation-inference_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from tests import cpu_backend

    cpu_backend.install(_pytest.MonkeyPatch())
    import types

    from tgis_amd.utils.dist import initialize_torch_distributed
    from tgis_amd.utils.layers import TensorParallelRowLinear
    from tgis_amd.utils.weights import DictWeights

    torch.set_num_threads(2)
    K, N, gs, M = 1024, 96, 128, 5
    _, t = _act_order_tensors(K, N, gs, seed=77)
    pg = initialize_torch_distributed(world, rank)
    w = DictWeights(t, torch.device("cpu"), torch.float16, pg)
    w.gptq_bits, w.gptq_groupsize = 4, gs
    lin = TensorParallelRowLinear.load(types.SimpleNamespace(quantize="gptq"), "p", w, bias=False)
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(M, K, generator=g) * 0.5).half()
    rows = K // world
    y = lin(x[:, rank * rows:(rank + 1) * rows].contiguous())
    ret[rank] = y.float()


def test_act_order_row_parallel_linear_world2_matches_the_unsharded_formula():
    """Two gloo ranks: TensorParallelRowLinear over act-order shards + all-reduce == ops_ref.gptq_linear on the whole matrix."""
    from oracle import ops_ref

    world = 2
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_act_order_worker, args=(r, world, port, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
            assert p.exitcode == 0
        got = dict(ret)
    (qw, qz, sc, gi), _ = _act_order_tensors(1024, 96, 128, seed=77)
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(5, 1024, generator=g) * 0.5).half()
    want = ops_ref.gptq_linear(x, qw, qz, sc, gi, 128, None).float()
    for r in range(world):
        assert torch.allclose(got[r], want, rtol=4e-3, atol=4e-3 * float(want.abs().mean()) + 1e-3), r
