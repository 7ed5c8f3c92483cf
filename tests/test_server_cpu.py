"""CPU: the gRPC servicer's batch lifecycle (Prefill / NextToken / prune / concatenate / healthcheck / errors) over a
real grpc.aio server on a unix socket, with a stub Model so that no kernel runs.  Mirrors what the router does
(router/src/batcher.rs:399-570) and the contracts of SURVEY.md §8b boundary #1."""
import asyncio
import tempfile

import grpc
import pytest
import torch

from tests.fixture_utils import FixtureTokenizer, prompt_text
from tgis_amd.cache import Cache
from tgis_amd.models.flash_causal_lm import FlashCausalLMBatch
from tgis_amd.pb import generate_pb2 as pb
from tgis_amd.pb import generate_pb2_grpc
from tgis_amd.server import HEALTHCHECK_BATCH_ID, MemoryScalingModel, TextGenerationService
from tgis_amd.utils.kv_cache import PagedKVCache
from tgis_amd.utils.token_types import TokenInfo


class StubModel:
    """Produces token id = request id * 10 + step; owns a real page pool so page accounting is exercised."""

    def __init__(self, pages=8):
        self.tokenizer = FixtureTokenizer(256)
        self.dtype = torch.float16
        self.device = torch.device("cpu")
        self.word_embeddings = None
        self.prefix_cache = None
        self.use_position_ids = True
        self.context_manager = torch.inference_mode
        self.batch_type = FlashCausalLMBatch
        self.kv_cache = PagedKVCache(1, 1, 64, pages, torch.float16, "cpu")
        self.steps = {}

    def generate_token(self, batch, first=False, for_concat=False):
        if first:
            batch.allocate_pages(self.kv_cache)
            batch.cu_seqlens_q = torch.arange(len(batch) + 1, dtype=torch.int32)
            batch.position_ids = torch.tensor(batch.input_lengths)
            batch.input_ids = torch.zeros(len(batch), dtype=torch.int64)
        toks = []
        for r in batch.requests:
            n = self.steps.get(r.id, 0)
            self.steps[r.id] = n + 1
            toks.append(TokenInfo(request_id=r.id, token_id=r.id * 10 + n))
        batch.cu_seqlens.add_(batch.cu_seqlens_q)
        return toks, None, [], 123


def _req(i, n=3, max_new=5):
    return pb.Request(id=i, inputs=prompt_text([7] * n), input_length=n, max_output_length=max_new)


def _cached(batch_id, completed):
    cb = pb.CachedBatch(batch_id=batch_id)
    cb.status.completed_ids.extend(completed)
    return cb


def test_servicer_lifecycle():
    async def run():
        model = StubModel()
        with tempfile.TemporaryDirectory() as d:
            url = f"unix://{d}/shard-0"
            server = grpc.aio.server()
            svc = TextGenerationService(model, Cache(), [url], MemoryScalingModel(1000))
            generate_pb2_grpc.add_TextGenerationServiceServicer_to_server(svc, server)
            server.add_insecure_port(url)
            await server.start()
            async with grpc.aio.insecure_channel(url) as ch:
                stub = generate_pb2_grpc.TextGenerationServiceStub(ch)
                assert list((await stub.ServiceDiscovery(pb.ServiceDiscoveryRequest())).urls) == [url]
                info = await stub.ModelInfo(pb.ModelInfoRequest())
                assert info.batch_padding is False and info.eos_token == 2
                assert info.memory_scaling_model.weight_limit == 1000
                await stub.Health(pb.HealthRequest())
                # health probe batch: generates but must not be cached, pages returned
                r = await stub.Prefill(pb.PrefillRequest(batch=pb.Batch(id=HEALTHCHECK_BATCH_ID, requests=[_req(99)])))
                assert len(r.result.output_tokens) == 1 and len(svc.cache) == 0
                assert model.kv_cache.free_pages == 8
                # batch 1 (requests 0,1)
                r = await stub.Prefill(pb.PrefillRequest(batch=pb.Batch(id=1, requests=[_req(0), _req(1)])))
                assert [t.request_id for t in r.result.output_tokens] == [0, 1] and r.result.batch_id == 1
                assert r.result.forward_time_ns == 123 and svc.cache.keys() == [1]
                r = await stub.NextToken(pb.NextTokenRequest(batches=[_cached(1, [])]))
                assert [t.token_id for t in r.result.output_tokens] == [1, 11]
                # add-on batch 2 (request 2), then NextToken over both -> concatenated under batch id 1
                r = await stub.Prefill(pb.PrefillRequest(batch=pb.Batch(id=2, requests=[_req(2)])))
                assert sorted(svc.cache.keys()) == [1, 2]
                r = await stub.NextToken(pb.NextTokenRequest(batches=[_cached(1, []), _cached(2, [])]))
                assert [t.request_id for t in r.result.output_tokens] == [0, 1, 2] and r.result.batch_id == 1
                assert svc.cache.keys() == [1]
                # request 0 completed -> pruned on the next call; then whole batch finished (no status)
                r = await stub.NextToken(pb.NextTokenRequest(batches=[_cached(1, [0])]))
                assert [t.request_id for t in r.result.output_tokens] == [1, 2]
                r = await stub.NextToken(pb.NextTokenRequest(batches=[pb.CachedBatch(batch_id=1)]))
                assert not r.HasField("result") and len(svc.cache) == 0 and model.kv_cache.free_pages == 8
                # unknown batch id -> error; to_prune path of Prefill; PrefixLookup NOT_FOUND; ClearCache frees pages
                with pytest.raises(grpc.aio.AioRpcError):
                    await stub.NextToken(pb.NextTokenRequest(batches=[_cached(77, [])]))
                await stub.Prefill(pb.PrefillRequest(batch=pb.Batch(id=3, requests=[_req(5), _req(6)])))
                await stub.Prefill(pb.PrefillRequest(batch=pb.Batch(id=4, requests=[_req(7)]), to_prune=[_cached(3, [5])]))
                assert sorted(svc.cache.keys()) == [3, 4] and len(svc.cache.cache[3]) == 1
                with pytest.raises(grpc.aio.AioRpcError) as e:
                    await stub.PrefixLookup(pb.PrefixLookupRequest(prefix_id="nope"))
                assert e.value.code() == grpc.StatusCode.NOT_FOUND
                # pool exhaustion maps to RESOURCE_EXHAUSTED like a CUDA OOM (server.py:48-51)
                with pytest.raises(grpc.aio.AioRpcError) as e:
                    # (pages are taken as tokens arrive: a long PROMPT exhausts the 8-page pool, a large max_new does not)
                    await stub.Prefill(pb.PrefillRequest(batch=pb.Batch(id=9, requests=[_req(8, n=300, max_new=5)])))
                assert e.value.code() == grpc.StatusCode.RESOURCE_EXHAUSTED
                await stub.ClearCache(pb.ClearCacheRequest())
                assert len(svc.cache) == 0 and model.kv_cache.free_pages == 8
                with pytest.raises(grpc.aio.AioRpcError) as e:
                    await stub.PruneBatch(pb.PruneBatchRequest())
                assert e.value.code() == grpc.StatusCode.UNIMPLEMENTED
            await server.stop(0)

    asyncio.run(run())


def test_memory_scaling_model_matches_the_routers_arithmetic():
    """utils/memory_characterizer.py: the paged (token-unit) model admits exactly what the page pool can hold, and the
    manual models reproduce the reference's percent-of-capacity formulas (memory_characterizer.py:110-143)."""
    from tgis_amd.utils.memory_characterizer import MemoryScalingModel

    m = MemoryScalingModel.paged(num_pages=1000, max_batch_size=32, safety_margin=20, prefill_tokens_max=8000)
    pb = m.as_pb()
    assert pb.weight_limit == (1000 - 32) * 32 * 80 // 100 and pb.nexttoken_linear_coef1 == 1.0
    assert pb.nexttoken_linear_coef0 == 0.0 and pb.prefill_quadratic_coef1 == 0.0
    # router: FlashBatch.prefill_weight = tokens * prefill_linear_coef0 <= weight_limit  <=>  tokens <= 8000
    assert 8000 * pb.prefill_linear_coef0 <= pb.weight_limit * (1 + 1e-6) < 8001 * pb.prefill_linear_coef0
    # whatever batch the router admits for decode fits the pool even with every request on a partly filled page
    for B, tokens in ((32, 600), (7, 3000), (1, 24000)):
        if m.next_token_weight(B, tokens, 0) <= m.weight_limit:
            assert B * ((tokens + 31) // 32) <= 1000
    # a prefill can never be cheaper than its own KV
    assert MemoryScalingModel.paged(1000, 32, 20, 10 ** 9).as_pb().prefill_linear_coef0 == 1.0
    q = MemoryScalingModel.manual_quadratic(20, max_seq_len=2048, max_batch_size=8).as_pb()
    assert q.weight_limit == 100 and abs(q.prefill_quadratic_coef1 - 100.0 / (0.8 * 2048 * 2048 * 8)) < 1e-12
    assert abs(q.nexttoken_linear_coef0 - 100.0 / (0.8 * 2048 * 8)) < 1e-9 and q.nexttoken_linear_coef0 == q.nexttoken_linear_coef1
    lin = MemoryScalingModel.manual_linear(0, 100, 4)
    assert lin.prefill_weight(4, 100) == pytest.approx(100.0) and lin.next_token_weight(4, 60, 40) == pytest.approx(100.0)
