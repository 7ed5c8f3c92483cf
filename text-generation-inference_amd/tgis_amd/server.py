"""The shard: gRPC `generate.v1.TextGenerationService` on `unix://{uds_path}-{rank}`, driven by the unmodified Rust
router (drop-in boundary #1).  Mirrors server.py:65-231,251-460 of the reference:

* Prefill: prune `to_prune` batches, build the batch with `batch_type.from_pb`, `generate_token(first=True,
  for_concat=len(cache) > 0)`, cache it unless it is the healthcheck batch (id 2^64-1);
* NextToken: pop the cached batches, prune by `completed_ids` (absent status = whole batch finished), concatenate,
  `generate_token`, cache; empty response when everything finished; leftover cached batches are cleared with a warning;
* ModelInfo: CAUSAL_LM, eos id, batch_padding (False for flash/var-len batches), memory scaling model
  (utils/memory_characterizer.py: token-unit weights of the paged pool + a measured prefill activation fit);
* OOM -> RESOURCE_EXHAUSTED (a KV-pool exhaustion counts as OOM); PrefixLookup -> NOT_FOUND (no prefix store yet).
The shard never decides termination: it keeps generating for a row until the router reports it completed."""
import asyncio
import logging
import os
from pathlib import Path
from typing import List, Optional

import grpc
import torch

from tgis_amd.cache import Cache
from tgis_amd.models.model import Model
from tgis_amd.pb import generate_pb2, generate_pb2_grpc
from tgis_amd.utils.kv_cache import OutOfPages
from tgis_amd.utils.memory_characterizer import ESTIMATE_MEMORY, MemoryScalingModel, characterize_paged

HEALTHCHECK_BATCH_ID = (1 << 64) - 1
COMPACT_BEFORE_PREFILL = os.getenv("COMPACT_BEFORE_PREFILL", "true") != "false"


def log_rpc_handler_errors(func):
    async def wrapped(self, request, context):
        try:
            return await func(self, request, context)
        except grpc.aio.AbortError:
            raise
        except (torch.cuda.OutOfMemoryError, OutOfPages) as e:
            logging.exception(f"{func.__name__} caused GPU OOM error")
            await context.abort(grpc.StatusCode.RESOURCE_EXHAUSTED, str(e))
        except Exception:
            logging.exception(f"{func.__name__} failed")
            raise

    wrapped.__name__ = func.__name__
    return wrapped


class TextGenerationService(generate_pb2_grpc.TextGenerationServiceServicer):
    def __init__(self, model: Model, cache: Cache, server_urls: List[str], memory_scaling_model: MemoryScalingModel):
        self.cache = cache
        self.model = model
        self.server_urls = server_urls
        self.memory_scaling_model = memory_scaling_model

    async def ServiceDiscovery(self, request, context):
        return generate_pb2.ServiceDiscoveryResponse(urls=self.server_urls)

    @log_rpc_handler_errors
    async def ClearCache(self, request, context):
        self.cache.clear()
        return generate_pb2.ClearCacheResponse()

    @log_rpc_handler_errors
    async def ModelInfo(self, request, context):
        tok = self.model.tokenizer
        return generate_pb2.ModelInfoResponse(
            model_type=generate_pb2.ModelInfoResponse.ModelType.Value("CAUSAL_LM"),
            eos_token=getattr(tok, "model_eos_token_id", tok.eos_token_id),
            # var-len (flash) batches -> the router's FlashBatch weights; padded CausalLM batches -> PaddedBatch (:92)
            batch_padding=getattr(self.model, "kv_cache", None) is None,
            memory_scaling_model=self.memory_scaling_model.as_pb())

    @log_rpc_handler_errors
    async def Health(self, request, context):
        if self.model.device.type == "cuda":
            torch.zeros((2, 2)).cuda()
        return generate_pb2.HealthResponse()

    @log_rpc_handler_errors
    async def PrefixLookup(self, request, context):
        prefix_cache = getattr(self.model, "prefix_cache", None)
        prefix = None
        if prefix_cache is not None:
            try:
                prefix = prefix_cache.get(request.prefix_id)
            except Exception:
                prefix = None
        if prefix is None:
            await context.abort(grpc.StatusCode.NOT_FOUND, f"prefix id \"{request.prefix_id}\" not found")
        n = len(prefix) if torch.is_tensor(prefix) else sum(len(t) for t in prefix if t is not None)
        return generate_pb2.PrefixLookupResponse(prefix_length=n)

    def _prune_cached(self, cbatch) -> Optional[object]:
        batch = self.cache.pop(cbatch.batch_id)
        if batch is None:
            raise ValueError(f"Batch ID {cbatch.batch_id} not found in cache.")
        if cbatch.HasField("status"):
            return self.model.batch_type.prune(batch, list(cbatch.status.completed_ids))
        rel = getattr(batch, "release", None)  # absent status: the whole batch is finished
        if rel is not None:
            rel()
        return None

    @log_rpc_handler_errors
    async def Prefill(self, request, context):
        with self.model.context_manager():
            for cbatch in request.to_prune:
                self.cache.set(self._prune_cached(cbatch))
            is_healthcheck = request.batch.id == HEALTHCHECK_BATCH_ID
            if COMPACT_BEFORE_PREFILL and not is_healthcheck:
                self.cache.compact()
            input_token_info = None
            forward_time_ns = 0
            batch, errors = self.model.batch_type.from_pb(
                request.batch, tokenizer=self.model.tokenizer, dtype=self.model.dtype, device=self.model.device,
                embeddings_lookup=self.model.word_embeddings, prefix_cache=self.model.prefix_cache,
                use_position_ids=self.model.use_position_ids)
            batch_id = 0
            if batch is not None:
                for_concat = len(self.cache) > 0
                try:
                    output_tokens, input_token_info, decode_errors, forward_time_ns = self.model.generate_token(
                        batch, first=True, for_concat=for_concat)
                except BaseException:
                    batch.release()
                    raise
                if is_healthcheck:
                    batch.release()
                else:
                    self.cache.set(batch)
                batch_id = batch.get_id()
                errors = (errors or []) + decode_errors
            else:
                output_tokens = []
            return generate_pb2.PrefillResponse(
                result=generate_pb2.GenerateResult(
                    output_tokens=[t.to_pb() for t in output_tokens],
                    errors=[e.to_pb() for e in errors] if errors else None,
                    batch_id=batch_id, forward_time_ns=forward_time_ns),
                input_tokens=[t.to_pb() for t in input_token_info] if input_token_info is not None else None)

    @log_rpc_handler_errors
    async def NextToken(self, request, context):
        if len(request.batches) == 0:
            raise ValueError("Must provide at least one batch")
        with self.model.context_manager():
            batches = []
            for cbatch in request.batches:
                batch = self._prune_cached(cbatch)
                if batch is not None:
                    batches.append(batch)
            if len(self.cache) > 0:
                print(f"WARN: Clearing additional batches found in cache: {self.cache.keys()}")
                self.cache.clear()
            if not batches:
                return generate_pb2.NextTokenResponse()  # all batches finished
            batch = batches[0] if len(batches) == 1 else self.model.batch_type.concatenate(batches)
            del batches
            try:
                output_tokens, _, errors, forward_time_ns = self.model.generate_token(batch)
            except BaseException:
                batch.release()
                raise
            self.cache.set(batch)
            return generate_pb2.NextTokenResponse(
                result=generate_pb2.GenerateResult(
                    output_tokens=[t.to_pb() for t in output_tokens],
                    errors=[e.to_pb() for e in errors] if errors else None,
                    batch_id=batch.get_id(), forward_time_ns=forward_time_ns))


def serve(model_name: str, revision: Optional[str], deployment_framework: str, dtype_str: Optional[str],
          quantize: Optional[str], max_sequence_length: int, max_new_tokens: int, max_batch_size: int,
          batch_safety_margin: int, sharded: bool, uds_path: Path, model: Optional[Model] = None,
          ready_event: Optional[asyncio.Event] = None):
    """Load the model and serve on unix://{uds_path}-{rank} until cancelled (reference server.py:251-460)."""

    async def serve_inner():
        nonlocal model
        if sharded:
            world = int(os.environ["WORLD_SIZE"])
            server_urls = [f"unix://{uds_path}-{rank}" for rank in range(world)]
            local_url = server_urls[int(os.environ["RANK"])]
        else:
            local_url = f"unix://{uds_path}-0"
            server_urls = [local_url]
        if model is None:
            from tgis_amd.models import get_model

            model = get_model(model_name, revision, deployment_framework, dtype_str, quantize, max_sequence_length)
        if getattr(model, "kv_cache", None) is not None:
            # paged KV: token-unit weights, prefill activation peak measured with two synthetic prefills
            msm = characterize_paged(model, max_sequence_length, max_batch_size, batch_safety_margin)
        elif ESTIMATE_MEMORY == "off":
            msm = MemoryScalingModel.disabled()
        else:  # padded batches on a library model: the reference's manual model (memory_characterizer.py:110-128)
            msm = MemoryScalingModel.manual_quadratic(batch_safety_margin, max_sequence_length, max_batch_size)
        print(f"Memory scaling model: {msm.as_pb()}".replace("\n", " "), flush=True)
        server = grpc.aio.server()
        generate_pb2_grpc.add_TextGenerationServiceServicer_to_server(
            TextGenerationService(model, Cache(), server_urls, msm), server)
        server.add_insecure_port(local_url)
        await server.start()
        print(f"Server started at {local_url}", flush=True)
        if ready_event is not None:
            ready_event.set()
        try:
            await server.wait_for_termination()
        except (KeyboardInterrupt, asyncio.CancelledError):
            print("Signal received. Shutting down", flush=True)
            await server.stop(0)

    return serve_inner()
