"""Host cost of the segmented decode graph with real RCCL calls at the seams, on ONE GPU.

A world-size-1 `nccl` group makes every all-reduce a real RCCL launch (a no-op reduction); the row-parallel linears
are patched to issue it, so the cfg3 step has the 64 seams of a tensor-parallel step while every kernel keeps its
single-GPU shape.  Compares: one full graph (no collectives), segments + RCCL seams, eager + RCCL.
With --tp N the model is built as rank 0 of an N-way shard (its kernels have the per-rank shapes of TP=N, its KV pages
the rank's heads) and every collective of the real TP code path is routed to the world-size-1 group: the GPU and host
work of one rank of a TP=N step, minus the wire time of the collectives.
    python tools/tp_segments_rccl1.py [--steps 32] [--tp 8]"""
import argparse
import os
import sys
import time

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [root, os.path.join(root, "text-generation-inference_amd")]
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--ctx", type=int, default=1024)
    ap.add_argument("--tp", type=int, default=1)
    ap.add_argument("--config", default="llama2-7b-gptq")
    args = ap.parse_args()
    if args.tp > 1:
        os.environ.update(WORLD_SIZE=str(args.tp), RANK="0", TGIS_ALLOW_SHARED_GPU="1")
    import bench
    from tgis_amd.inference_engine.synthetic import InferenceEngine, llama_tensors
    from tgis_amd.models.custom_modeling.flash_llama_modeling import LlamaConfig
    from tgis_amd.models.flash_causal_lm import FlashCausalLM
    from tgis_amd.testing import SyntheticTokenizer, make_batch_pb
    from tgis_amd.utils import layers
    from tgis_amd.utils.graph_segments import collective
    from tgis_amd.utils.kv_cache import PagedKVCache

    torch.cuda.set_device(0)
    torch.distributed.init_process_group("nccl", world_size=1, rank=0)
    pg = torch.distributed.group.WORLD
    if args.tp > 1:
        from tgis_amd.inference_engine import synthetic
        from tgis_amd.utils.dist import FakeGroup

        shard_group = FakeGroup(0, args.tp)
        synthetic.initialize_torch_distributed = lambda world, rank: shard_group
        real_ar, real_ag = torch.distributed.all_reduce, torch.distributed.all_gather_into_tensor
        torch.distributed.all_reduce = lambda t, group=None, **k: real_ar(t, group=pg, **k)
        # rank 0's block of the gathered logits + a launch of the same kind
        torch.distributed.all_gather_into_tensor = lambda o, i, group=None, **k: real_ag(o[:i.shape[0]], i, group=pg, **k)
    kw, quantize, dtype_s, _, _ = bench.CONFIGS[args.config]
    bigcode = "n_inner" in kw
    if bigcode:
        from tgis_amd.inference_engine.synthetic import BigCodeConfig, bigcode_tensors
    cfg = BigCodeConfig(**kw) if bigcode else LlamaConfig(**kw)
    dtype = getattr(torch, dtype_s)
    B, K = args.batch, args.steps
    L_in = args.ctx - K
    dev = torch.device("cuda", 0)
    tok = SyntheticTokenizer(cfg.vocab_size)
    tensors = (bigcode_tensors(cfg, seed=1234, device=dev, dtype=dtype) if bigcode
               else llama_tensors(cfg, quantize, seed=1234, device=dev, dtype=dtype))
    eng = InferenceEngine(tensors, cfg, dtype, quantize, tokenizer=tok)
    del tensors
    pages = B * PagedKVCache.pages_for(args.ctx + 3 * K + 16) + 8
    lm = FlashCausalLM("synthetic", None, "synthetic", dtype, quantize, engine=eng, kv_cache_pages=pages)
    plain = layers.TensorParallelRowLinear.forward

    def with_reduce(self, x, **k):
        k.pop("partial", None)
        out = self.linear.forward(x, **k)
        collective(lambda: torch.distributed.all_reduce(out, group=pg))
        return out

    def run(label, mode, forward):
        layers.TensorParallelRowLinear.forward = forward
        layers.TensorParallelRowLinear.__call__ = forward
        lm._graphs.clear()
        lm.graph_mode, lm.use_graphs, lm.tp_world = mode, mode is not None, max(args.tp, 2 if forward is not plain else 1)
        pb = make_batch_pb([L_in] * B, max_new=3 * K + 8)
        with lm.context_manager():
            batch, errs = lm.batch_type.from_pb(pb, tok, lm.dtype, lm.device, lm.word_embeddings, None, True)
            lm.generate_token(batch, first=True)
            for _ in range(3):
                lm.generate_token(batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(K):
                lm.generate_token(batch)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / K * 1e3
            # host-only cost of issuing the step: enqueue without waiting (the D2H of the token ids still syncs)
            seg = [getattr(g.graph, "num_segments", 1) for g in lm._graphs.values() if g.graph is not None]
            batch.release()
        print(f"{label.format(mode=lm.graph_mode):34s} {ms:7.3f} ms/step  {B / ms * 1e3:9.1f} tok/s  graphs={lm.use_graphs} segments={seg}", flush=True)

    if args.tp > 1:  # the TP layers issue the collectives themselves
        run(f"tp{args.tp} rank: one graph, RCCL inside", "full", plain)
        run(f"tp{args.tp} rank: segments, RCCL seams", "segments", plain)
        run(f"tp{args.tp} rank: eager", None, plain)
        lm.process_group = pg  # the probe of the default mode against a real RCCL group
        run(f"tp{args.tp} rank: auto (probe -> {{mode}})", "auto", plain)
    else:
        run("full graph, no collectives", "full", plain)
        run("segments, RCCL(world=1) seams", "segments", with_reduce)
        run("eager, RCCL(world=1) all-reduces", None, with_reduce)
        run("eager, no collectives", None, plain)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
