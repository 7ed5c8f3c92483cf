"""cProfile of eager (graph-less) decode steps: where the host time per step goes (tp > 1 runs eagerly)."""
import cProfile, os, pstats, sys
os.environ["TGIS_DISABLE_GRAPHS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
import torch
import bench
from tgis_amd.inference_engine.synthetic import InferenceEngine, llama_tensors
from tgis_amd.models.custom_modeling.flash_llama_modeling import LlamaConfig
from tgis_amd.models.flash_causal_lm import FlashCausalLM
from tgis_amd.testing import SyntheticTokenizer, make_batch_pb

kw, quantize, dtype_s, _, _ = bench.CONFIGS["llama2-7b-gptq"]
cfg = LlamaConfig(**kw)
dev = torch.device("cuda:0")
tok = SyntheticTokenizer(cfg.vocab_size)
eng = InferenceEngine(llama_tensors(cfg, quantize, seed=1, device=dev), cfg, torch.float16, quantize, tokenizer=tok)
lm = FlashCausalLM("synthetic", None, "synthetic", torch.float16, quantize, engine=eng, kv_cache_pages=200)
with lm.context_manager():
    batch, _ = lm.batch_type.from_pb(make_batch_pb([64] * 32, max_new=64), tok, lm.dtype, lm.device, lm.word_embeddings, None, True)
    lm.generate_token(batch, first=True)
    for _ in range(3):
        lm.generate_token(batch)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        lm.generate_token(batch)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
