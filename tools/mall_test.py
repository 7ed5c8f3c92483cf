"""GEMM time with the weights resident in the Infinity Cache (one weight set re-used) vs streamed from HBM (8 sets)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
os.environ["TGIS_GPTQ_NOREDUCE"] = "1"
import microbench as mb
for (K, N) in [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)]:
    for sets in (8, 1):
        print(f"sets={sets}: ", end="")
        mb.bench_gptq(32, K, N, sets=sets)
