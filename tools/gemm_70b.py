"""GPTQ GEMM time at the Llama-2-70B shapes (full and TP=8 shards) for M = 32 / 64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
import microbench as mb
for M in (32, 64):
    for K, N in ((8192, 10240), (8192, 8192), (8192, 57344), (28672, 8192), (8192, 1280), (1024, 8192), (8192, 7168), (3584, 8192)):
        mb.bench_gptq(M, K, N, sets=3)
for M in (32,):
    for K, N in ((4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)):
        mb.bench_gptq(M, K, N, sets=8)
