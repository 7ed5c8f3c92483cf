"""Dense GEMM time at the Starcoder-15B shapes (M = 32) and the 49152-row tied head."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
import microbench as mb
for K, N in ((6144, 6400), (6144, 6144), (6144, 24576), (24576, 6144), (6144, 49152)):
    mb.bench_dense(32, K, N, sets=3)
# TinyLlama-1.1B (cfg2) shapes
for K, N in ((2048, 2560), (2048, 2048), (2048, 11264), (5632, 2048), (2048, 32000)):
    mb.bench_dense(16, K, N, sets=8)
