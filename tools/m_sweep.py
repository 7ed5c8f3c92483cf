import os, sys
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
os.environ["TGIS_GPTQ_NOREDUCE"] = "1"
import microbench as mb
for M in (32, 64, 96, 128):
    for (K, N) in [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)]:
        print(f"M={M}: ", end=""); mb.bench_gptq(M, K, N, sets=4)
