"""Per-rank GEMM shapes of llama-7B under tp = 2, 4, 8 (column-parallel N/tp, row-parallel K/tp with regrouped
scales where K/tp is not a multiple of the group size): GPU time of each, to spot plans that fall off a cliff."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
import microbench as mb
E, I = 4096, 11008
for tp in (1, 2, 4, 8):
    tot = 0.0
    for name, K, N in [("qkv", E, 3 * E // tp), ("o", E // tp, E), ("gate_up", E, 2 * I // tp), ("down", I // tp, E)]:
        gs = 128 if K % 128 == 0 else math.gcd(128, K)
        print(f"tp={tp} {name:8s} gs={gs:3d} ", end="")
        tot += mb.bench_gptq(32, K, N, gs=gs, sets=4)
    print(f"tp={tp}: GEMMs per layer {tot*1e6:.1f} us")
