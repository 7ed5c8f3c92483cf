import os, sys
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
import microbench as mb
for (B, H, Hkv, D, ctx) in [(32, 16, 16, 128, 1024), (32, 8, 8, 128, 1024), (32, 4, 4, 128, 1024), (64, 8, 1, 128, 2048), (32, 12, 1, 128, 4096), (1, 32, 32, 128, 1024), (8, 32, 32, 128, 1024)]:
    for ns in (None, 1, 2, 4, 8, 16):
        try:
            mb.bench_attn(B, H, Hkv, D, ctx, ns=ns)
        except Exception as e:
            print("ns", ns, "failed", str(e)[:80])
