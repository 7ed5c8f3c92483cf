"""`if (base >= 128) ns = 1` in tgis_attn_num_splits was decided on ONE case (B 32 x 4 kv heads, ctx 1024).  Sweep
the unsplit-vs-split choice for 128 - 255 (sequence, kv head) groups over the context lengths a server sees."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
import microbench as mb
for B, H, Hkv in ((32, 4, 4), (24, 8, 8), (64, 16, 2), (40, 6, 6)):
    for ctx in (1024, 2048, 4096, 8192, 16384):
        for ns in (None, 1, 2, 4):
            mb.bench_attn(B, H, Hkv, 128, ctx, sets=4, ns=ns)
