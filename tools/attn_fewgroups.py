"""Round 6: key splits for decode attention with FEW (sequence, kv head) groups (< 128: small batches, TP ranks) — how many bytes
should one 8-wave block stream before a second block pays for the in-launch merge?  D = 128 and D = 64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
import microbench as mb
for B, H, Hkv, D, ctx in ((1, 32, 32, 128, 512), (1, 32, 32, 128, 1024), (1, 32, 32, 128, 4096), (2, 32, 32, 128, 1024), (3, 32, 32, 128, 1024),
                          (2, 32, 32, 128, 512), (8, 32, 4, 64, 1024), (16, 32, 4, 64, 1024), (4, 64, 8, 128, 1024), (8, 64, 8, 128, 512),
                          (1, 32, 4, 64, 2048), (64, 8, 1, 128, 1024), (16, 4, 4, 128, 2048)):
    for ns in (None, 1, 2, 3, 4, 6, 8):
        mb.bench_attn(B, H, Hkv, D, ctx, sets=4, ns=ns)
