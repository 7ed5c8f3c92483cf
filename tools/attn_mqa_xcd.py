"""MQA decode attention (Starcoder shape: 48 q heads on 1 kv head): one block per 16-head chunk with the chunk blocks of a
group on ONE XCD (TGIS_ATTN_XCD, attention.hip) against the three-chunk blocks.  TGIS_ATTN_XCD is read once per process.
    python tools/attn_mqa_xcd.py [B H Hkv D ctx]"""
import os
import sys

sys.path.insert(0, "tools")
sys.path.insert(0, "text-generation-inference_amd")
import microbench as mb  # noqa: E402

B, H, Hkv, D, ctx = (int(v) for v in sys.argv[1:6]) if len(sys.argv) >= 6 else (32, 48, 1, 128, 4096)
tag = f"XCD={os.environ.get('TGIS_ATTN_XCD', '1')}"
for ch in (3, 1):
    os.environ["TGIS_ATTN_CH"] = str(ch)
    for nw in ((4,) if ch == 3 else (2, 4, 8)):
        os.environ["TGIS_ATTN_NW"] = str(nw)
        for ns in ((4, 6) if ch == 3 else (1, 2, 3, 4, 6, 8)):
            print(f"{tag} CH={ch} NW={nw} ", end="")
            mb.bench_attn(B, H, Hkv, D, ctx, ns=ns, sets=6)
