"""Decode attention: 16-head chunks per block x waves per block x key splits sweep (TGIS_ATTN_CH / TGIS_ATTN_NW are
read per launch).  GPU box only.
    python tools/attn_nw.py B H Hkv D ctx [chunk values, e.g. 1,3]"""
import os
import sys

sys.path.insert(0, "tools")
sys.path.insert(0, "text-generation-inference_amd")
import microbench as mb  # noqa: E402

B, H, Hkv, D, ctx = (int(v) for v in sys.argv[1:6]) if len(sys.argv) >= 6 else (16, 32, 4, 64, 512)
chs = [int(v) for v in sys.argv[6].split(",")] if len(sys.argv) > 6 else [0]
for ch in chs:
    if ch:
        os.environ["TGIS_ATTN_CH"] = str(ch)
    for nw in (4, 8):
        os.environ["TGIS_ATTN_NW"] = str(nw)
        for ns in (1, 2, 4, 8, 16, 32):
            print(f"CH={ch} NW={nw:2d} ", end="")
            mb.bench_attn(B, H, Hkv, D, ctx, ns=ns, sets=6)
