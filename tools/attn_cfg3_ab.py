"""cfg3 decode attention (B=32, 32 heads, D=128, ctx 1024): waves per block x two-pages-in-flight x key splits.
TGIS_ATTN_NW / TGIS_ATTN_PIPE are read per launch (PIPE once per process: one process per PIPE value)."""
import os
import sys

sys.path.insert(0, "tools")
sys.path.insert(0, "text-generation-inference_amd")
import microbench as mb  # noqa: E402

ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for nw in (1, 2, 4):
    os.environ["TGIS_ATTN_NW"] = str(nw)
    for ns in (1, 2):
        print(f"PIPE={os.environ.get('TGIS_ATTN_PIPE', '-')} NW={nw} ", end="")
        mb.bench_attn(32, 32, 32, 128, ctx, ns=ns, sets=6)
