"""Round 6: the MQA decode attention (Starcoder shape: 48 q heads on one kv head, B 32, ctx 4096, bf16 in the
product; f16 here, same kernel) over chunks per block x two pages in flight x key splits.  TGIS_ATTN_CH / TGIS_ATTN_PIPE are
read once per process: run once per (CH, PIPE) pair, e.g.
    for ch in 3 2 1; do for pipe in 1 0; do TGIS_ATTN_CH=$ch TGIS_ATTN_PIPE=$pipe python tools/attn_mqa_variants.py; done; done"""
import os
import sys

sys.path.insert(0, "tools")
sys.path.insert(0, "text-generation-inference_amd")
import microbench as mb  # noqa: E402

B, H, Hkv, D, ctx = (int(v) for v in sys.argv[1:6]) if len(sys.argv) >= 6 else (32, 48, 1, 128, 4096)
tag = f"CH={os.environ.get('TGIS_ATTN_CH', '3')} PIPE={os.environ.get('TGIS_ATTN_PIPE', 'default')}"
for ns in (4, 6, 8, 12, 16):
    print(tag + " ", end="")
    try:
        mb.bench_attn(B, H, Hkv, D, ctx, ns=ns, sets=6)
    except Exception as e:  # noqa: BLE001
        print("failed", str(e)[:100])
