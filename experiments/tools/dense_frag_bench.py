"""Round 5: the dense decode GEMM on a fragment-order operand (csrc/dense_wide_body.h) against the row-major streaming kernel
(csrc/dense_gemm_body.h), GPU time per launch from a captured graph over rotating weights.  GPU box only; runs on the tree of commit 1e13390 (the round-5 commit that had the kernel wired into the library).
    python tools/dense_frag_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
import microbench as mb  # noqa: E402
from tgis_amd import native as nat  # noqa: E402

dev = mb.dev
SHAPES = [("tinyllama qkv", 16, 2048, 2560), ("tinyllama o", 16, 2048, 2048), ("tinyllama gate_up", 16, 2048, 11264),
          ("tinyllama down", 16, 5632, 2048), ("starcoder c_attn", 32, 6144, 6400), ("starcoder c_proj", 32, 6144, 6144),
          ("starcoder c_fc", 32, 6144, 24576), ("starcoder mlp c_proj", 32, 24576, 6144), ("llama-7b lm_head", 32, 4096, 32000)]
for name, M, K, N in SHAPES:
    sets = max(2, int(700e6 // (K * N * 2)))
    ws_ = [nat.DenseWeight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(sets)]
    x = torch.randn(M, K, device=dev).bfloat16()
    xf = nat.FragAct.from_rows(x)
    ws = nat.Workspace(ws_[0].workspace_bytes(M), dev)
    f32 = N == 32000
    for plan in (os.environ.get("TGIS_DENSE_WIDE_PLAN"),):
        t_row = mb.timeit(lambda i: nat.dense_gemm(x, ws_[i], ws, out_f32=f32), sets)
        t_frag = mb.timeit(lambda i: nat.dense_gemm(xf, ws_[i], ws, out_f32=f32), sets)
        t_rowp = mb.timeit(lambda i: nat.dense_gemm_partial(x, ws_[i]), sets) if not f32 else float("nan")
        t_fragp = mb.timeit(lambda i: nat.dense_gemm_partial(xf, ws_[i]), sets) if not f32 else float("nan")
        mbytes = K * N * 2 / 1e6
        print(f"{name:22s} M={M:2d} {K}x{N} ({mbytes:6.1f} MB, {sets} sets): finished row-major {t_row*1e6:6.1f} us, fragments "
              f"{t_frag*1e6:6.1f} us | slabs row-major {t_rowp*1e6:6.1f} us, fragments {t_fragp*1e6:6.1f} us "
              f"({mbytes / (min(t_frag, t_fragp if t_fragp == t_fragp else t_frag) * 1e6):.2f} TB/s)", flush=True)
    del ws_
    torch.cuda.empty_cache()
