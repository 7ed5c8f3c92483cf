"""add+RMSNorm time vs the form of its input: plain f16 rows, or S fp32 split-K slabs (deferred GEMM reduce)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
import torch
import microbench as mb
from tgis_amd import native as nat
dev = mb.dev
B, E = 32, 4096
res = [torch.randn(B, E, device=dev).half() for _ in range(4)]
w = torch.ones(E, device=dev).half()
x = [torch.randn(B, E, device=dev).half() for _ in range(4)]
t = mb.timeit(lambda i: nat.rmsnorm_residual(x[i], res[i], w, 1e-5), 4)
print(f"plain f16 input: {t*1e6:.2f} us")
for S in (1, 2, 4, 8):
    slabs = [torch.randn(S, 32, E, device=dev) for _ in range(4)]
    t = mb.timeit(lambda i: nat.rmsnorm_residual(nat.Partial(slabs[i], S, E, B, E, None), res[i], w, 1e-5), 4)
    print(f"S={S} fp32 slabs: {t*1e6:.2f} us")
