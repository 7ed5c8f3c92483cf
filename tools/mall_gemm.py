"""Does it matter where the int4 weights come from?  The streaming GEMM at the cfg3 shapes (M = 32) with its weights
(a) rotated over more sets than the 256 MiB Infinity Cache holds, (b) one set re-read every launch (L2 / Infinity Cache
resident), (c) cold, but swept into the caches by a prefetch launch just before (tools/floor/prefetch.hip), and (d) cold,
with the NEXT launch's weights prefetched on a second captured stream while this launch runs.  GPU only.
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/floor/libprefetch.so tools/floor/prefetch.hip
    python tools/mall_gemm.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from tgis_amd import native as nat  # noqa: E402
from microbench import timeit  # noqa: E402

dev = torch.device("cuda:0")
pf = ctypes.CDLL(os.path.join(ROOT, "tools", "floor", "libprefetch.so"))
pf.prefetch.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
M = 32
SHAPES = [("qkv", 4096, 12288, 0), ("o", 4096, 4096, 0), ("gate_up", 4096, 22016, 2), ("down", 11008, 4096, 0)]
PF_BLOCKS = int(os.environ.get("PF_BLOCKS", "256"))


def prefetch(h, nt=0, blocks=PF_BLOCKS):
    rc = pf.prefetch(h.image.data_ptr(), h.image.numel(), blocks, nt, torch.cuda.current_stream().cuda_stream)
    assert rc == 0


for name, K, N, act in SHAPES:
    nbytes = K * N // 2 + (K // 128) * N * 4
    sets = max(3, int(640e6 / nbytes))  # > 2 x the Infinity Cache
    hs = []
    for i in range(sets):
        qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
        qz = torch.randint(-2**31, 2**31 - 1, (K // 128, N // 8), dtype=torch.int32, device=dev)
        sc = (torch.rand(K // 128, N, device=dev) * 0.002 + 0.001).half()
        hs.append(nat.GptqWeight(qw, qz, sc, None, 4, 128, gate_up=act == 2))
        del qw, qz, sc
    x = torch.randn(M, K, device=dev).half()
    ws = nat.Workspace(hs[0].workspace_bytes(M), dev)
    out = torch.empty(M, N // 2 if act == 2 else N, device=dev, dtype=torch.float16)

    def gemm(i):
        if act == 2:
            nat.gptq_gemm(x, hs[i], ws, act=2, out=out)
        else:
            nat.gptq_gemm_partial(x, hs[i])

    iters = sets  # every set once per graph
    cold = timeit(gemm, sets, iters=iters)
    hot = timeit(lambda i: gemm(0), sets, iters=iters)
    few = timeit(lambda i: gemm(i % 3), sets, iters=iters)
    p_only = timeit(lambda i: prefetch(hs[i]), sets, iters=iters)
    p_nt = timeit(lambda i: prefetch(hs[i], nt=1), sets, iters=iters)

    def both(i):
        prefetch(hs[i])
        gemm(i)

    seq = timeit(both, sets, iters=iters)

    side = torch.cuda.Stream()

    def overlapped(i):
        # the weights of launch i were prefetched while launch i - 1 ran
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            prefetch(hs[(i + 1) % sets])
        gemm(i)
        main.wait_stream(side)

    ovl = timeit(overlapped, sets, iters=iters)
    print(f"{name:8s} {nbytes/1e6:5.1f} MB x {sets} sets: cold {cold*1e6:6.2f} us   3 sets {few*1e6:6.2f}   1 set {hot*1e6:6.2f}   "
          f"prefetch alone {p_only*1e6:6.2f} ({nbytes/p_only/1e12:.2f} TB/s; nt {p_nt*1e6:.2f})   prefetch, then GEMM {seq*1e6:6.2f} "
          f"(GEMM {1e6*(seq-p_only):5.2f})   GEMM || prefetch of the next {ovl*1e6:6.2f}", flush=True)
    del hs
    torch.cuda.empty_cache()
