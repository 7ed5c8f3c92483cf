"""In-kernel timeline of the loader / consumer int4 GEMM (debug build with -DTGIS_TRACE, built here as lib/trace.so):
per-wave s_memrealtime stamps (100 MHz).   python tools/trace_ld.py K N [act]"""
import ctypes
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pkg = os.path.join(root, "text-generation-inference_amd")
os.environ["TGIS_HIP_LIB"] = os.path.join(pkg, "lib", "trace.so")
sys.path.insert(0, pkg)
import torch  # noqa: E402
from tgis_amd import native as nat  # noqa: E402

K, N = int(sys.argv[1]), int(sys.argv[2])
act = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda:0")
M, G = 32, K // 128
sets = []
for i in range(4):
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
    qz = torch.randint(-2**31, 2**31 - 1, (G, N // 8), dtype=torch.int32, device=dev)
    sc = (torch.rand(G, N, device=dev) * 0.002 + 0.001).half()
    sets.append(nat.GptqWeight(qw, qz, sc, None, 4, 128, gate_up=(act == 2)))
x = torch.randn(M, K, device=dev).half()
xs = nat.xsum(x)
ws = nat.Workspace(sets[0].workspace_bytes(M), dev)
NB = 4096
trace = torch.zeros(NB * 16 * 32, dtype=torch.int64, device=dev)
L = nat.load_library()
L.tgis_debug_set_trace_lean.argtypes = [ctypes.c_void_p]


def run(i):
    if act == 2:
        return nat.gptq_gemm_lean(x, xs, sets[i], ws, act=2, want_xs=True)
    return nat.gptq_gemm_partial_lean(x, xs, sets[i])


for i in range(3):
    run(i)
torch.cuda.synchronize()
assert L.tgis_debug_set_trace_lean(ctypes.c_void_p(trace.data_ptr())) == 0
trace.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run(3)
e1.record()
torch.cuda.synchronize()
print(f"K={K} N={N} act={act}: event time {e0.elapsed_time(e1) * 1e3:.1f} us (with stamps)")
t = trace.view(NB, 16, 32).cpu()
used = t[:, :, 0] != 0
t0 = t[:, :, 0][used].min().item()
print(f"blocks {int(used.any(dim=1).sum())}, waves {int(used.sum())}; microseconds since the first entry")
ld = {0: "entry", 1: "init barrier", 2: "round 0 issued", 3: "all issued", 4: "all landed", 6: "barrier 2"}
cs = {0: "entry", 1: "init barrier", 2: "x0 staged + sync", 3: "chunk 0 landed", 4: "chunk 0 done", 9: "last chunk landed",
      5: "loop done", 6: "barrier 2", 7: "reduce barrier", 8: "epilogue done"}
for role, sel, names in (("loader", slice(0, 1), ld), ("consumers", slice(1, 16), cs)):
    print(role)
    for i in sorted(names, key=lambda k: (k if k != 9 else 4.5)):
        v = t[:, sel, i]
        m = v != 0
        if not m.any():
            continue
        r = (v[m] - t0).float() / 100.0
        q = torch.quantile(r, torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0]))
        print(f"  {names[i]:20s} n={int(m.sum()):6d}  min={q[0]:6.2f} p10={q[1]:6.2f} p50={q[2]:6.2f} p90={q[3]:6.2f} max={q[4]:6.2f}")
