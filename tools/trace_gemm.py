"""In-kernel timeline of the GPTQ GEMM (debug build with -DTGIS_TRACE): per-wave s_memtime stamps.

    python tools/trace_gemm.py K N [act] [plan]      (on the GPU box; builds lib/trace.so with hipcc first)
"""
import ctypes
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pkg = os.path.join(root, "text-generation-inference_amd")
lib = os.path.join(pkg, "lib", "trace.so")
if not os.path.exists(lib) or os.environ.get("TRACE_REBUILD"):
    srcs = [os.path.join(pkg, "csrc", f) for f in sorted(os.listdir(os.path.join(pkg, "csrc"))) if f.endswith(".hip")]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DTGIS_TRACE",
                           "-o", lib] + srcs)
os.environ["TGIS_HIP_LIB"] = lib
sys.path.insert(0, pkg)
import torch  # noqa: E402
from tgis_amd import native as nat  # noqa: E402

K, N = int(sys.argv[1]), int(sys.argv[2])
act = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if len(sys.argv) > 4 and sys.argv[4] != "-":
    os.environ["TGIS_GPTQ_PLAN"] = sys.argv[4]
LEAN = len(sys.argv) > 5 and sys.argv[5] == "lean"  # the lean kernel (TGIS_LEAN_RING / TGIS_LEAN_LD select its form)
dev = torch.device("cuda:0")
gs, M = 128, int(os.environ.get("TRACE_M", "32"))
G = K // gs
sets = []
for i in range(4):
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
    qz = torch.randint(-2**31, 2**31 - 1, (G, N // 8), dtype=torch.int32, device=dev)
    sc = (torch.rand(G, N, device=dev) * 0.002 + 0.001).half()
    sets.append(nat.GptqWeight(qw, qz, sc, None, 4, gs, gate_up=(act == 2)))
x = torch.randn(M, K, device=dev).half()
ws = nat.Workspace(sets[0].workspace_bytes(M), dev)
NB = 4096
trace = torch.zeros(NB * 16 * 32, dtype=torch.int64, device=dev)
L = nat.load_library()
set_trace = L.tgis_debug_set_trace_lean if LEAN else L.tgis_debug_set_trace
set_trace.argtypes = [ctypes.c_void_p]
xs = nat.xsum(x)


def run(i):
    if LEAN:
        return nat.gptq_gemm_lean(x, xs, sets[i], ws, act=act)
    return nat.gptq_gemm(x, sets[i], ws, act=act)


for i in range(3):
    run(i)
torch.cuda.synchronize()
assert set_trace(ctypes.c_void_p(trace.data_ptr())) == 0
trace.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run(3)
e1.record()
torch.cuda.synchronize()
print(f"M={M} K={K} N={N} act={act}: event time {e0.elapsed_time(e1) * 1e3:.1f} us")
t = trace.view(NB, 16, 32).cpu()
used = t[:, :, 0] != 0
t0 = t[:, :, 0][used].min().item()
names = {0: "entry", 1: "weights issued", 2: "x0 staged+barrier", 3: "chunk0 done", 4: "chunk0 barrier", 5: "chunk1 done",
         6: "chunk1 barrier", 7: "chunk2 done", 8: "chunk2 barrier", 9: "last chunk done", 10: "last barrier",
         11: "reduce barrier", 12: "epilogue start", 13: "epilogue done"}
print(f"blocks used: {int(used.any(dim=1).sum())}, waves: {int(used.sum())}   (ticks of s_memtime, relative to first entry)")
for i in range(14):
    v = t[:, :, i]
    m = v != 0
    if not m.any():
        continue
    r = (v[m] - t0).float()
    q = torch.quantile(r, torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0]))
    print(f"  {i:2d} {names[i]:20s} n={int(m.sum()):6d}  min={q[0]:8.0f} p10={q[1]:8.0f} p50={q[2]:8.0f} p90={q[3]:8.0f} max={q[4]:8.0f}")
# per-wave durations
for i in range(4):
    names[16 + 4 * i] = f"c1 step{i} top"
    names[17 + 4 * i] = f"c1 step{i} dequant done"
    names[18 + 4 * i] = f"c1 step{i} mfma issued"
names[9] = "last chunk done"
pairs = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8), (8, 9), (6, 9), (4, 9), (2, 9), (9, 11), (11, 12), (12, 13), (0, 12)]
for a_, b_ in pairs:
    va, vb = t[:, :, a_], t[:, :, b_]
    m = (va != 0) & (vb != 0)
    if m.any():
        d = (vb[m] - va[m]).float()
        print(f"  dt {names[a_]:>24s} -> {names[b_]:24s} p50={d.median():8.0f} p90={torch.quantile(d, 0.9):8.0f} max={d.max():8.0f}")

# chip-wide picture from s_memrealtime (100 MHz): when do waves enter, when do reducer waves reach the epilogue
rt0, rt1 = t[:, :, 14], t[:, :, 15]
m0, m1 = rt0 != 0, rt1 != 0
if m0.any() and m1.any():
    base = rt0[m0].min().item()
    e = (rt0[m0] - base).float() * 10.0
    x = (rt1[m1] - base).float() * 10.0
    qs = torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0])
    print("  realtime ns since first wave entry:  entry  ", [int(v) for v in torch.quantile(e, qs)])
    print("                                       epilogue", [int(v) for v in torch.quantile(x, qs)])

# per-wave picture: when does each wave of a block finish its last chunk, relative to the block's earliest entry
t9 = t[:, :, 9].double()
t0b = t[:, :, 0].double()
valid = (t[:, :, 9] != 0) & (t[:, :, 0] != 0)
if valid.any():
    nw = int(valid.any(dim=0).sum())
    rows_ = valid.any(dim=1)
    base_b = torch.where(valid, t0b, torch.full_like(t0b, float("inf"))).min(dim=1).values
    rel = (t9 - base_b[:, None])
    ent = (t0b - base_b[:, None])
    print("  per wave (index = wk * TN + wn): median cycles from the block's first entry to [entry | last chunk done]")
    for w_ in range(nw):
        m_ = valid[:, w_] & rows_
        print(f"    wave {w_:2d}: entry {ent[m_, w_].median():7.0f}   last chunk done {rel[m_, w_].median():8.0f}   p90 {torch.quantile(rel[m_, w_], 0.9):8.0f}")
