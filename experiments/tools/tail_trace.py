"""Phase timeline of the persistent decode tail (csrc/decode_tail.hip): per workgroup s_memrealtime stamps at the
phase edges -> median / max duration of every phase and barrier.  GPU box only.
    python experiments/build.py first;
    python experiments/tools/tail_trace.py            cfg3 shapes (Llama-2-7B, int4 GPTQ, fp16)
    python experiments/tools/tail_trace.py dense      cfg2 shapes (TinyLlama-1.1B, dense bf16);  M=<rows> in the environment"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "experiments"))
import native_experiments as nat  # noqa: E402  (tgis_amd.native bound to libtgis_experiments.so + the experimental calls)

dev = torch.device("cuda:0")
DENSE = len(sys.argv) > 1 and sys.argv[1] == "dense"
if DENSE:
    E, I, H, HKV, D, M, DT = 2048, 5632, 32, 4, 64, int(os.getenv("M", "16")), torch.bfloat16
else:
    E, I, H, HKV, D, M, DT = 4096, 11008, 32, 32, 128, int(os.getenv("M", "32")), torch.float16


def gptq(K, N, gate_up=False):
    G = K // 128
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
    qz = torch.randint(-2**31, 2**31 - 1, (G, N // 8), dtype=torch.int32, device=dev)
    sc = (torch.rand(G, N, device=dev) * 0.002 + 0.001).half()
    return nat.GptqWeight(qw, qz, sc, None, 4, 128, gate_up=gate_up)


def dense(K, N, gate_up=False):
    return nat.DenseWeight((torch.randn(N, K, device=dev) * 0.02).to(DT), gate_up=gate_up)


lin = dense if DENSE else gptq
sets = []
for _ in range(12 if DENSE else 3):  # rotate weight sets: no Infinity-Cache residency (88 MB per dense set)
    sets.append(nat.DecodeTail((lin(E, E), None), (lin(E, 2 * I, True), None), (lin(I, E), None),
                               torch.ones(E, device=dev).to(DT), torch.ones(E, device=dev).to(DT), 1e-5,
                               qkv=(lin(E, (H + 2 * HKV) * D), None), H=H, Hkv=HKV, D=D, rot_dim=D))
NS = len(sets)
attn = (torch.randn(M, E, device=dev) * 0.1).to(DT)
res = torch.randn(M, E, device=dev).to(DT)
cos = torch.ones(2048, D // 2, device=dev).to(DT)
sin = torch.zeros(2048, D // 2, device=dev).to(DT)
pos = torch.arange(M, dtype=torch.int32, device=dev)
slots = torch.arange(M, dtype=torch.int32, device=dev) * 32
kp = torch.zeros(M + 1, HKV, 32 * D, device=dev).to(DT)
vp = torch.zeros_like(kp)
lib = nat.load_library()
for i in range(2 * NS):
    sets[i % NS].run(attn, res, cos, sin, pos, slots, kp, vp)
torch.cuda.synchronize()
lib.tgis_llama_decode_tail_trace(1, None, 0)
names = ["o_proj", "bar", "norm1", "bar", "gate_up", "bar", "down", "bar", "norm2", "bar", "qkv", "bar", "rope"]
acc = []
for i in range(12):
    sets[i % NS].run(attn, res, cos, sin, pos, slots, kp, vp)
    torch.cuda.synchronize()
    buf = np.zeros((256, 16), dtype=np.int64)
    lib.tgis_llama_decode_tail_trace(-1, buf.ctypes.data_as(ctypes.c_void_p), 256)
    acc.append(buf[:, :14].copy())
lib.tgis_llama_decode_tail_trace(0, None, 0)
a = np.stack(acc[2:]).astype(np.float64) * 0.01  # us
t0 = a[:, :, 0].min(axis=1, keepdims=True)
print(f"workgroup entry spread: {np.median(a[:, :, 0].max(axis=1) - t0[:, 0]):.2f} us; whole launch (first entry -> last exit): "
      f"{np.median(a[:, :, 13].max(axis=1) - t0[:, 0]):.2f} us")
for j, nm in enumerate(names):
    d = a[:, :, j + 1] - a[:, :, j]
    # a phase lasts from the last workgroup's start to the last workgroup's end; per-workgroup spans show the skew
    span = a[:, :, j + 1].max(axis=1) - a[:, :, j].max(axis=1)
    print(f"{nm:8s} per-workgroup median {np.median(d):6.2f}  max {np.median(d.max(axis=1)):6.2f}   critical-path span {np.median(span):6.2f} us")
