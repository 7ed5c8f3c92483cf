"""Phase timeline of the persistent decode tail (csrc/decode_tail.hip) at cfg3 shapes: per workgroup s_memrealtime
stamps at the phase edges -> median / max duration of every phase and barrier.  GPU box only."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
from tgis_amd import native as nat  # noqa: E402

dev = torch.device("cuda:0")
E, I, H, D, M = 4096, 11008, 32, 128, int(os.getenv("M", "32"))


def gptq(K, N, gate_up=False):
    G = K // 128
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
    qz = torch.randint(-2**31, 2**31 - 1, (G, N // 8), dtype=torch.int32, device=dev)
    sc = (torch.rand(G, N, device=dev) * 0.002 + 0.001).half()
    return nat.GptqWeight(qw, qz, sc, None, 4, 128, gate_up=gate_up)


sets = []
for _ in range(3):  # rotate weight sets: no Infinity-Cache residency
    sets.append(nat.DecodeTail((gptq(E, E), None), (gptq(E, 2 * I, True), None), (gptq(I, E), None),
                               torch.ones(E, device=dev).half(), torch.ones(E, device=dev).half(), 1e-5,
                               qkv=(gptq(E, 3 * E), None), H=H, Hkv=H, D=D, rot_dim=D))
attn = torch.randn(M, E, device=dev).half() * 0.1
res = torch.randn(M, E, device=dev).half()
cos = torch.ones(2048, D // 2, device=dev).half()
sin = torch.zeros(2048, D // 2, device=dev).half()
pos = torch.arange(M, dtype=torch.int32, device=dev)
slots = torch.arange(M, dtype=torch.int32, device=dev) * 32
kp = torch.zeros(M + 1, H, 32 * D, device=dev).half()
vp = torch.zeros_like(kp)
lib = nat.load_library()
for i in range(6):
    sets[i % 3].run(attn, res, cos, sin, pos, slots, kp, vp)
torch.cuda.synchronize()
lib.tgis_llama_decode_tail_trace(1, None, 0)
names = ["o_proj", "bar", "norm1", "bar", "gate_up", "bar", "down", "bar", "norm2", "bar", "qkv", "bar", "rope"]
acc = []
for i in range(12):
    sets[i % 3].run(attn, res, cos, sin, pos, slots, kp, vp)
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
