"""Can the issue-bound GPTQ GEMMs of one half-batch hide under the HBM-bound attention of the other half?

cfg3 shapes.  Sequential full batch (what the step does today): per layer attention(B=32) then the four GEMMs (M=32).
Micro-batched: two half-batches, each layer = attention(B=16) || GEMMs(M=16) on two streams (graph branches), twice.
Prints the GPU time per layer of each arrangement (captured graphs, weights and KV rotated past the Infinity Cache)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
import microbench as mb
from tgis_amd import native as nat
dev = mb.dev
H, Hkv, D, ctx, gs = 32, 32, 128, 1024, 128
E, I = 4096, 11008
SETS, LAYERS = 6, 12


def gptq(K, N, gate_up=False):
    G = K // gs
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
    qz = torch.randint(-2**31, 2**31 - 1, (G, N // 8), dtype=torch.int32, device=dev)
    sc = (torch.rand(G, N, device=dev) * 0.002 + 0.001).half()
    return nat.GptqWeight(qw, qz, sc, None, 4, gs, gate_up=gate_up)


weights = [(gptq(E, 3 * E), gptq(E, E), gptq(E, 2 * I, True), gptq(I, E)) for _ in range(SETS)]
pages_per = ctx // 32
total = 32 * pages_per
pools = [(torch.randn(total, Hkv, 32 * D, device=dev, dtype=torch.float16), torch.randn(total, Hkv, 32 * D, device=dev, dtype=torch.float16))
         for _ in range(SETS)]


class Half:
    def __init__(self, B, first_seq):
        self.B = B
        self.x = torch.randn(B, E, device=dev).half()
        self.qkv = torch.empty(B, 3 * E, device=dev, dtype=torch.float16)
        self.att = torch.empty(B, E, device=dev, dtype=torch.float16)
        self.o = torch.empty(B, E, device=dev, dtype=torch.float16)
        self.mid = torch.empty(B, I, device=dev, dtype=torch.float16)
        self.dn = torch.empty(B, E, device=dev, dtype=torch.float16)
        self.ws = nat.Workspace(64 << 20, dev)
        self.bt = (torch.arange(B * pages_per, device=dev).int() + first_seq * pages_per).view(B, pages_per).contiguous()
        self.ctxl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
        self.cu = torch.arange(B + 1, dtype=torch.int32, device=dev)
        self.ns = nat.attn_num_splits(B, Hkv, H, 1, ctx)

    def attn(self, i):
        k, v = pools[i % SETS]
        nat.attn_paged(self.qkv, 3 * E, k, v, self.bt, self.ctxl, self.cu, self.att, self.B, H, Hkv, D, 1, ctx, D ** -0.5, self.ns, None)

    def gemms(self, i):
        wq, wo, wgu, wd = weights[i % SETS]
        nat.gptq_gemm(self.x, wq, self.ws, out=self.qkv)
        nat.gptq_gemm(self.att, wo, self.ws, out=self.o)
        nat.gptq_gemm(self.o, wgu, self.ws, act=2, out=self.mid)
        nat.gptq_gemm(self.mid, wd, self.ws, out=self.dn)


def time_graph(body, reps=5):
    body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps / LAYERS * 1e3  # us per layer


full, ha, hb = Half(32, 0), Half(16, 0), Half(16, 16)
side = torch.cuda.Stream()


def seq_full():
    for i in range(LAYERS):
        full.attn(i); full.gemms(i)


def only(fn):
    def body():
        for i in range(LAYERS):
            fn(i)
    return body


def overlapped():
    cur = torch.cuda.current_stream()
    for i in range(LAYERS):
        # phase 1: attention of half A || GEMMs of half B; phase 2: the other way round
        for a, b in ((ha, hb), (hb, ha)):
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                b.gemms(i)
            a.attn(i)
            cur.wait_stream(side)


def free_running():
    """Upper bound on overlap: the two halves as independent chains, no per-layer joins."""
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        for i in range(LAYERS):
            hb.gemms(i); hb.attn(i)
    for i in range(LAYERS):
        ha.attn(i); ha.gemms(i)
    cur.wait_stream(side)


print(f"attention B=32 alone          {time_graph(only(full.attn)):7.1f} us/layer")
print(f"GEMMs M=32 alone              {time_graph(only(full.gemms)):7.1f} us/layer")
print(f"sequential full batch         {time_graph(seq_full):7.1f} us/layer   <- today")
print(f"attention B=16 alone          {time_graph(only(ha.attn)):7.1f} us/layer")
print(f"GEMMs M=16 alone              {time_graph(only(ha.gemms)):7.1f} us/layer")
print(f"2 x (attn16 || gemms16)       {time_graph(overlapped):7.1f} us/layer")
print(f"two free-running half chains  {time_graph(free_running):7.1f} us/layer")
