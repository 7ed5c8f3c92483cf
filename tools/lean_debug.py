import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from tgis_amd import native as nat
from lean_gemm import make
dev = torch.device("cuda:0")
K, N = 512, 64
h, w = make(K, N, 128, 1)
x = torch.randn(4, K).half()
xd = x.to(dev)
xs = nat.xsum(xd)
xf = x.float().view(4, K // 16, 4, 4)
ref_xs = torch.stack([xf[..., :2].sum((-1, -2)), xf[..., 2:].sum((-1, -2))], -1)
print("xsum err", (xs.cpu() - ref_xs).abs().max().item())
ws = nat.Workspace(h.workspace_bytes(4), dev)
ref = (x.double() @ w.double()).float()
old = nat.gptq_gemm(xd, h, ws).float().cpu()
new = nat.gptq_gemm_lean(xd, xs, h, ws).float().cpu()
print("old err", (old - ref).abs().max().item(), "new err", (new - ref).abs().max().item(), "ref max", ref.abs().max().item())
print("ref[0,:8]", ref[0, :8]); print("new[0,:8]", new[0, :8]); print("old[0,:8]", old[0, :8])
# one-hot probes: x = e_k -> row of W
for k in [0, 1, 2, 3, 4, 8, 16, 31, 32, 33, 64, 127, 128, 200, 300, 511]:
    x1 = torch.zeros(1, K).half(); x1[0, k] = 1.0
    x1d = x1.to(dev)
    n1 = nat.gptq_gemm_lean(x1d, nat.xsum(x1d), h, ws).float().cpu()
    print(k, "w", w[k, :4].tolist(), "got", n1[0, :4].tolist())
print("---- structured probes")
def run(x, tag):
    xd = x.to(dev).half()
    n1 = nat.gptq_gemm_lean(xd, nat.xsum(xd), h, ws).float().cpu()
    r = (xd.cpu().double() @ w.double()).float()
    print(tag, "max err", (n1 - r).abs().max().item(), "ref max", r.abs().max().item(), "err row0[:4]", (n1 - r)[0, :4].tolist())
run(torch.ones(1, K), "ones M=1")
run(torch.ones(4, K), "ones M=4")
run(torch.randint(-3, 4, (1, K)).float(), "ints M=1")
run(torch.randint(-3, 4, (4, K)).float(), "ints M=4")
run(torch.randn(1, K), "randn M=1")
run(torch.randn(1, K).half().float() * 0 + 0.3, "0.3 M=1")
x2 = torch.zeros(1, K); x2[0, :16] = torch.randn(16); run(x2, "randn first16")
x2 = torch.zeros(1, K); x2[0, :2] = torch.tensor([0.3, 0.7]); run(x2, "two vals")
x2 = torch.zeros(1, K); x2[0, 0] = 0.3; x2[0, 2] = 0.7; run(x2, "A+B vals")
x2 = torch.zeros(1, K); x2[0, 0] = 0.3; x2[0, 16] = 0.7; run(x2, "two blocks")
x2 = torch.zeros(1, K); x2[0, 0] = 0.3; x2[0, 32] = 0.7; run(x2, "two quarters")
x2 = torch.zeros(1, K); x2[0, 0] = 0.3; x2[0, 128] = 0.7; run(x2, "two groups")
x2 = torch.zeros(1, K); x2[0, 0] = 0.3; x2[0, 256] = 0.7; run(x2, "two kparts")
print("---- all one-hots")
bad = []
for k0 in range(0, K, 32):
    x1 = torch.zeros(32, K).half()
    for r in range(32):
        x1[r, k0 + r] = 1.0
    x1d = x1.to(dev)
    n1 = nat.gptq_gemm_lean(x1d, nat.xsum(x1d), h, ws).float().cpu()
    err = (n1 - w[k0:k0 + 32]).abs().max(1).values
    for r in range(32):
        if err[r] > 1e-4:
            bad.append((k0 + r, round(err[r].item(), 4)))
print("bad k:", bad)
# M=32 rows all ones vs M=1
run(torch.ones(32, K), "ones M=32")
x2 = torch.zeros(1, K); x2[0, :128] = 1; run(x2, "ones group0")
x2 = torch.zeros(1, K); x2[0, :64] = 1; run(x2, "ones first 64")
x2 = torch.zeros(1, K); x2[0, :32] = 1; run(x2, "ones first 32")
x2 = torch.zeros(1, K); x2[0, :17] = 1; run(x2, "ones first 17")
