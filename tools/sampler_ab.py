"""Same inputs through the sampler with rows in registers and (TGIS_SAMPLER_GLOBAL_ROWS=1) rows in memory: outputs must be
bit-identical.  Runs itself twice as subprocesses."""
import os, subprocess, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.join(ROOT, "text-generation-inference_amd"))
    import torch
    from tgis_amd import native as nat
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    h = hashlib.sha256()
    for (B, V) in ((32, 32000), (5, 32768), (7, 1000), (3, 31999)):
        logits = (torch.randn(B, V, generator=g) * 4).to(dev)
        temp = (torch.rand(B, generator=g) * 1.5 + 0.3).to(dev)
        topk = torch.randint(0, 80, (B,), generator=g).int().to(dev)
        cut = (torch.rand(B, generator=g) * 0.5).to(dev)
        typ = (torch.rand(B, generator=g) * 0.6 + 0.4).to(dev); typ[::3] = 1.0
        pen = (torch.rand(B, generator=g) + 0.8).to(dev)
        ids = torch.randint(0, V, (B, 40), generator=g).to(dev)
        do = (torch.rand(B, generator=g) > 0.3).int().to(dev)
        rng = torch.stack([torch.arange(B) + 11, torch.zeros(B, dtype=torch.int64)], 1).to(dev)
        for rep in range(2):
            out = nat.warp_sample(logits, temperature=temp, top_k=topk, top_p_cut=cut, typical_p=typ, rep_penalty=pen,
                                  input_ids=ids, exclude_id=-1, do_sample=do, rng=rng)
            for t in out:
                h.update(t.cpu().numpy().tobytes())
            h.update(rng.cpu().numpy().tobytes())
    print(h.hexdigest())
else:
    a = subprocess.run([sys.executable, __file__, "x"], capture_output=True, text=True, env=dict(os.environ))
    b = subprocess.run([sys.executable, __file__, "x"], capture_output=True, text=True, env=dict(os.environ, TGIS_SAMPLER_GLOBAL_ROWS="1"))
    ha, hb = a.stdout.strip().splitlines()[-1:], b.stdout.strip().splitlines()[-1:]
    print("registers:", ha, "memory:", hb, "IDENTICAL" if ha == hb and ha else "DIFFERENT")
    if not ha or ha != hb:
        print(a.stderr[-2000:], b.stderr[-2000:])
