import os, subprocess, sys
code = '''
import sys, torch
sys.path.insert(0, "tools"); sys.path.insert(0, "text-generation-inference_amd")
import microbench as mb
K, N = int(sys.argv[1]), int(sys.argv[2])
mb.bench_gptq(32, K, N)
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for (K, N, plan) in [(4096, 22016, "4096,1,1"), (4096, 22016, "4096,1,2"), (4096, 22016, "4096,1,4"), (4096, 22016, "1024,4,1"), (4096, 22016, "1024,4,4"), (4096, 22016, "512,8,2")]:
    for v in ["", "LOADONLY"]:
        env = dict(os.environ); env["TGIS_GPTQ_PLAN"] = plan; env["TGIS_GPTQ_NOREDUCE"] = "1"
        if v: env["TGIS_HIP_LIB"] = os.path.join(root, "text-generation-inference_amd", "lib", f"abl_{v}.so")
        r = subprocess.run([sys.executable, "-c", code, str(K), str(N)], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("gptq_gemm")]
        print(f"plan={plan} abl={v or 'none'}: {line[0] if line else r.stderr[-300:]}", flush=True)
