import os, subprocess, sys
code = '''
import sys, torch
sys.path.insert(0, "tools"); sys.path.insert(0, "text-generation-inference_amd")
import microbench as mb
K, N = int(sys.argv[1]), int(sys.argv[2])
mb.bench_gptq(32, K, N)
'''
for (K, N) in [(4096, 22016), (4096, 4096)]:
    for plan in ["4096,1,4", "1024,4,4"]:
        for dbg in ["0", "1", "2", "3"]:
            env = dict(os.environ); env["TGIS_GPTQ_PLAN"] = plan; env["TGIS_GPTQ_DBG"] = dbg
            r = subprocess.run([sys.executable, "-c", code, str(K), str(N)], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("gptq_gemm")]
            print(f"plan={plan} dbg={dbg}: {line[0] if line else r.stderr[-300:]}", flush=True)
