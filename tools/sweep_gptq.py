import os, subprocess, sys
shapes = [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)]
plans = {
 (4096,12288): ["4096,1,4","2048,2,4","1024,4,4","512,8,4","2048,2,8","1024,4,8","512,8,8","256,16,8"],
 (4096,4096): ["2048,2,4","1024,4,4","512,8,4","256,16,4","1024,4,8","512,8,8","256,16,8"],
 (4096,22016): ["4096,1,4","2048,2,4","1024,4,4","2048,2,8","1024,4,8","512,8,8"],
 (11008,4096): ["2816,4,4","1536,8,4","1024,11,4","768,15,4","1536,8,8","1024,11,8","768,15,8","512,22,8"],
}
code = '''
import sys, torch
sys.path.insert(0, "tools"); sys.path.insert(0, "text-generation-inference_amd")
import microbench as mb
K, N = int(sys.argv[1]), int(sys.argv[2])
mb.bench_gptq(32, K, N)
'''
for (K, N) in shapes:
    for pl in [None] + plans[(K, N)]:
        for nored in (False,):
            env = dict(os.environ)
            if pl: env["TGIS_GPTQ_PLAN"] = pl
            if nored: env["TGIS_GPTQ_NOREDUCE"] = "1"
            r = subprocess.run([sys.executable, "-c", code, str(K), str(N)], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("gptq_gemm")]
            print(f"plan={pl} noreduce={nored}: {line[0] if line else r.stderr[-300:]}", flush=True)
