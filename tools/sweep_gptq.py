import os, subprocess, sys
shapes = [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)]
plans = {
 (4096,12288): ["2048,2,4,3","4096,1,4,3","4096,1,4,2","2048,2,4,2","2048,2,2,4","1024,4,2,4","2048,2,4,4","4096,1,4,4","1024,4,4,4"],
 (4096,4096): ["1024,4,4,2","1024,4,2,2","2048,2,4,2","2048,2,4,3","1024,4,4,3","512,8,2,4","512,8,2,2","1024,4,4,4","2048,2,4,4"],
 (4096,22016): ["4096,1,4,3","4096,1,4,4","4096,1,2,4","2048,2,4,3","2048,2,2,4","4096,1,4,2","4096,1,2,3","4096,1,2,2"],
 (11008,4096): ["3072,4,4,2","3072,4,4,3","2048,6,4,2","1536,8,2,2","2048,6,2,4","3072,4,2,4","3072,4,4,4","2048,6,4,4","1536,8,4,2","2048,6,4,3"],
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
        for nored in (True,):
            env = dict(os.environ)
            if pl: env["TGIS_GPTQ_PLAN"] = pl
            if nored: env["TGIS_GPTQ_NOREDUCE"] = "1"
            r = subprocess.run([sys.executable, "-c", code, str(K), str(N)], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("gptq_gemm")]
            print(f"plan={pl} noreduce={nored}: {line[0] if line else r.stderr[-300:]}", flush=True)
