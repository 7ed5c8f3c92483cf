import os, subprocess, sys
code = '''
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, "text-generation-inference_amd")
import microbench as mb
mb.bench_gptq(32, int(sys.argv[1]), int(sys.argv[2]))
'''
for (K, N, plans) in [(4096, 11008, ["4096,1,4,3", "4096,1,4,2", "4096,1,2,4", "4096,1,2,2", "4096,1,4,4"]),
                      (4096, 5504, ["4096,1,4,3", "4096,1,4,2", "4096,1,2,2", "4096,1,2,4"]),
                      (4096, 2752, ["4096,1,4,3", "4096,1,4,2", "4096,1,2,2"])]:
    for pl in plans:
        env = dict(os.environ, TGIS_GPTQ_PLAN=pl, TGIS_GPTQ_NOREDUCE="1")
        r = subprocess.run([sys.executable, "-c", code, str(K), str(N)], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("gptq_gemm")]
        print(pl, line[0] if line else r.stderr[-200:], flush=True)
