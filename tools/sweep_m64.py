"""Plans with 12-wave blocks (TN=6, two k-parts, 64-row passes) against the default at M = 64."""
import os, subprocess, sys
plans = {
    (8192, 10240): ["4096,2,2,6", "2048,4,2,6", "8192,1,2,6"],
    (8192, 8192): ["2048,4,2,6", "1536,6,2,6", "4096,2,2,6"],
    (8192, 57344): ["8192,1,2,6", "4096,2,2,6"],
    (28672, 8192): ["7168,4,2,6", "5120,6,2,6", "3584,8,2,6"],
    (4096, 12288): ["2048,2,2,6", "1024,4,2,6", "4096,1,2,6"],
    (4096, 4096): ["1024,4,2,6", "512,8,2,6", "2048,2,2,6"],
    (4096, 22016): ["4096,1,2,6", "2048,2,2,6"],
    (11008, 4096): ["2048,6,2,6", "1536,8,2,6", "3072,4,2,6"],
}
code = '''
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, "text-generation-inference_amd")
import microbench as mb
mb.bench_gptq(64, int(sys.argv[1]), int(sys.argv[2]), sets=3)
'''
for (K, N), pls in plans.items():
    for pl in [None] + pls:
        env = dict(os.environ, TGIS_GPTQ_NOREDUCE="1")
        if pl:
            env["TGIS_GPTQ_PLAN"] = pl
        r = subprocess.run([sys.executable, "-c", code, str(K), str(N)], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("gptq_gemm")]
        print(f"plan={pl}: {line[0] if line else r.stderr[-300:]}", flush=True)
