"""Plan sweep (TGIS_GPTQ_PLAN = "KR,S,WK,TN") of the GPTQ GEMM at Llama-2-70B shapes, M = 32 and 64."""
import os, subprocess, sys
plans = {
    (8192, 10240): ["2048,4,2,4", "2048,4,2,2", "4096,2,2,2", "4096,2,4,2", "2048,4,4,2", "1024,8,2,4", "4096,2,2,4"],
    (8192, 8192): ["2048,4,2,2", "2048,4,2,4", "1024,8,2,4", "4096,2,2,2", "2048,4,4,2", "1024,8,4,2"],
    (8192, 57344): ["8192,1,2,4", "4096,2,2,4", "8192,1,4,4", "8192,1,4,3", "8192,1,2,2", "8192,1,4,2", "8192,1,2,3"],
    (28672, 8192): ["3584,8,2,4", "7168,4,2,2", "7168,4,2,4", "4096,7,2,2", "7168,4,4,2", "3584,8,2,2", "2048,14,2,4"],
}
code = '''
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, "text-generation-inference_amd")
import microbench as mb
mb.bench_gptq(int(sys.argv[3]), int(sys.argv[1]), int(sys.argv[2]), sets=3)
'''
for M in (32, 64):
    for (K, N), pls in plans.items():
        for pl in [None] + pls:
            if M == 64 and pl and pl.split(",")[2] != "2":
                continue
            env = dict(os.environ)
            if pl:
                env["TGIS_GPTQ_PLAN"] = pl
            r = subprocess.run([sys.executable, "-c", code, str(K), str(N), str(M)], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("gptq_gemm")]
            print(f"plan={pl}: {line[0] if line else r.stderr[-300:]}", flush=True)
