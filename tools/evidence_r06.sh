# Round-6 evidence on ONE box: every bench line of the final tree, the round profile (kernel stats + counter traffic of cfg3),
# per-config kernel breakdowns, one-rank TP steps and the GPU suite.   gpurun -- bash tools/evidence_r06.sh <tag>
TAG=${1:-r06}
O=gpurun_out
mkdir -p $O
python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg3.json
python bench.py --config tinyllama-1.1b --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg2.json
python bench.py --config starcoder-15b --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg5.json
python bench.py --config llama2-70b-gptq --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg4_1gpu.json
python bench.py --batch 64 --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg3_b64.json
python bench.py --churn 2>/dev/null | tail -1 > $O/${TAG}_churn_cfg3.json
bash tools/profile_round.sh $TAG > $O/${TAG}_profile.log 2>&1
bash tools/pmc_step.sh $TAG > $O/${TAG}_pmc.log 2>&1
tools/profile_config.sh ${TAG}_cfg2 --config tinyllama-1.1b > /dev/null 2>&1
tools/profile_config.sh ${TAG}_cfg5 --config starcoder-15b > /dev/null 2>&1
tools/profile_config.sh ${TAG}_cfg4 --config llama2-70b-gptq > /dev/null 2>&1
( echo "== tools/tp_segments_rccl1.py --steps 16 --tp N  (one rank's step at TP = N shard shapes, collectives on a world-size-1 RCCL group: no wire time)"
  for tp in 2 4 8; do timeout 600 python tools/tp_segments_rccl1.py --steps 16 --tp $tp 2>&1 | grep "tp$tp rank"; done
  echo "== --tp 8 --config llama2-70b-gptq --batch 64 --ctx 2048"
  timeout 900 python tools/tp_segments_rccl1.py --steps 16 --tp 8 --config llama2-70b-gptq --batch 64 --ctx 2048 2>&1 | grep "tp8 rank" ) > $O/${TAG}_tp_rank_steps.log 2>&1
python -m pytest tests -q -m gpu --durations=12 2>&1 | tail -20 > $O/${TAG}_gpu_tests.log
python - <<PY
import json
for c in ("cfg3", "cfg2", "cfg5", "cfg4_1gpu", "cfg3_b64"):
    try:
        d = json.load(open("$O/${TAG}_bench_%s.json" % c))
        print(c, d["ms_per_step"], d.get("ms_per_step_blocks"), d["value"], d.get("step_roofline", {}).get("frac_of_hbm_peak"), "attn", d["roofline"]["avg_launch_us"], d["roofline"]["frac"],
              "gemm", d.get("roofline_gemm", {}).get("avg_launch_us"), d.get("roofline_gemm", {}).get("frac"))
    except Exception as e:
        print(c, "failed", e)
PY
python - <<PY
import json
d = json.load(open("$O/${TAG}_churn_cfg3.json"))
for r in d["runs"]:
    print("churn", r["run"], r["pool"], r["graph_rows"], "p50", r["p50_ms"], "p99", r["p99_ms"], "captures", r["graph_captures"], r["graph_capture_ms"])
PY
cat $O/${TAG}_tp_rank_steps.log; tail -4 $O/${TAG}_gpu_tests.log
