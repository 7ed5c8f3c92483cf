# 64-row decode batches: the fragment-order kernel with two row blocks (MR = 2) against the 64-row passes of the streaming kernel
for cfgb in "llama2-7b-gptq:64" "llama2-70b-gptq:64" "llama2-7b-gptq:48"; do
  cfg=${cfgb%%:*}; b=${cfgb#*:}
  for mr in 32 64; do
    TGIS_GPTQ_FRAGMENTS_MAX_ROWS=$mr python bench.py --config $cfg --batch $b --steps 16 --no-cpu-baseline > gpurun_out/ab_rows_$cfg.$b.$mr.json 2>/dev/null
    python - <<PY
import json
d=json.load(open("gpurun_out/ab_rows_$cfg.$b.$mr.json"))
print("$cfg B=$b max_rows=$mr", d["ms_per_step"], d["graph_ms_per_step"], "attn", d["roofline"]["avg_launch_us"], "gemm", d["roofline_gemm"]["avg_launch_us"], d["roofline_gemm"]["frac"])
PY
  done
done
