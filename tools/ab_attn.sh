for v in "base:" "pipe2:TGIS_ATTN_PIPE=2" "nw4:TGIS_ATTN_NW=4" "nw4pipe:TGIS_ATTN_NW=4 TGIS_ATTN_PIPE=2" "nw1:TGIS_ATTN_NW=1" "base2:"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs python bench.py --no-cpu-baseline > gpurun_out/ab_$name.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$name.json"))
print("$name", d["ms_per_step"], d["graph_ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline_gemm"]["avg_launch_us"])
PY
done
