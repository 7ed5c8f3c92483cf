"""CPU: the committed bench line (profiles/r01_bench_cfg3.json, printed by `python bench.py` on an MI355X) carries every
field the driver's contract names, with consistent arithmetic; and bench.py's byte accounting reproduces SURVEY.md
§8(d)'s worked numbers for the headline workload."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r01_bench_cfg3.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "tokens/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["n_gpus"] == 1 and d["dtype"] == "f16" and "synthetic" in d["data"]
    assert "workload" in d["config"] and "model" not in d["config"]
    B = d["config"]["global_batch"]
    assert abs(d["value"] - B / d["ms_per_step"] * 1e3) / d["value"] < 1e-3  # value = whole-job tokens per second
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # achieved = algorithmic bytes per launch / average launch duration
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) / r["achieved"] < 1e-2
    assert r["traffic"] is None or 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.2  # no wasted re-reads
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == "tokens/s"


def test_algorithmic_bytes_match_the_survey_worked_numbers():
    import bench
    from tgis_amd.models.custom_modeling.flash_llama_modeling import LlamaConfig

    kw, quantize, _dtype, B, ctx = bench.CONFIGS["llama2-7b-gptq"]
    ab = bench.algorithmic_bytes_per_step(LlamaConfig(**kw), quantize, B, ctx, tp=1)
    assert abs(ab["total"] - 20.8e9) / 20.8e9 < 0.01  # ~20.8 GB per cfg3 decode step
    assert abs(ab["total"] / 8e12 * 1e3 - 2.60) < 0.02  # 2.60 ms at 8 TB/s


def test_round5_bench_line_reports_the_median_of_three_timed_blocks():
    """three timed blocks of `steps` steps inside one invocation; ms_per_step / value are the median block,
    ms_per_step_range its fastest and slowest (profiles/r05b_bench_cfg3.json is a line printed by this tree's bench.py)."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r05b_bench_cfg3.json")))
    blocks = d["ms_per_step_blocks"]
    assert d["timed_blocks"] == 3 and len(blocks) == 3
    assert d["ms_per_step"] == sorted(blocks)[1]
    assert d["ms_per_step_range"] == [min(blocks), max(blocks)]
    B = d["config"]["global_batch"]
    assert abs(d["value"] - B / d["ms_per_step"] * 1e3) / d["value"] < 1e-3
    for k in ("roofline", "cpu_baseline", "step_roofline"):
        assert k in d, k
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3


def test_gpt2_cpu_bench_leg_runs_and_prints_the_contract_line():
    """BASELINE config 1 (`bench.py --config gpt2-cpu`, the README's command) prints one JSON line; round 5 shipped it with a
    NameError that no test saw."""
    import subprocess
    import sys

    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "gpt2-cpu", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 0 and d["dtype"] == "f32" and d["steps"] == 1 and d["value"] > 0
