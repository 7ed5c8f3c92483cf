set -x
O=gpurun_out
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r03_bench_cfg3.json
python bench.py --config tinyllama-1.1b --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_cfg2.json
python bench.py --config starcoder-15b --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_cfg5.json
python bench.py --config llama2-70b-gptq --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_cfg4_1gpu.json
(cd gpurun_in/r02 && python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1) > $O/r03_bench_cfg3_round2_same_box.json
(echo "# register-ring lean kernel (TGIS_LEAN_LD=0) vs the streaming kernel"; TGIS_LEAN_LD=0 python tools/lean_gemm.py 32 both; echo "# loader/consumer (LDS-DMA) form (TGIS_LEAN_LD=1)"; TGIS_LEAN_LD=1 python tools/lean_gemm.py 32 both) 2>&1 | grep -v amdgpu.ids > $O/r03_lean_gemm.log
python tools/rope_gemm_bench.py 2>&1 | grep -v amdgpu.ids > $O/r03_rope_gemm.log
for s in "4096 12288 0" "4096 22016 2" "11008 4096 0" "4096 4096 0"; do python tools/trace_gemm.py $s 2>&1 | grep -E "^K=|dt |blocks"; done > $O/r03_gemm_timeline.log
for s in "4096 12288 0" "4096 22016 2"; do TGIS_LEAN_LD=1 python tools/trace_ld.py $s 2>&1 | grep -v amdgpu.ids; done > $O/r03_ld_timeline.log
./tools/floor/lean > $O/r03_lean_step.log 2>&1
./tools/floor/ldsdma > $O/r03_ldsdma.log 2>&1
cut -c1-200 $O/r03_bench_cfg2.json $O/r03_bench_cfg5.json $O/r03_bench_cfg4_1gpu.json
(export TGIS_LEAN_LD=0; echo "# lean register-ring kernel, us per launch (qkv / o / gate_up / down); LEAN_ABL bits: 1 no correction MFMA + fold, 2 no row-sum staging, 4 no x loads, 8 no x stores + chunk syncs, 16 no weight refills, 32 no arithmetic"; echo "== full"; python tools/lean_gemm.py 32 time 2>&1 | grep -v amdgpu.ids | sed "s/old.*lean/lean/"; for a in 3 6 14 30 32 46 62; do echo "== LEAN_ABL=$a"; TGIS_HIP_LIB=$PWD/text-generation-inference_amd/lib/lean_abl$a.so python tools/lean_gemm.py 32 time 2>&1 | grep -v amdgpu.ids | sed "s/old.*lean/lean/"; done) > $O/r03_lean_ablations.log
