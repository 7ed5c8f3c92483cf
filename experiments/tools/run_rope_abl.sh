# NOTE (round 5): one-off of round 4 (rope-epilogue ablation builds libtgis_abl[1-4].so of commit 1892754).
L=text-generation-inference_amd/lib
TGIS_GPTQ_WIDE_PLAN=2,1 python tools/rope_frag_bench.py
python tools/rope_frag_bench.py
for n in 1 2 3 4; do TGIS_HIP_LIB=$PWD/$L/libtgis_abl$n.so TGIS_GPTQ_WIDE_PLAN=2,1 python tools/rope_frag_bench.py; done
